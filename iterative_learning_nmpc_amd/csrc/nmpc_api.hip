// nmpc_api.hip -- host side of the C-ABI declared in include/nmpc.h (libnmpc_hip.so).
// Single translation unit: kernels are included below.  Build: csrc/build.sh (hipcc, gfx950).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "nmpc_device_guard.hpp"
#include "nmpc_solve.hip"
#include "nmpc_wb.hip"
#include "nmpc_aux.hip.inc"
#include "nmpc_rollout.hip.inc"
#include "nmpc_wb_rollout.hip.inc"

namespace {

thread_local std::string g_create_error;

struct Handle {
    nmpc_dims dims{};
    int device = 0;
    int nx = 0, nu = 0, np = 0, ng = 0;
    int ny = 0, nye = 0;     // cost residuals of a stage / of the terminal node (lengths of W, yref / W_e, yref_e)
    float* ws = nullptr;
    size_t ws_bytes = 0;
    size_t ws_stride = 0;    // floats per problem
    float* dbg = nullptr;    // diagnostic builds only (nmpc_debug_set_buffer)
    float* roll = nullptr;   // rollout problem tensors: x0 alias, yref, yref_e, params (B_max sized)
    int n_cu = 256;          // compute units of the device
    int force_variant = 0;   // NMPC_QP_VARIANT: 0 chosen per call by batch size and horizon (launch_solve), 1 resident, 2 lean (tests, tuning)
    int all_patterns = 0;    // nmpc_set_contact_patterns: 1 = kernel with a static stage body per contact pattern
    const int* skip = nullptr;   // nmpc_set_skip: problems with skip[b] & skip_mask != 0 are left out of the solves
    int skip_mask = 0;
    bool ws_dirty = false;   // a dense-LQ call left foreign padding in the tile workspace
    bool mp_set = false, w_set = false;
    nmpc::ModelParams mp{};
    float W[96]{}, We[96]{};
    bool pos_rows = false;   // whole-body: some foot-placement weight is non-zero
    float reg = 1e-6f, reg_e = 1e-5f;
    int max_sqp = 1, n_ipm = 6, line_search = 0;
    float nlp_tol = 0.0f, qp_tol = 1e-2f;
    float mu0 = 10.0f, sigma = 0.2f, s_min = 1.0f, gamma = 0.995f, tau_min = 0.1f, rho = 1e3f;
    std::string err;
};

int fail(Handle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                  \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(h, NMPC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// the handle's configuration as kernel arguments (pointers and batch size left to the caller)
nmpc::SolveArgs base_args(const Handle* h) {
    nmpc::SolveArgs a{};
    a.mp = h->mp;
    std::memcpy(a.W, h->W, sizeof(a.W));        // the tile-family models have ny <= 32, ny_e <= 16
    std::memcpy(a.We, h->We, sizeof(a.We));
    a.reg = h->reg; a.reg_e = h->reg_e;
    for (int j = 0; j < 16; ++j)
        a.rs_free[j] = (j < h->nu) ? 1.0f / std::sqrt(h->W[h->nx + j] + h->reg) : 1.0f;
    a.N = h->dims.N;
    a.max_sqp = h->max_sqp; a.n_ipm = h->n_ipm; a.line_search = h->line_search;
    a.nlp_tol = h->nlp_tol; a.mu0 = h->mu0; a.sigma = h->sigma; a.s_min = h->s_min;
    a.gamma = h->gamma; a.tau_min = h->tau_min; a.rho = h->rho;
    a.ws = h->ws;
    a.dbg = h->dbg;
    a.skip = h->skip_mask ? h->skip : nullptr; a.skip_mask = h->skip_mask;
    return a;
}

template <class M>
size_t ws_floats_per_problem(int N) { return nmpc::WsLayout<M>(N).stride; }

// whole-body model (nmpc_wb.hip): the handle's configuration as kernel arguments
nmpc::wb::WbArgs wb_args(const Handle* h) {
    nmpc::wb::WbArgs a{};
    a.mp = h->mp;
    std::memcpy(a.W, h->W, sizeof(a.W));
    std::memcpy(a.We, h->We, sizeof(a.We));
    a.reg = h->reg; a.reg_e = h->reg_e;
    a.N = h->dims.N;
    a.max_sqp = h->max_sqp; a.n_ipm = h->n_ipm;
    a.precision = h->dims.precision;
    a.pos_rows = h->pos_rows ? 1 : 0;
    a.nlp_tol = h->nlp_tol; a.mu0 = h->mu0; a.sigma = h->sigma; a.s_min = h->s_min;
    a.gamma = h->gamma; a.tau_min = h->tau_min;
    a.ws = h->ws;
    a.skip = h->skip_mask ? h->skip : nullptr; a.skip_mask = h->skip_mask;
    return a;
}

int launch_wb(Handle* h, nmpc::wb::WbArgs a, hipStream_t st) {
    if (h->line_search) return fail(h, NMPC_E_ARG, "the whole-body model takes full steps (line_search = 0)");
    // the linearisation reads rows of 42, 30, 90 / 66 floats in 8 B pieces and parameter rows in 16 B pieces (include/nmpc.h)
    auto misaligned = [](const void* p, unsigned m) { return (reinterpret_cast<uintptr_t>(p) & (m - 1u)) != 0; };
    if (misaligned(a.x0, 8) || misaligned(a.yref, 8) || misaligned(a.yref_e, 8) || misaligned(a.X, 8) || misaligned(a.U, 8) || misaligned(a.params, 16))
        return fail(h, NMPC_E_ARG, "whole-body arrays must be 8 B aligned (params: 16 B)");
    const nmpc::wb::WbLds L(a.N);
    const size_t bytes = (size_t)L.total * sizeof(float);
    if (bytes > 64 * 1024)
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&nmpc::wb::nmpc_wb_qp_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const long long nthreads = (long long)a.B * (a.N + 1);
    const unsigned lin_blocks = (unsigned)((nthreads + 63) / 64);
    const int shift = a.shift;
    for (int it = 0; it < a.max_sqp; ++it) {
        a.it = it;
        a.shift = (it == 0) ? shift : 0;
        hipLaunchKernelGGL(nmpc::wb::nmpc_wb_linearize_kernel, dim3(lin_blocks), dim3(64), 0, st, a);
        hipLaunchKernelGGL(nmpc::wb::nmpc_wb_qp_kernel, dim3(a.B), dim3(64), bytes, st, a);
    }
    HIP_TRY(h, hipGetLastError());
    return NMPC_OK;
}

// One SQP iteration = linearise (thread per stage) + QP/step (wave per problem).  Problems that
// finish early (converged, NaN, QP failure) set their workspace flag and later launches skip them.
// Two variants of the QP kernel (nmpc_solve.hip, Lds).  Resident: stage arrays in the LDS (39.6 KB at N = 50: four waves per CU),
// pinned to one wave per SIMD -- the choice while the batch fits that many waves (B <= 4 x CUs: 2.2 M solves/s at B = 1024).
// Lean: stage arrays in the workspace, 17.7 KB, two waves per SIMD -- the choice for larger batches, where the second wave fills
// the first one's dependency stalls (2.7 M at B = 8192 against 2.3 M resident), for horizons whose resident layout does not
// fit the LDS, and NMPC_QP_VARIANT=lean.  Both run the same arithmetic in the same order (bit-identical results, tested).
template <class M, bool LEAN, bool BF16B, bool ALLV>
int launch_qp(Handle* h, nmpc::SolveArgs a, hipStream_t st, unsigned lin_blocks) {
    const nmpc::Lds<M, LEAN> L(a.N);
    const size_t bytes = (size_t)L.total * sizeof(float);
    if (bytes > 160 * 1024) return fail(h, NMPC_E_ARG, "horizon too long for the LDS-resident layout");
    if (bytes > 64 * 1024)
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&nmpc::nmpc_qp_kernel<M, LEAN, BF16B, ALLV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    const int shift = a.shift;
    for (int it = 0; it < a.max_sqp; ++it) {
        a.it = it;
        a.shift = (it == 0) ? shift : 0;     // later iterations read their own iterate
        if (!nmpc::qp_linearizes_itself(LEAN, a.N))
            hipLaunchKernelGGL(nmpc::nmpc_linearize_kernel<M>, dim3(lin_blocks), dim3(64), 0, st, a);
        hipLaunchKernelGGL((nmpc::nmpc_qp_kernel<M, LEAN, BF16B, ALLV>), dim3(a.B), dim3(64), bytes, st, a);
    }
    HIP_TRY(h, hipGetLastError());
    return NMPC_OK;
}

template <class M>
int launch_solve(Handle* h, nmpc::SolveArgs a, hipStream_t st) {
    if (a.N > 64 * nmpc::N_LANE_STAGES) return fail(h, NMPC_E_ARG, "horizon too long for the lane = stage phases");
    const long long nthreads = (long long)a.B * (a.N + 1);
    const unsigned lin_blocks = (unsigned)((nthreads + 63) / 64);
    // problems in flight: a CU holds as many waves as its 160 KB of LDS take, at most one (resident) or two (lean) per SIMD
    auto in_flight = [&](size_t lds_bytes, long long per_simd) -> long long {
        if (lds_bytes > 160 * 1024) return 0;
        const long long by_lds = (long long)((160 * 1024) / lds_bytes);
        return (long long)h->n_cu * (by_lds < 4 * per_simd ? by_lds : 4 * per_simd);
    };
    const long long resident_waves = in_flight((size_t)nmpc::Lds<M, false>(a.N).total * sizeof(float), 1);
    const long long lean_waves = in_flight((size_t)nmpc::Lds<M, true>(a.N).total * sizeof(float), 2);
    const bool lean = h->force_variant ? (h->force_variant > 1)
                                       : (resident_waves == 0 || ((long long)a.B > resident_waves && lean_waves > resident_waves));
    // all static variants only where the model has more than its short list and the caller asked for them
    if constexpr (M::N_STATIC_MASKS > 4) {
        if (h->all_patterns) {
            if (h->dims.precision == 1)
                return lean ? launch_qp<M, true, true, true>(h, a, st, lin_blocks) : launch_qp<M, false, true, true>(h, a, st, lin_blocks);
            return lean ? launch_qp<M, true, false, true>(h, a, st, lin_blocks) : launch_qp<M, false, false, true>(h, a, st, lin_blocks);
        }
    }
    if (h->dims.precision == 1)
        return lean ? launch_qp<M, true, true, false>(h, a, st, lin_blocks) : launch_qp<M, false, true, false>(h, a, st, lin_blocks);
    return lean ? launch_qp<M, true, false, false>(h, a, st, lin_blocks) : launch_qp<M, false, false, false>(h, a, st, lin_blocks);
}

template <class M>
int read_tile(Handle* h, int b, int k, int which, float* out) {
    using G = nmpc::TileGeom<M>;
    const nmpc::WsLayout<M> wl(h->dims.N);
    const size_t off[4] = {wl.At + (size_t)k * G::A_FLOATS, wl.Bt + (size_t)k * G::B_FLOATS,
                           wl.Kt + (size_t)k * G::K_FLOATS, wl.Ct + (size_t)k * G::C_FLOATS};
    const size_t len[4] = {G::A_FLOATS, G::B_FLOATS, G::K_FLOATS, G::C_FLOATS};
    float img[512];
    HIP_TRY(h, hipMemcpy(img, h->ws + (size_t)b * wl.stride + off[which], len[which] * sizeof(float),
                         hipMemcpyDeviceToHost));
    std::memset(out, 0, 256 * sizeof(float));
    if (which == 0) {          // A~ : column-major, stride SA, columns 0..nx ; row nx = e_nx
        for (int c = 0; c <= M::NX; ++c)
            for (int i = 0; i < M::NX; ++i) out[i * 16 + c] = img[c * G::SA + i];
        out[M::NX * 16 + M::NX] = 1.0f;
    } else if (which == 1) {   // B~
        for (int c = 0; c < M::NU; ++c)
            for (int i = 0; i < M::NX; ++i) out[i * 16 + c] = img[c * G::SA + i];
    } else {                   // K~ (nu rows) / Acl~ (nx rows): row-major, 16 floats per row, columns by slot
        const int rows = which == 2 ? M::NU : M::NX;
        for (int i = 0; i < rows; ++i) {
            for (int c = 0; c < M::NX; ++c) out[i * 16 + c] = img[i * 16 + nmpc::slot_of(c)];
            out[i * 16 + M::NX] = img[i * 16 + nmpc::HS];
        }
        if (which == 3) out[M::NX * 16 + M::NX] = 1.0f;
    }
    return NMPC_OK;
}
}  // namespace

extern "C" {

int nmpc_model_dims(int model_id, int* nx, int* nu, int* np, int* ng) {
    int d[4];
    if (model_id == NMPC_MODEL_DOUBLE_INTEGRATOR) {
        d[0] = nmpc::DoubleIntegrator::NX; d[1] = nmpc::DoubleIntegrator::NU;
        d[2] = nmpc::DoubleIntegrator::NP; d[3] = nmpc::DoubleIntegrator::NG;
    } else if (model_id == NMPC_MODEL_CENTROIDAL) {
        d[0] = nmpc::Centroidal::NX; d[1] = nmpc::Centroidal::NU;
        d[2] = nmpc::Centroidal::NP; d[3] = nmpc::Centroidal::NG;
    } else if (model_id == NMPC_MODEL_WHOLEBODY) {
        d[0] = nmpc::wb::NX; d[1] = nmpc::wb::NU; d[2] = nmpc::wb::NP; d[3] = nmpc::wb::NG;
    } else {
        return NMPC_E_ARG;
    }
    if (nx) *nx = d[0];
    if (nu) *nu = d[1];
    if (np) *np = d[2];
    if (ng) *ng = d[3];
    return NMPC_OK;
}

int nmpc_model_output_dims(int model_id, int* ny, int* ny_e) {
    int nx, nu;
    if (nmpc_model_dims(model_id, &nx, &nu, nullptr, nullptr)) return NMPC_E_ARG;
    const bool wbm = model_id == NMPC_MODEL_WHOLEBODY;
    if (ny) *ny = wbm ? nmpc::wb::NY : nx + nu;
    if (ny_e) *ny_e = wbm ? nmpc::wb::NYE : nx;
    return NMPC_OK;
}

int nmpc_create(const nmpc_dims* dims, int device_id, void** handle) {
    if (!dims || !handle) return fail(nullptr, NMPC_E_ARG, "null argument");
    *handle = nullptr;
    int nx, nu, np, ng;
    if (nmpc_model_dims(dims->model_id, &nx, &nu, &np, &ng)) return fail(nullptr, NMPC_E_ARG, "unknown model_id");
    if (dims->N < 1 || dims->B_max < 1) return fail(nullptr, NMPC_E_ARG, "N and B_max must be positive");
    if (dims->precision < 0 || dims->precision > 3 || (dims->precision >= 2 && dims->model_id != NMPC_MODEL_WHOLEBODY))
        return fail(nullptr, NMPC_E_ARG, "precision must be 0 (fp32), 1 (bf16 contraction) or, whole-body only, 2 (split bf16) / 3 (three-way split bf16)");
    Handle* h = new Handle();
    h->dims = *dims;
    h->device = device_id;
    h->nx = nx; h->nu = nu; h->np = np; h->ng = ng;
    nmpc_model_output_dims(dims->model_id, &h->ny, &h->nye);
    if (dims->model_id == NMPC_MODEL_WHOLEBODY && dims->N > 64)
        { delete h; return fail(nullptr, NMPC_E_ARG, "the whole-body model needs N <= 64 (lane = stage phases)"); }
    nmpc::DeviceGuard guard(device_id);
    hipError_t e = guard.err;
    if (e == hipSuccess) e = hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, device_id);
    if (const char* v = std::getenv("NMPC_QP_VARIANT")) {
        if (!std::strcmp(v, "resident")) h->force_variant = 1;
        else if (!std::strcmp(v, "lean")) h->force_variant = 2;
    }
    if (e == hipSuccess) {
        h->ws_stride = (dims->model_id == NMPC_MODEL_DOUBLE_INTEGRATOR) ? ws_floats_per_problem<nmpc::DoubleIntegrator>(dims->N)
                     : (dims->model_id == NMPC_MODEL_CENTROIDAL)      ? ws_floats_per_problem<nmpc::Centroidal>(dims->N)
                                                                      : nmpc::wb::WsLayout(dims->N).stride;
        h->ws_bytes = (size_t)dims->B_max * h->ws_stride * sizeof(float);
        e = hipMalloc(reinterpret_cast<void**>(&h->ws), h->ws_bytes);
    }
    if (e == hipSuccess) e = hipMemset(h->ws, 0, h->ws_bytes);
    if (e == hipSuccess) {
        const size_t per = (size_t)dims->N * h->ny + h->nye + (size_t)(dims->N + 1) * (np > 0 ? np : 1) + nx;   // yref, yref_e, params, x0 of the rollouts
        e = hipMalloc(reinterpret_cast<void**>(&h->roll), ((size_t)dims->B_max * per + 16) * sizeof(float));   // (+ the 16 B roundings of the carve-up)
    }
    if (e != hipSuccess) {
        g_create_error = std::string("nmpc_create: ") + hipGetErrorString(e);
        if (h->ws) (void)hipFree(h->ws);
        if (h->roll) (void)hipFree(h->roll);
        delete h;
        return NMPC_E_HIP;
    }
    *handle = h;
    return NMPC_OK;
}

void nmpc_destroy(void* handle) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return;
    nmpc::DeviceGuard guard(h->device);
    if (h->ws) (void)hipFree(h->ws);
    if (h->roll) (void)hipFree(h->roll);
    delete h;
}

const char* nmpc_last_error(void* handle) {
    Handle* h = static_cast<Handle*>(handle);
    return h ? h->err.c_str() : g_create_error.c_str();
}

size_t nmpc_workspace_bytes(void* handle) {
    Handle* h = static_cast<Handle*>(handle);
    return h ? h->ws_bytes : 0;
}

int nmpc_set_model_params(void* handle, const float* mp, int count) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !mp) return fail(h, NMPC_E_ARG, "null argument");
    if (count != NMPC_MP_COUNT) return fail(h, NMPC_E_ARG, "model parameter vector must have NMPC_MP_COUNT entries");
    if (!(mp[NMPC_MP_DT] > 0.0f)) return fail(h, NMPC_E_ARG, "dt must be positive");
    if (h->dims.model_id != NMPC_MODEL_DOUBLE_INTEGRATOR &&
        !(mp[NMPC_MP_MASS] > 0 && mp[NMPC_MP_IXX] > 0 && mp[NMPC_MP_IYY] > 0 && mp[NMPC_MP_IZZ] > 0))
        return fail(h, NMPC_E_ARG, "mass and inertia must be positive");
    if (h->dims.model_id == NMPC_MODEL_WHOLEBODY && !(mp[NMPC_MP_L1] > 0 && mp[NMPC_MP_L2] > 0 && mp[NMPC_MP_PGAIN] >= 0))
        return fail(h, NMPC_E_ARG, "link lengths must be positive, p_gain non-negative");
    h->mp = nmpc::ModelParams{mp[0], mp[1], mp[2], mp[3], mp[4], mp[5], mp[6], mp[7],
                              mp[8], mp[9], mp[10], mp[11], mp[12], mp[13], mp[14], mp[15]};
    h->mp_set = true;
    return NMPC_OK;
}

int nmpc_set_weights(void* handle, const float* W, const float* W_e, float reg, float reg_e) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !W || !W_e) return fail(h, NMPC_E_ARG, "null argument");
    for (int i = 0; i < h->ny; ++i) {
        if (!(W[i] >= 0.0f)) return fail(h, NMPC_E_ARG, "weights must be non-negative");
        h->W[i] = W[i];
    }
    for (int i = 0; i < h->nye; ++i) {
        if (!(W_e[i] >= 0.0f)) return fail(h, NMPC_E_ARG, "weights must be non-negative");
        h->We[i] = W_e[i];
    }
    if (!(reg >= 0.0f) || !(reg_e >= 0.0f)) return fail(h, NMPC_E_ARG, "regularisation must be non-negative");
    h->reg = reg; h->reg_e = reg_e;
    h->w_set = true;
    if (h->dims.model_id == NMPC_MODEL_WHOLEBODY) {
        bool pos = false;
        for (int i = 0; i < 8; ++i) pos = pos || W[nmpc::wb::RY_POS + i] > 0.0f || W_e[nmpc::wb::RE_POS + i] > 0.0f;
        if (h->pos_rows && !pos) h->ws_dirty = true;     // the foot-placement rows of the Js images go back to exact zeros
        h->pos_rows = pos;
    }
    return NMPC_OK;
}

int nmpc_set_opts(void* handle, int max_sqp_iter, int max_qp_iter, float nlp_tol, float qp_tol,
                  int line_search) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return NMPC_E_ARG;
    if (max_sqp_iter < 1 || max_qp_iter < 0) return fail(h, NMPC_E_ARG, "iteration counts out of range");
    h->max_sqp = max_sqp_iter; h->n_ipm = max_qp_iter;
    h->nlp_tol = nlp_tol; h->qp_tol = qp_tol; h->line_search = line_search ? 1 : 0;
    return NMPC_OK;
}

int nmpc_set_contact_patterns(void* handle, int all_patterns) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return NMPC_E_ARG;
    h->all_patterns = all_patterns ? 1 : 0;
    return NMPC_OK;
}

int nmpc_set_skip(void* handle, const int* flags, int mask) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return NMPC_E_ARG;
    h->skip = flags;
    h->skip_mask = flags ? mask : 0;
    return NMPC_OK;
}

int nmpc_set_ipm(void* handle, float mu0, float sigma, float s_min, float gamma, float tau_min,
                 float merit_rho) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return NMPC_E_ARG;
    if (!(mu0 > 0 && sigma > 0 && sigma < 1 && s_min > 0 && gamma > 0 && gamma < 1 && tau_min >= 0 && merit_rho >= 0))
        return fail(h, NMPC_E_ARG, "interior-point constants out of range");
    h->mu0 = mu0; h->sigma = sigma; h->s_min = s_min; h->gamma = gamma; h->tau_min = tau_min; h->rho = merit_rho;
    return NMPC_OK;
}

int nmpc_shift_warm_start(void* handle, int B, int shift, float* X, float* U, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;
    if (!X || !U) return fail(h, NMPC_E_ARG, "null argument");
    if (B < 0 || B > h->dims.B_max) return fail(h, NMPC_E_ARG, "B exceeds B_max");
    if (shift < 0) return fail(h, NMPC_E_ARG, "negative shift");
    if (shift == 0) return NMPC_OK;
    const int N = h->dims.N;
    if (shift > N) shift = N;
    if ((size_t)N * h->nx > 256 * 16 || (size_t)N * h->nu > 256 * 16)
        return fail(h, NMPC_E_ARG, "trajectory too long for the shift kernel");
    nmpc::DeviceGuard guard(h->device);
    HIP_TRY(h, guard.err);
    hipLaunchKernelGGL(nmpc::nmpc_shift_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream),
                       N, h->nx, h->nu, h->dims.model_id == NMPC_MODEL_WHOLEBODY ? nmpc::wb::WF : 0, shift, X, U);
    HIP_TRY(h, hipGetLastError());
    return NMPC_OK;
}

int nmpc_shift_solve_batch(void* handle, int B, int shift, const float* x0, const float* yref,
                           int yref_per_stage, const float* yref_e, const float* params, float* X, float* U,
                           int* status, float* stats, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;     // an empty batch is a no-op (its tensors have no storage)
    if (!x0 || !yref || !yref_e || !X || !U || (h->np > 0 && !params)) return fail(h, NMPC_E_ARG, "null argument");
    if (B < 0 || B > h->dims.B_max) return fail(h, NMPC_E_ARG, "B exceeds B_max");
    if (shift < 0) return fail(h, NMPC_E_ARG, "negative shift");
    if (!h->mp_set || !h->w_set) return fail(h, NMPC_E_STATE, "model parameters / weights not set");
    hipStream_t st = static_cast<hipStream_t>(stream);
    nmpc::DeviceGuard guard(h->device);
    HIP_TRY(h, guard.err);
    if (h->ws_dirty) {
        HIP_TRY(h, hipMemsetAsync(h->ws, 0, h->ws_bytes, st));
        h->ws_dirty = false;
    }
    if (h->dims.model_id == NMPC_MODEL_WHOLEBODY) {
        nmpc::wb::WbArgs w = wb_args(h);
        w.B = B;
        w.shift = shift > h->dims.N ? h->dims.N : shift;
        w.yref_per_stage = yref_per_stage ? 1 : 0;
        w.x0 = x0; w.yref = yref; w.yref_e = yref_e; w.params = params;
        w.X = X; w.U = U; w.status = status; w.stats = stats;
        return launch_wb(h, w, st);
    }
    nmpc::SolveArgs a = base_args(h);
    a.B = B;
    a.shift = shift > h->dims.N ? h->dims.N : shift;
    a.yref_per_stage = yref_per_stage ? 1 : 0;
    a.x0 = x0; a.yref = yref; a.yref_e = yref_e; a.params = params ? params : x0;
    a.X = X; a.U = U; a.status = status; a.stats = stats;
    if (h->dims.model_id == NMPC_MODEL_DOUBLE_INTEGRATOR) return launch_solve<nmpc::DoubleIntegrator>(h, a, st);
    return launch_solve<nmpc::Centroidal>(h, a, st);
}

int nmpc_solve_batch(void* handle, int B, const float* x0, const float* yref, int yref_per_stage,
                     const float* yref_e, const float* params, float* X, float* U, int* status,
                     float* stats, void* stream) {
    return nmpc_shift_solve_batch(handle, B, 0, x0, yref, yref_per_stage, yref_e, params, X, U, status, stats, stream);
}

int nmpc_riccati_batch(void* handle, int Bsz, int nx, int nu, const float* Q, const float* R,
                       const float* q, const float* r, const float* A, const float* B_,
                       const float* d, const float* dx0, float* dX, float* dU, int* status,
                       void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return NMPC_E_ARG;
    if (!Q || !R || !q || !r || !A || !B_ || !d || !dx0 || !dX || !dU) return fail(h, NMPC_E_ARG, "null argument");
    if (nx < 1 || nx > 15 || nu < 1 || nu > 16) return fail(h, NMPC_E_ARG, "need 1 <= nx <= 15, 1 <= nu <= 16");
    if (h->dims.model_id == NMPC_MODEL_WHOLEBODY) return fail(h, NMPC_E_ARG, "nmpc_riccati_batch needs a handle of the tile-family models");
    if (Bsz < 0 || Bsz > h->dims.B_max) return fail(h, NMPC_E_ARG, "B exceeds B_max");
    if (Bsz == 0) return NMPC_OK;
    nmpc::DeviceGuard guard(h->device);
    HIP_TRY(h, guard.err);
    nmpc::RiccatiArgs a{h->dims.N, Bsz, nx, nu, Q, R, q, r, A, B_, d, dx0, dX, dU, status, h->ws};
    h->ws_dirty = true;
    hipLaunchKernelGGL(nmpc::nmpc_riccati_kernel, dim3(Bsz), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    HIP_TRY(h, hipGetLastError());
    return NMPC_OK;
}

int nmpc_tracking_error(void* handle, int B, int T, int ns, const float* S, const float* S_nom,
                        float* err, float* weight, float threshold, float ood_weight, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (B == 0) return NMPC_OK;
    if (!S || !S_nom || !err) return fail(h, NMPC_E_ARG, "null argument");
    if (B < 0 || T < 1 || ns < 2) return fail(h, NMPC_E_ARG, "need B >= 0, T >= 1, ns >= 2");
    const size_t lds = (size_t)nmpc::TRB * (ns | 1) * sizeof(float);
    if (lds > 64 * 1024) return fail(h, NMPC_E_ARG, "state dimension too large for the staging tile");
    nmpc::DeviceGuard guard(nmpc::device_of(S));
    const long long rows = (long long)B * T;
    const long long blocks = (rows + nmpc::TRB - 1) / nmpc::TRB;
    if (blocks > 0x7fffffffLL) return fail(h, NMPC_E_ARG, "too many rows");
    hipLaunchKernelGGL(nmpc::nmpc_tracking_error_kernel, dim3((unsigned)blocks), dim3(nmpc::TRB), lds,
                       static_cast<hipStream_t>(stream), rows, T, ns, S, S_nom, err, weight, threshold,
                       ood_weight);
    HIP_TRY(h, hipGetLastError());
    return NMPC_OK;
}

int nmpc_rollout_batch(void* handle, int B, const nmpc_rollout_cfg* cfg, const signed char* gait, float* x,
                       const double* v_des, const double* w_des, double* ref_state, float* foot_pos,
                       const float* push_force, const float* phase, float* X, float* U, float* S,
                       int* status, int* failed, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;
    if (!cfg || !gait || !x || !v_des || !w_des || !ref_state || !foot_pos || !phase || !X || !U || !S || !status || !failed)
        return fail(h, NMPC_E_ARG, "null argument");
    if (h->dims.model_id != NMPC_MODEL_CENTROIDAL) return fail(h, NMPC_E_ARG, "rollouts need the centroidal model");
    if (B < 0 || B > h->dims.B_max) return fail(h, NMPC_E_ARG, "B exceeds B_max");
    if (h->dims.N > 128) return fail(h, NMPC_E_ARG, "rollouts need N <= 128");
    if (cfg->n_replans < 1 || cfg->nodes_per_replan < 1 || cfg->nodes_per_replan > h->dims.N ||
        cfg->replanning_steps < 1 || cfg->nodes_per_cycle < 1 || cfg->start_node < 0)
        return fail(h, NMPC_E_ARG, "rollout configuration out of range");
    if (!h->mp_set || !h->w_set) return fail(h, NMPC_E_STATE, "model parameters / weights not set");
    if (cfg->footsteps && !(cfg->nominal_period > 0.0f)) return fail(h, NMPC_E_ARG, "footsteps need the gait period");
    if (cfg->record_sim_steps &&
        std::fabs(cfg->nodes_per_replan * (cfg->time_horizon / h->dims.N) - cfg->replanning_steps * cfg->sim_dt) > 1e-9)
        return fail(h, NMPC_E_ARG, "per-step recording needs nodes_per_replan * dt_nodes = replanning_steps * sim_dt");
    hipStream_t st = static_cast<hipStream_t>(stream);
    nmpc::DeviceGuard guard(h->device);
    HIP_TRY(h, guard.err);
    if (h->ws_dirty) {
        HIP_TRY(h, hipMemsetAsync(h->ws, 0, h->ws_bytes, st));
        h->ws_dirty = false;
    }
    const int N = h->dims.N;
    float* yref = h->roll;
    float* yref_e = yref + (size_t)h->dims.B_max * N * 24;
    float* params = yref_e + (size_t)h->dims.B_max * 12;
    nmpc::RolloutArgs r{};
    r.B = B; r.N = N; r.npc = cfg->nodes_per_cycle;
    r.nodes_per_replan = cfg->nodes_per_replan; r.replanning_steps = cfg->replanning_steps;
    r.n_replans = cfg->n_replans;
    r.sim_dt = cfg->sim_dt; r.t_horizon = cfg->time_horizon; r.nom_height = cfg->nom_height;
    r.height_offset = cfg->height_offset;
    r.mass = h->mp.mass; r.gz = h->mp.gz;
    r.gait = gait; r.x = x; r.v_des = v_des; r.w_des = w_des; r.ref_state = ref_state; r.foot_pos = foot_pos;
    r.push_force = push_force; r.yref = yref; r.yref_e = yref_e; r.params = params;
    r.X = X; r.U = U; r.S = S; r.status = status; r.failed = failed;
    r.footsteps = cfg->footsteps ? 1 : 0; r.record_sim_steps = cfg->record_sim_steps ? 1 : 0;
    r.term_mask = cfg->terminate_mask & NMPC_ROLLOUT_FLAG_MASK; r.collision_height = cfg->collision_height;
    std::memcpy(r.hip_offset, cfg->hip_offset, sizeof(r.hip_offset));
    std::memcpy(r.stance_ratio, cfg->stance_ratio, sizeof(r.stance_ratio));
    r.nominal_period = cfg->nominal_period; r.foot_size = cfg->foot_size;
    r.dt_nodes = cfg->time_horizon / N;
    const int rows_per_replan = r.record_sim_steps ? cfg->replanning_steps : 1;
    r.n_rows = cfg->n_replans * rows_per_replan;
    nmpc::SolveArgs a = base_args(h);
    a.B = B; a.yref_per_stage = 1;
    a.x0 = x; a.yref = yref; a.yref_e = yref_e; a.params = params; a.X = X; a.U = U;
    a.status = status; a.stats = nullptr;
    a.skip = r.term_mask ? failed : nullptr; a.skip_mask = r.term_mask;     // terminated rollouts cost no solve
    const float dt_replan = (float)(cfg->replanning_steps * cfg->sim_dt);
    for (int i = 0; i < cfg->n_replans; ++i) {
        const bool cold = cfg->first_solve && i == 0;
        r.node = cfg->start_node + i * cfg->nodes_per_replan;
        r.first = cold ? 1 : 0;
        r.replan_index = i;
        r.row0 = i * rows_per_replan;
        r.phase = phase[i];
        // the push window in replanning intervals, half a simulation step of slack on both ends: with plain float compares
        // 5 * 0.04f = 0.19999999 misses a window that starts at 0.2 (the host loop, in doubles, does not)
        const double t_now = i * (cfg->replanning_steps * cfg->sim_dt), slack = 0.5 * cfg->sim_dt;
        r.push_dt = (push_force && cfg->push_duration > 0.0f && t_now >= (double)cfg->push_start - slack &&
                     t_now < (double)cfg->push_start + (double)cfg->push_duration - slack) ? dt_replan : 0.0f;
        hipLaunchKernelGGL(nmpc::nmpc_rollout_prepare_kernel, dim3(B), dim3(64), 0, st, r);
        a.shift = cold ? 0 : (cfg->nodes_per_replan > N ? N : cfg->nodes_per_replan);   // warm start folded into the solve
        a.max_sqp = cold ? cfg->max_sqp_first : h->max_sqp;
        a.nlp_tol = cold ? cfg->nlp_tol_first : cfg->nlp_tol;
        const int rc = launch_solve<nmpc::Centroidal>(h, a, st);
        if (rc) return rc;
        hipLaunchKernelGGL(nmpc::nmpc_rollout_advance_kernel, dim3((B + 63) / 64), dim3(64), 0, st, r);
    }
    HIP_TRY(h, hipGetLastError());
    return NMPC_OK;
}

int nmpc_wb_rollout_batch(void* handle, int B, const nmpc_wb_rollout_cfg* cfg, const signed char* gait, const signed char* peaks,
                          const int* nodes, float* q, float* v, const double* v_des, const double* w_des, double* ref_state,
                          const float* joint_ref, const float* push_force, float* X, float* U, float* S, int* status,
                          int* failed, void* stream) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;
    if (!cfg || !gait || !peaks || !nodes || !q || !v || !v_des || !w_des || !ref_state || !joint_ref || !X || !U || !S || !status || !failed)
        return fail(h, NMPC_E_ARG, "null argument");
    if (h->dims.model_id != NMPC_MODEL_WHOLEBODY) return fail(h, NMPC_E_ARG, "nmpc_wb_rollout_batch needs the whole-body model");
    if (B < 0 || B > h->dims.B_max) return fail(h, NMPC_E_ARG, "B exceeds B_max");
    if (cfg->n_replans < 1 || cfg->replanning_steps < 1 || cfg->nodes_per_cycle < 1 || !(cfg->sim_dt > 0) || !(cfg->time_horizon > 0))
        return fail(h, NMPC_E_ARG, "rollout configuration out of range");
    if (!h->mp_set || !h->w_set) return fail(h, NMPC_E_STATE, "model parameters / weights not set");
    if (h->line_search) return fail(h, NMPC_E_ARG, "the whole-body model takes full steps (line_search = 0)");
    const int N = h->dims.N;
    if (cfg->replanning_steps * cfg->sim_dt > cfg->time_horizon) return fail(h, NMPC_E_ARG, "replanning interval longer than the horizon");
    for (int i = 0; i < cfg->n_replans; ++i)
        if (nodes[i] < 0 || (i > 0 && nodes[i] < nodes[i - 1])) return fail(h, NMPC_E_ARG, "nodes must be non-negative and non-decreasing");
    hipStream_t st = static_cast<hipStream_t>(stream);
    nmpc::DeviceGuard guard(h->device);
    HIP_TRY(h, guard.err);
    if (h->ws_dirty) {
        HIP_TRY(h, hipMemsetAsync(h->ws, 0, h->ws_bytes, st));
        h->ws_dirty = false;
    }
    auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };       // every array starts on a 16 B boundary (launch_wb checks)
    float* yref = h->roll;
    float* yref_e = yref + up4((size_t)h->dims.B_max * N * h->ny);
    float* params = yref_e + up4((size_t)h->dims.B_max * h->nye);
    float* x0 = params + up4((size_t)h->dims.B_max * (N + 1) * h->np);
    nmpc::wb::WbRolloutArgs r{};
    r.B = B; r.N = N; r.npc = cfg->nodes_per_cycle; r.replanning_steps = cfg->replanning_steps; r.n_replans = cfg->n_replans;
    r.record_sim_steps = cfg->record_sim_steps ? 1 : 0; r.force_gravity = cfg->force_reference_gravity ? 1 : 0;
    r.sim_dt = cfg->sim_dt; r.t_horizon = cfg->time_horizon; r.nom_height = cfg->nom_height; r.height_offset = cfg->height_offset;
    r.dt_nodes = cfg->time_horizon / N;
    r.step_height = cfg->step_height; r.nominal_period = cfg->nominal_period; r.collision_height = cfg->collision_height;
    r.term_mask = cfg->terminate_mask & NMPC_ROLLOUT_FLAG_MASK;
    r.mp = h->mp;
    r.gait = gait; r.peaks = peaks; r.q = q; r.v = v; r.v_des = v_des; r.w_des = w_des; r.ref_state = ref_state;
    r.joint_ref = joint_ref; r.push_force = push_force;
    r.x0 = x0; r.yref = yref; r.yref_e = yref_e; r.params = params; r.X = X; r.U = U; r.S = S; r.status = status; r.failed = failed;
    const int rows_per_replan = r.record_sim_steps ? cfg->replanning_steps : 1;
    r.n_rows = cfg->n_replans * rows_per_replan;
    nmpc::wb::WbArgs w = wb_args(h);
    w.B = B; w.yref_per_stage = 1;
    w.x0 = x0; w.yref = yref; w.yref_e = yref_e; w.params = params; w.X = X; w.U = U; w.status = status; w.stats = nullptr;
    w.skip = r.term_mask ? failed : nullptr; w.skip_mask = r.term_mask;
    const double dt_replan = cfg->replanning_steps * cfg->sim_dt, slack = 0.5 * cfg->sim_dt;
    int last_node = cfg->last_node;
    for (int i = 0; i < cfg->n_replans; ++i) {
        const bool cold = cfg->first_solve && i == 0;
        r.node = nodes[i];
        r.first = cold ? 1 : 0;
        r.replan_index = i;
        r.row0 = i * rows_per_replan;
        const double t_now = i * dt_replan;
        r.push_dt = (push_force && cfg->push_duration > 0.0f && t_now >= (double)cfg->push_start - slack &&
                     t_now < (double)cfg->push_start + (double)cfg->push_duration - slack) ? (float)dt_replan : 0.0f;
        hipLaunchKernelGGL(nmpc::wb::nmpc_wb_rollout_prepare_kernel, dim3(B), dim3(64), 0, st, r);
        int shift = cold ? 0 : nodes[i] - last_node;          // warm_start_solver(i_node): start_node = i_node - last_node (solver.py:304-309)
        if (shift > N) shift = N;
        last_node = nodes[i];
        w.shift = shift;
        w.max_sqp = cold ? cfg->max_sqp_first : h->max_sqp;
        w.nlp_tol = cold ? cfg->nlp_tol_first : cfg->nlp_tol;
        const int rc = launch_wb(h, w, st);
        if (rc) return rc;
        hipLaunchKernelGGL(nmpc::wb::nmpc_wb_rollout_advance_kernel, dim3((B + 63) / 64), dim3(64), 0, st, r);
    }
    HIP_TRY(h, hipGetLastError());
    return NMPC_OK;
}

int nmpc_debug_set_buffer(void* handle, float* dev_buffer) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h) return NMPC_E_ARG;
    h->dbg = dev_buffer;
    return NMPC_OK;
}

int nmpc_debug_read_tile(void* handle, int b, int k, int which, float* out_host) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !out_host) return fail(h, NMPC_E_ARG, "null argument");
    if (b < 0 || b >= h->dims.B_max || k < 0 || k >= h->dims.N || which < 0 || which > 3)
        return fail(h, NMPC_E_ARG, "index out of range");
    nmpc::DeviceGuard guard(h->device);
    HIP_TRY(h, guard.err);
    HIP_TRY(h, hipDeviceSynchronize());
    if (h->dims.model_id == NMPC_MODEL_DOUBLE_INTEGRATOR) return read_tile<nmpc::DoubleIntegrator>(h, b, k, which, out_host);
    if (h->dims.model_id == NMPC_MODEL_WHOLEBODY) return fail(h, NMPC_E_ARG, "use nmpc_debug_read_workspace for the whole-body model");
    return read_tile<nmpc::Centroidal>(h, b, k, which, out_host);
}

int nmpc_debug_read_workspace(void* handle, int b, size_t offset, size_t count, float* out_host) {
    Handle* h = static_cast<Handle*>(handle);
    if (!h || !out_host) return fail(h, NMPC_E_ARG, "null argument");
    if (b < 0 || b >= h->dims.B_max || offset + count > h->ws_stride) return fail(h, NMPC_E_ARG, "index out of range");
    nmpc::DeviceGuard guard(h->device);
    HIP_TRY(h, guard.err);
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(out_host, h->ws + (size_t)b * h->ws_stride + offset, count * sizeof(float), hipMemcpyDeviceToHost));
    return NMPC_OK;
}

int nmpc_debug_wb_layout(int N, size_t* out8) {
    if (!out8 || N < 1) return NMPC_E_ARG;
    const nmpc::wb::WsLayout wl(N);
    const nmpc::wb::StageArr sa(N);
    out8[0] = wl.rec; out8[1] = wl.js; out8[2] = wl.qt; out8[3] = wl.kt; out8[4] = wl.arr; out8[5] = wl.stride;
    out8[6] = (size_t)sa.NS; out8[7] = (size_t)nmpc::wb::REC;
    return NMPC_OK;
}

}  // extern "C"
