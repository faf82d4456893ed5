// nmpc_solve.hip -- batched NMPC solve, one problem per wavefront (gfx950).
//
// Replaces the reference's per-step solve (mpc_controller/utils/solver.py:396-403 ->
// acados SQP / HPIPM) by one kernel launch per batch:
//   phase L (lane = stage)  dynamics, analytic A,B, defects, gradients, constraint values
//   phase R (serial stages)  Riccati backward on 16x16 tiles: MFMA products + readlane LDL'
//   phase F (serial stages)  rollout of the affine feedback law (MFMA mat-vec)
//   phase I (lane = stage)  interior-point step: slacks, multipliers, fraction to boundary
//   phase S (lane = stage)  step (optionally l1-merit backtracking), write back
// Stage matrices A~,B~,K~ live in an HBM/L2 workspace as 1 KiB column-major tile images;
// trajectories, gradients and IPM state of the problem live in LDS (~32 KB at N=50).
#include <hip/hip_runtime.h>

#include "../../include/nmpc.h"
#include "nmpc_models.hpp"
#include "nmpc_sweep.hpp"

namespace nmpc {

struct SolveArgs {
    ModelParams mp;
    float W[32];    // stage weights [nx+nu]
    float We[16];   // terminal weights [nx]
    float reg, reg_e;
    int N, B;
    int max_sqp, n_ipm, line_search, yref_per_stage;
    float nlp_tol, mu0, sigma, s_min, gamma, tau_min, rho;
    const float* x0;
    const float* yref;
    const float* yref_e;
    const float* params;
    float* X;
    float* U;
    int* status;
    float* stats;
    float* ws;            // workspace: per problem 3*N tiles (A~, B~, K~)
    float* dbg;           // diagnostic builds: [B][8] phase cycle counts, else unused
};

__host__ __device__ inline int round4(int n) { return (n + 3) & ~3; }

// Diagnostic build only (-DNMPC_STAMPS, tools/phase_shares.py): per-phase cycle counters written to
// a buffer of their own; the production kernel contains no stamp.
#ifdef NMPC_STAMPS
#define STAMP_DECL unsigned long long st_t0 = __builtin_readcyclecounter(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i) do { const unsigned long long st_t1 = __builtin_readcyclecounter(); st_acc[i] += st_t1 - st_t0; st_t0 = st_t1; } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#endif

template <class M>
struct Lds {
    // offsets in floats
    int Xs, Us, dX, dU, dXp, dUp, qv, rv, sv, lv, cv, act, conv, total;
    __host__ __device__ explicit Lds(int N) {
        int o = 0;
        Xs = o;  o += round4((N + 1) * M::NX);
        Us = o;  o += round4(N * M::NU);
        dX = o;  o += round4((N + 1) * M::NX);
        dU = o;  o += round4(N * M::NU);
        dXp = o; o += round4((N + 1) * M::NX);
        dUp = o; o += round4(N * M::NU);
        qv = o;  o += round4((N + 1) * M::NX);
        rv = o;  o += round4(N * M::NU);
        sv = o;  o += round4(N * M::NG);
        lv = o;  o += round4(N * M::NG);
        cv = o;  o += round4(N * M::NG);
        act = o; o += round4(N);
        conv = o; o += 2 * CTILE;
        total = o;
    }
};

// write one column (rows 0..15) of a column-major tile image
__device__ __forceinline__ void store_col16(float* tile, int c, const float (&v)[16], int nquads) {
    f32x4* p = reinterpret_cast<f32x4*>(tile + c * TS);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nquads) p[i] = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
}

template <class M>
__global__ __launch_bounds__(64) void nmpc_solve_kernel(const SolveArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NG = M::NG, NY = NX + NU;
    constexpr int NQ = (NX + 1 + 3) / 4;   // float4 per stored column (rows 0..NX)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b >= a.B) return;
    const int lane = lane_id();
    const int q4 = lane >> 4, c = lane & 15;
    const int N = a.N;
    const Lds<M> L(N);
    float* Xs = smem + L.Xs;   float* Us = smem + L.Us;
    float* dX = smem + L.dX;   float* dU = smem + L.dU;
    float* dXp = smem + L.dXp; float* dUp = smem + L.dUp;
    float* qv = smem + L.qv;   float* rv = smem + L.rv;
    float* sv = smem + L.sv;   float* lv = smem + L.lv;   float* cv = smem + L.cv;
    unsigned* actm = reinterpret_cast<unsigned*>(smem + L.act);
    float* conv = smem + L.conv;

    float* Xg = a.X + (size_t)b * (N + 1) * NX;
    float* Ug = a.U + (size_t)b * N * NU;
    const float* pg = a.params + (size_t)b * (N + 1) * NP;
    const float* yr = a.yref + (size_t)b * (a.yref_per_stage ? (size_t)N * NY : (size_t)NY);
    const float* yre = a.yref_e + (size_t)b * NX;
    const float* x0 = a.x0 + (size_t)b * NX;
    float* At = a.ws + (size_t)b * 3 * N * TILE;
    float* Bt = At + (size_t)N * TILE;
    float* Kt = Bt + (size_t)N * TILE;

    for (int i = lane; i < (N + 1) * NX; i += 64) Xs[i] = Xg[i];
    for (int i = lane; i < N * NU; i += 64) Us[i] = Ug[i];
    // per-lane weights of "its" column
    const float wq_c = (c < NX) ? a.W[c] + a.reg : 0.0f;
    const float wr_c = (c < NU) ? a.W[NX + (c < NU ? c : 0)] + a.reg : 0.0f;
    const float we_c = (c < NX) ? a.We[c < NX ? c : 0] + a.reg_e : 0.0f;
    __syncthreads();

    int status = NMPC_STATUS_MAXITER;
    STAMP_DECL;
    float cost = 0.0f, stepn = 0.0f, alpha = 1.0f;
    int it = 0;
    for (it = 0; it < a.max_sqp; ++it) {
        // ------------------------------------------------------------ phase L: linearise
        float cost_l = 0.0f;
        int nact_l = 0;
        for (int k = lane; k < N; k += 64) {
            float x[NX], u[NU], xn[NX], p[NP > 0 ? NP : 1];
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = Xs[k * NX + i];
#pragma unroll
            for (int i = 0; i < NU; ++i) u[i] = Us[k * NU + i];
#pragma unroll
            for (int i = 0; i < NP; ++i) p[i] = pg[(size_t)k * NP + i];
            float* Atk = At + (size_t)k * TILE;
            float* Btk = Bt + (size_t)k * TILE;
            auto emitA = [&](int j, const float (&colv)[NX]) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = (i < NX) ? colv[i < NX ? i : 0] : 0.0f;
                store_col16(Atk, j, v, NQ);
            };
            auto emitB = [&](int j, const float (&colv)[NX]) {
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = (i < NX) ? colv[i < NX ? i : 0] : 0.0f;
                store_col16(Btk, j, v, NQ);
            };
            M::linearize(a.mp, x, u, p, xn, emitA, emitB);
            {   // defect column  [d; 1]
                float v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    v[i] = (i < NX) ? xn[i < NX ? i : 0] - Xs[(k + 1) * NX + (i < NX ? i : 0)]
                                    : (i == NX ? 1.0f : 0.0f);
                store_col16(Atk, NX, v, NQ);
            }
            const float* yk = yr + (a.yref_per_stage ? (size_t)k * NY : 0);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const float e = x[i] - yk[i];
                qv[k * NX + i] = a.W[i] * e;
                cost_l += 0.5f * a.W[i] * e * e;
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const float e = u[i] - yk[NX + i];
                rv[k * NU + i] = a.W[NX + i] * e;
                cost_l += 0.5f * a.W[NX + i] * e * e;
            }
            if (a.n_ipm > 0) {
                const unsigned am = M::active_mask(a.mp, p);
                actm[k] = am;
                nact_l += __popc(am);
                float g[NG];
                M::gdot(a.mp, u, g);
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const float cj = g[j] - M::h(a.mp, j);
                    const float s = fmaxf(-cj, a.s_min);
                    cv[k * NG + j] = cj;
                    sv[k * NG + j] = s;
                    lv[k * NG + j] = a.mu0 / s;
                }
            }
        }
        if (lane < NX) {
            const float e = Xs[N * NX + lane] - yre[lane];
            qv[N * NX + lane] = a.We[lane] * e;
            cost_l += 0.5f * a.We[lane] * e * e;
        }
        for (int i = lane; i < (N + 1) * NX; i += 64) dX[i] = 0.0f;
        for (int i = lane; i < N * NU; i += 64) dU[i] = 0.0f;
        cost = wave_sum(cost_l);
        const int n_act = (int)(wave_sum((float)nact_l) + 0.5f);
        // make the tile stores of this wave visible to its own later loads
        __threadfence_block();
        __syncthreads();

        STAMP(0);
        // ------------------------------------------------------------ QP: interior point loop
        const bool use_ipm = (a.n_ipm > 0) && (n_act > 0);
        const int n_sweeps = use_ipm ? a.n_ipm : 1;
        bool qp_ok = true;
        for (int ii = 0; ii < n_sweeps; ++ii) {
            float tau = 0.0f;
            if (use_ipm) {
                float m_l = 0.0f;
                for (int k = lane; k < N; k += 64) {
                    const unsigned am = actm[k];
#pragma unroll
                    for (int j = 0; j < NG; ++j)
                        if ((am >> j) & 1u) m_l += sv[k * NG + j] * lv[k * NG + j];
                }
                tau = fmaxf(a.sigma * wave_sum(m_l) / (float)n_act, a.tau_min);
            }
            STAMP(1);
            // -------------------------------------------------------- phase R: backward sweep
            f32x4 P;
            {   // terminal: P~ = [diag(We)+reg_e, q_N; q_N', 0]
                const float qc = (c < NX) ? qv[N * NX + (c < NX ? c : 0)] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * q4 + r;
                    float v = 0.0f;
                    if (row == c && row < NX) v = we_c;
                    if (c == NX && row < NX) v = qv[N * NX + (row < NX ? row : 0)];
                    if (row == NX && c < NX) v = qc;
                    P[r] = v;
                }
            }
            for (int k = N - 1; k >= 0; --k) {
                const f32x4 Aa = load_tile(At + (size_t)k * TILE, lane);
                const f32x4 Ba = load_tile(Bt + (size_t)k * TILE, lane);
                f32x4 Qt, St, Rt;
                const float qc = (c < NX) ? qv[k * NX + (c < NX ? c : 0)] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * q4 + r;
                    float vq = 0.0f;
                    if (row == c && row < NX) vq = wq_c;
                    if (c == NX && row < NX) vq = qv[k * NX + (row < NX ? row : 0)];
                    if (row == NX && c < NX) vq = qc;
                    Qt[r] = vq;
                    St[r] = (c == NX && row < NU) ? rv[k * NU + (row < NU ? row : 0)] : 0.0f;
                    Rt[r] = (row == c && row < NU) ? wr_c : 0.0f;
                }
                if (use_ipm) {
                    const unsigned am = actm[k];
                    f32x4 Gs, Vt;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 4 * q4 + r;
                        const bool on = (j < NG) && ((am >> j) & 1u);
                        const int jj = (j < NG) ? j : 0;
                        const float s = sv[k * NG + jj], l = lv[k * NG + jj], cj = cv[k * NG + jj];
                        const float D = on ? l / s : 0.0f;
                        const float sq = sqrtf(D);
                        const float vj = on ? (tau / s + l + D * cj) : 0.0f;
                        Gs[r] = (on && c < NU) ? M::G(a.mp, jj, c) * sq : 0.0f;
                        Vt[r] = (on && c == NX) ? vj / sq : 0.0f;
                    }
                    Rt = xty(Gs, Gs, Rt);
                    St = xty(Gs, Vt, St);
                }
                f32x4 Kk;
                qp_ok = backward_stage<NU>(P, Aa, Ba, Qt, St, Rt, conv, lane, NX, Kk) && qp_ok;
                store_tile(Kt + (size_t)k * TILE, lane, Kk);
            }
            __threadfence_block();
            __syncthreads();
            STAMP(2);
            // -------------------------------------------------------- phase F: forward sweep
            float* oX = use_ipm ? dXp : dX;
            float* oU = use_ipm ? dUp : dU;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * q4 + r;
                float t = 0.0f;
                if (c == 0 && row < NX) t = x0[row < NX ? row : 0] - Xs[row < NX ? row : 0];
                if (c == 0 && row == NX) t = 1.0f;
                v[r] = t;
            }
            if (c == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * q4 + r < NX) oX[4 * q4 + r] = v[r];
            }
            for (int k = 0; k < N; ++k) {
                const f32x4 du = forward_stage(v, At + (size_t)k * TILE, Bt + (size_t)k * TILE,
                                               Kt + (size_t)k * TILE, lane);
                if (c == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * q4 + r;
                        if (row < NU) oU[k * NU + row] = du[r];
                        if (row < NX) oX[(k + 1) * NX + row] = v[r];
                    }
                }
            }
            __syncthreads();
            STAMP(3);
            // -------------------------------------------------------- phase I: IPM update
            if (use_ipm) {
                float ap_l = 1.0f, ad_l = 1.0f;
                for (int k = lane; k < N; k += 64) {
                    float du[NU], g[NG];
#pragma unroll
                    for (int i = 0; i < NU; ++i) du[i] = dUp[k * NU + i];
                    M::gdot(a.mp, du, g);
                    const unsigned am = actm[k];
#pragma unroll
                    for (int j = 0; j < NG; ++j) {
                        if (!((am >> j) & 1u)) continue;
                        const float s = sv[k * NG + j], l = lv[k * NG + j], cj = cv[k * NG + j];
                        const float ds = -(g[j] + cj) - s;
                        const float dl = tau / s - l - l / s * ds;
                        if (ds < 0.0f) ap_l = fminf(ap_l, -a.gamma * s / ds);
                        if (dl < 0.0f) ad_l = fminf(ad_l, -a.gamma * l / dl);
                    }
                }
                const float ap = wave_min(ap_l), ad = wave_min(ad_l);
                for (int k = lane; k < N; k += 64) {
                    float du[NU], g[NG];
#pragma unroll
                    for (int i = 0; i < NU; ++i) du[i] = dUp[k * NU + i];
                    M::gdot(a.mp, du, g);
                    const unsigned am = actm[k];
#pragma unroll
                    for (int j = 0; j < NG; ++j) {
                        if (!((am >> j) & 1u)) continue;
                        const float s = sv[k * NG + j], l = lv[k * NG + j], cj = cv[k * NG + j];
                        const float ds = -(g[j] + cj) - s;
                        const float dl = tau / s - l - l / s * ds;
                        sv[k * NG + j] = s + ap * ds;
                        lv[k * NG + j] = l + ad * dl;
                    }
                }
                for (int i = lane; i < (N + 1) * NX; i += 64) dX[i] += ap * (dXp[i] - dX[i]);
                for (int i = lane; i < N * NU; i += 64) dU[i] += ap * (dUp[i] - dU[i]);
                __syncthreads();
            }
        }
        STAMP(4);
        // ------------------------------------------------------------ phase S: step
        float sn_l = 0.0f;
        bool bad_l = false;
        for (int i = lane; i < (N + 1) * NX; i += 64) {
            const float v = dX[i];
            bad_l = bad_l || !(fabsf(v) <= 1e30f);
            sn_l = fmaxf(sn_l, fabsf(v));
        }
        for (int i = lane; i < N * NU; i += 64) {
            const float v = dU[i];
            bad_l = bad_l || !(fabsf(v) <= 1e30f);
            sn_l = fmaxf(sn_l, fabsf(v));
        }
        stepn = wave_max(sn_l);
        const bool bad = __any(bad_l);
        if (bad) { status = NMPC_STATUS_NAN; ++it; break; }
        alpha = 1.0f;
        if (a.line_search) {
            // l1 merit: cost + rho * (|x0 - X0| + sum |defects| + sum max(0, G u - h))
            auto merit = [&](float al) -> float {
                float m_l = 0.0f;
                for (int k = lane; k < N; k += 64) {
                    float x[NX], u[NU], xn[NX], p[NP > 0 ? NP : 1];
#pragma unroll
                    for (int i = 0; i < NX; ++i) x[i] = Xs[k * NX + i] + al * dX[k * NX + i];
#pragma unroll
                    for (int i = 0; i < NU; ++i) u[i] = Us[k * NU + i] + al * dU[k * NU + i];
#pragma unroll
                    for (int i = 0; i < NP; ++i) p[i] = pg[(size_t)k * NP + i];
                    M::step(a.mp, x, u, p, xn);
                    const float* yk = yr + (a.yref_per_stage ? (size_t)k * NY : 0);
                    float viol = 0.0f, cst = 0.0f;
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        const float e = x[i] - yk[i];
                        cst += 0.5f * a.W[i] * e * e;
                        viol += fabsf(xn[i] - (Xs[(k + 1) * NX + i] + al * dX[(k + 1) * NX + i]));
                    }
#pragma unroll
                    for (int i = 0; i < NU; ++i) {
                        const float e = u[i] - yk[NX + i];
                        cst += 0.5f * a.W[NX + i] * e * e;
                    }
                    if (a.n_ipm > 0) {
                        float g[NG];
                        M::gdot(a.mp, u, g);
                        const unsigned am = M::active_mask(a.mp, p);
#pragma unroll
                        for (int j = 0; j < NG; ++j)
                            if ((am >> j) & 1u) viol += fmaxf(g[j] - M::h(a.mp, j), 0.0f);
                    }
                    m_l += cst + a.rho * viol;
                }
                if (lane < NX) {
                    const float xe = Xs[N * NX + lane] + al * dX[N * NX + lane];
                    const float e = xe - yre[lane];
                    m_l += 0.5f * a.We[lane] * e * e;
                    m_l += a.rho * fabsf(x0[lane] - (Xs[lane] + al * dX[lane]));
                }
                return wave_sum(m_l);
            };
            const float m0 = merit(0.0f);
            for (int t = 0; t < 6; ++t) {
                const float m1 = merit(alpha);
                if (m1 < m0 || t == 5) break;
                alpha *= 0.5f;
            }
        }
        for (int i = lane; i < (N + 1) * NX; i += 64) Xs[i] += alpha * dX[i];
        for (int i = lane; i < N * NU; i += 64) Us[i] += alpha * dU[i];
        __syncthreads();
        if (!qp_ok) { status = NMPC_STATUS_QP; ++it; break; }
        if (a.nlp_tol > 0.0f && stepn < a.nlp_tol) { status = NMPC_STATUS_OK; ++it; break; }
    }
    for (int i = lane; i < (N + 1) * NX; i += 64) Xg[i] = Xs[i];
    for (int i = lane; i < N * NU; i += 64) Ug[i] = Us[i];
    STAMP(5);
#ifdef NMPC_STAMPS
    if (lane == 0 && a.dbg)
        for (int i = 0; i < 8; ++i) a.dbg[8 * b + i] = (float)st_acc[i];
#endif
    if (lane == 0) {
        if (a.status) a.status[b] = status;
        if (a.stats) {
            a.stats[4 * b + 0] = cost;
            a.stats[4 * b + 1] = stepn;
            a.stats[4 * b + 2] = alpha;
            a.stats[4 * b + 3] = (float)it;
        }
    }
}

}  // namespace nmpc
