// nmpc_policy.hip -- the learning update on gfx950 (C-ABI: include/nmpc_policy.h).
//
// Policy network  in -> [Linear, BatchNorm1d, ReLU] x L -> Linear -> out  (DAgger/utils/network.py:7-81)
// and one training step  L1 loss -> backward -> Adam  (DAgger/utils/train_locosafedagger.py:93-102).
// Everything is fp32, as the reference trains.  The three GEMMs of a layer -- Z = A W' (forward),
// dA = dZ W (backward data), dW = dZ' A (backward weights) -- are one LDS-tiled kernel on
// v_mfma_f32_32x32x2_f32; BatchNorm statistics, the ReLU mask, the L1 gradient, column sums and
// Adam are small HBM-bound kernels around it.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>

#include "../../include/nmpc.h"
#include "../../include/nmpc_policy.h"

namespace nmpc_policy {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float BN_EPS = 1e-5f, BN_MOMENTUM = 0.1f;
constexpr float ADAM_B1 = 0.9f, ADAM_B2 = 0.999f, ADAM_EPS = 1e-8f;

// ------------------------------------------------------------------------------------------------
// C[M][N] = A'[M][K] B'[K][N] (+ bias[N]),  row-major C with leading dimension ldc.
//   TA = false: A' = A, stored [M][K] (k contiguous);  TA = true: A' = A^T, A stored [K][M].
//   TB = false: B'[k][n] = B[n][k], B stored [N][K] (a Linear weight); TB = true: B stored [K][N].
// Block = 256 threads = 4 waves, 64 x 64 output tile, K in slices of 16 through the LDS.  Both LDS
// tiles are [row of the output][k] with k contiguous and an 80 B row stride (conflict-free 16 B
// reads); a wave owns a 32 x 32 quarter and runs v_mfma_f32_32x32x2_f32: lane (h, i) supplies row
// 32w + i and the k slots 8h .. 8h+7 of the slice as two 16 B reads per operand (the order of the
// k terms is irrelevant as long as A and B agree), eight MFMAs per slice.
// Loads are guarded element-wise (K = 47, N = 12 ... are not tile multiples; rows of X are not 16 B
// aligned), which is good enough: these GEMMs are 0.5 GFLOP each.
constexpr int BM = 64, BN = 64, BK = 16, LDT = BK + 4;

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                   const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                   int ldc, const float* __restrict__ bias) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wm = 32 * (w >> 1), wn = 32 * (w & 1);
    const int h = lane >> 5, li = lane & 31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += BK) {
        // global -> LDS: 64 x 16 elements per operand, four per thread
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int row, kk;
            if (!TA) { row = tid >> 2; kk = 4 * (tid & 3) + e; }          // k contiguous in memory
            else { kk = tid >> 4; row = 4 * (tid & 15) + e; }             // row contiguous in memory
            const int m = m0 + row, k = k0 + kk;
            float v = 0.0f;
            if (m < M && k < K) v = TA ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k];
            As[row * LDT + kk] = v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int row, kk;
            if (!TB) { row = tid >> 2; kk = 4 * (tid & 3) + e; }
            else { kk = tid >> 4; row = 4 * (tid & 15) + e; }
            const int n = n0 + row, k = k0 + kk;
            float v = 0.0f;
            if (n < N && k < K) v = TB ? B[(size_t)k * ldb + n] : B[(size_t)n * ldb + k];
            Bs[row * LDT + kk] = v;
        }
        __syncthreads();
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + (wm + li) * LDT + 8 * h);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + (wm + li) * LDT + 8 * h + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + (wn + li) * LDT + 8 * h);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bs + (wn + li) * LDT + 8 * h + 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc, 0, 0, 0);
        __syncthreads();
    }
    // accumulator layout of 32x32: lane (h, j) register r holds row 8*(r/4) + 4h + r%4 of column j
    const int n = n0 + wn + li;
    if (n < N) {
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm + 8 * (r >> 2) + 4 * h + (r & 3);
            if (m < M) C[(size_t)m * ldc + n] = acc[r] + bv;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Column statistics of Z[M][H] (BatchNorm1d in train mode): mean, 1/sqrt(biased var + eps); updates
// the running statistics (momentum 0.1, unbiased variance).  One block per 64 features, 256 threads
// = 64 features x 4 row groups; two passes (mean, then centred squares).
__global__ __launch_bounds__(256) void bn_stats_kernel(int M, int H, const float* __restrict__ Z, float* __restrict__ mu,
                                                       float* __restrict__ inv, float* __restrict__ run_mean,
                                                       float* __restrict__ run_var) {
    __shared__ float red[4][64];
    const int f = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const bool ok = f < H;
    float s = 0.0f;
    if (ok) for (int m = g; m < M; m += 4) s += Z[(size_t)m * H + f];
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    const float mean = (red[0][threadIdx.x & 63] + red[1][threadIdx.x & 63] + red[2][threadIdx.x & 63] + red[3][threadIdx.x & 63]) / (float)M;
    __syncthreads();
    float q = 0.0f;
    if (ok) for (int m = g; m < M; m += 4) { const float d = Z[(size_t)m * H + f] - mean; q += d * d; }
    red[g][threadIdx.x & 63] = q;
    __syncthreads();
    if (g == 0 && ok) {
        const float var = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)M;
        mu[f] = mean;
        inv[f] = 1.0f / sqrtf(var + BN_EPS);
        run_mean[f] = (1.0f - BN_MOMENTUM) * run_mean[f] + BN_MOMENTUM * mean;
        run_var[f] = (1.0f - BN_MOMENTUM) * run_var[f] + BN_MOMENTUM * var * (float)M / (float)(M > 1 ? M - 1 : 1);
    }
}

// out = relu(bn(z)) element-wise; mu/inv are the batch statistics (train) or derived from the running
// ones (eval: inv_from_var = true means inv[] holds a variance)
__global__ void bn_relu_kernel(size_t n, int H, const float* __restrict__ Z, const float* __restrict__ mu,
                               const float* __restrict__ inv, bool inv_is_var, const float* __restrict__ gamma,
                               const float* __restrict__ beta, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = (int)(i % H);
    float y = Z[i];
    if (mu) {
        const float iv = inv_is_var ? 1.0f / sqrtf(inv[f] + BN_EPS) : inv[f];
        y = (y - mu[f]) * iv * gamma[f] + beta[f];
    }
    out[i] = y > 0.0f ? y : 0.0f;
}

// L1 loss: dP = sign(P - Y) / n ; loss += sum |P - Y| / n   (nn.L1Loss, mean reduction)
__global__ void l1_kernel(size_t n, const float* __restrict__ P, const float* __restrict__ Y, float* __restrict__ dP,
                          float* __restrict__ loss) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float a = 0.0f;
    if (i < n) {
        const float d = P[i] - Y[i];
        a = fabsf(d) / (float)n;
        dP[i] = (d > 0.0f ? 1.0f : d < 0.0f ? -1.0f : 0.0f) / (float)n;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (loss && (threadIdx.x & 63) == 0 && a != 0.0f) atomicAdd(loss, a);
}

// Backward through ReLU and BatchNorm, pass 1: D <- D * (y > 0) in place; per feature
// dbeta = sum D, dgamma = sum D * xhat  (without BatchNorm: dbeta only = the bias gradient).
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(int M, int H, float* __restrict__ D, const float* __restrict__ Z,
                                                            const float* __restrict__ mu, const float* __restrict__ inv,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float rb[4][64], rg[4][64];
    const int fl = threadIdx.x & 63, f = blockIdx.x * 64 + fl, g = threadIdx.x >> 6;
    float sb = 0.0f, sg = 0.0f;
    if (f < H) {
        const bool bn = mu != nullptr;
        const float m_ = bn ? mu[f] : 0.0f, iv = bn ? inv[f] : 1.0f, ga = bn ? gamma[f] : 1.0f, be = bn ? beta[f] : 0.0f;
        for (int m = g; m < M; m += 4) {
            const size_t i = (size_t)m * H + f;
            const float xhat = (Z[i] - m_) * iv;
            const float y = bn ? xhat * ga + be : Z[i];
            const float d = y > 0.0f ? D[i] : 0.0f;
            D[i] = d;
            sb += d; sg += d * xhat;
        }
    }
    rb[g][fl] = sb; rg[g][fl] = sg;
    __syncthreads();
    if (g == 0 && f < H) {
        dbeta[f] = rb[0][fl] + rb[1][fl] + rb[2][fl] + rb[3][fl];
        if (dgamma) dgamma[f] = rg[0][fl] + rg[1][fl] + rg[2][fl] + rg[3][fl];
    }
}

// pass 2 (BatchNorm only): dz = gamma inv / M (M d - dbeta - xhat dgamma), in place
__global__ void bn_bwd_apply_kernel(size_t n, int M, int H, float* __restrict__ D, const float* __restrict__ Z,
                                    const float* __restrict__ mu, const float* __restrict__ inv,
                                    const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                    const float* __restrict__ dbeta) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = (int)(i % H);
    const float xhat = (Z[i] - mu[f]) * inv[f];
    D[i] = gamma[f] * inv[f] / (float)M * ((float)M * D[i] - dbeta[f] - xhat * dgamma[f]);
}

// column sums of D[M][H] (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(int M, int H, const float* __restrict__ D, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int fl = threadIdx.x & 63, f = blockIdx.x * 64 + fl, g = threadIdx.x >> 6;
    float s = 0.0f;
    if (f < H) for (int m = g; m < M; m += 4) s += D[(size_t)m * H + f];
    red[g][fl] = s;
    __syncthreads();
    if (g == 0 && f < H) out[f] = red[0][fl] + red[1][fl] + red[2][fl] + red[3][fl];
}

// torch.optim.Adam, defaults (betas 0.9 / 0.999, eps 1e-8, no weight decay); c1 = 1 - b1^t, c2 = 1 - b2^t
__global__ void adam_kernel(size_t n, float* __restrict__ theta, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float lr, float c1, float c2) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = ADAM_B1 * m[i] + (1.0f - ADAM_B1) * gi;
    const float vi = ADAM_B2 * v[i] + (1.0f - ADAM_B2) * gi * gi;
    m[i] = mi; v[i] = vi;
    theta[i] -= lr * (mi / c1) / (sqrtf(vi / c2) + ADAM_EPS);
}

}  // namespace nmpc_policy

// ================================================================================================
namespace {

using namespace nmpc_policy;

thread_local std::string g_policy_create_error;

struct Policy {
    nmpc_policy_dims d{};
    int device = 0;
    size_t n_theta = 0;
    // offsets into theta
    size_t oW[17]{}, ob[17]{}, og[16]{}, obe[16]{};
    float *theta = nullptr, *grad = nullptr, *m = nullptr, *v = nullptr;
    float *run_mean = nullptr, *run_var = nullptr, *mu = nullptr, *inv = nullptr;
    float *act = nullptr;      // activations a_0 = X copy is not needed: a_l for l = 1..L  [L][B][hidden]
    float *z = nullptr;        // pre-BatchNorm outputs z_l  [L][B][hidden]
    float *dbuf[2] = {nullptr, nullptr};   // [B][max(hidden, n_in)]
    float *pred = nullptr, *dpred = nullptr;
    long long step = 0;
    std::string err;
};

int pfail(Policy* p, int code, const std::string& msg) {
    if (p) p->err = msg; else g_policy_create_error = msg;
    return code;
}
#define PTRY(p, expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return pfail(p, NMPC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <bool TA, bool TB>
void gemm(hipStream_t st, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
          const float* bias) {
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    hipLaunchKernelGGL((gemm_kernel<TA, TB>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias);
}

unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

// forward pass; train: batch statistics (and the caches z, act for the backward pass)
int forward(Policy* p, int B, const float* X, float* out, bool train, hipStream_t st) {
    const int L = p->d.n_hidden, H = p->d.hidden;
    const float* a = X;
    int fan_in = p->d.n_in;
    for (int l = 0; l < L; ++l) {
        float* z = p->z + (size_t)l * p->d.batch_max * H;
        float* o = p->act + (size_t)l * p->d.batch_max * H;
        gemm<false, false>(st, B, H, fan_in, a, fan_in, p->theta + p->oW[l], fan_in, z, H, p->theta + p->ob[l]);
        const size_t n = (size_t)B * H;
        if (p->d.batch_norm) {
            float *mu = p->mu + (size_t)l * H, *inv = p->inv + (size_t)l * H;
            if (train) {
                hipLaunchKernelGGL(bn_stats_kernel, dim3((H + 63) / 64), dim3(256), 0, st, B, H, z, mu, inv,
                                   p->run_mean + (size_t)l * H, p->run_var + (size_t)l * H);
                hipLaunchKernelGGL(bn_relu_kernel, dim3(blocks_for(n)), dim3(256), 0, st, n, H, z, mu, inv, false,
                                   p->theta + p->og[l], p->theta + p->obe[l], o);
            } else {
                hipLaunchKernelGGL(bn_relu_kernel, dim3(blocks_for(n)), dim3(256), 0, st, n, H, z,
                                   p->run_mean + (size_t)l * H, p->run_var + (size_t)l * H, true,
                                   p->theta + p->og[l], p->theta + p->obe[l], o);
            }
        } else {
            hipLaunchKernelGGL(bn_relu_kernel, dim3(blocks_for(n)), dim3(256), 0, st, n, H, z, (const float*)nullptr,
                               (const float*)nullptr, false, (const float*)nullptr, (const float*)nullptr, o);
        }
        a = o; fan_in = H;
    }
    gemm<false, false>(st, B, p->d.n_out, H, a, H, p->theta + p->oW[L], H, out, p->d.n_out, p->theta + p->ob[L]);
    return NMPC_OK;
}

}  // namespace

extern "C" {

int nmpc_policy_create(const nmpc_policy_dims* dims, int device_id, void** handle) {
    if (!dims || !handle) return pfail(nullptr, NMPC_E_ARG, "null argument");
    *handle = nullptr;
    if (dims->n_in < 1 || dims->n_out < 1 || dims->hidden < 1 || dims->n_hidden < 1 || dims->n_hidden > 16 || dims->batch_max < 1)
        return pfail(nullptr, NMPC_E_ARG, "need n_in, n_out, hidden, batch_max >= 1 and 1 <= n_hidden <= 16");
    Policy* p = new Policy();
    p->d = *dims;
    p->d.batch_norm = dims->batch_norm ? 1 : 0;
    p->device = device_id;
    const int L = dims->n_hidden, H = dims->hidden;
    size_t off = 0;
    for (int l = 0; l < L; ++l) {
        const int fan_in = l == 0 ? dims->n_in : H;
        p->oW[l] = off; off += (size_t)H * fan_in;
        p->ob[l] = off; off += H;
        if (p->d.batch_norm) { p->og[l] = off; off += H; p->obe[l] = off; off += H; }
    }
    p->oW[L] = off; off += (size_t)dims->n_out * H;
    p->ob[L] = off; off += dims->n_out;
    p->n_theta = off;
    const size_t Bm = dims->batch_max;
    const size_t wide = (size_t)(H > dims->n_in ? H : dims->n_in);
    struct { float** ptr; size_t n; } bufs[] = {
        {&p->theta, off}, {&p->grad, off}, {&p->m, off}, {&p->v, off},
        {&p->run_mean, (size_t)L * H}, {&p->run_var, (size_t)L * H}, {&p->mu, (size_t)L * H}, {&p->inv, (size_t)L * H},
        {&p->act, (size_t)L * Bm * H}, {&p->z, (size_t)L * Bm * H}, {&p->dbuf[0], Bm * wide}, {&p->dbuf[1], Bm * wide},
        {&p->pred, Bm * dims->n_out}, {&p->dpred, Bm * dims->n_out}};
    hipError_t e = hipSetDevice(device_id);
    for (auto& b : bufs) {
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(b.ptr), b.n * sizeof(float));
        if (e == hipSuccess) e = hipMemset(*b.ptr, 0, b.n * sizeof(float));
    }
    if (e != hipSuccess) {
        g_policy_create_error = std::string("nmpc_policy_create: ") + hipGetErrorString(e);
        for (auto& b : bufs) if (*b.ptr) (void)hipFree(*b.ptr);
        delete p;
        return NMPC_E_HIP;
    }
    *handle = p;
    return NMPC_OK;
}

void nmpc_policy_destroy(void* handle) {
    Policy* p = static_cast<Policy*>(handle);
    if (!p) return;
    (void)hipSetDevice(p->device);
    float* all[] = {p->theta, p->grad, p->m, p->v, p->run_mean, p->run_var, p->mu, p->inv, p->act, p->z,
                    p->dbuf[0], p->dbuf[1], p->pred, p->dpred};
    for (float* q : all) if (q) (void)hipFree(q);
    delete p;
}

const char* nmpc_policy_last_error(void* handle) {
    Policy* p = static_cast<Policy*>(handle);
    return p ? p->err.c_str() : g_policy_create_error.c_str();
}

size_t nmpc_policy_param_count(void* handle) {
    Policy* p = static_cast<Policy*>(handle);
    return p ? p->n_theta : 0;
}

int nmpc_policy_set_params(void* handle, const float* theta, const float* running_mean, const float* running_var,
                           void* stream) {
    Policy* p = static_cast<Policy*>(handle);
    if (!p || !theta) return pfail(p, NMPC_E_ARG, "null argument");
    if (p->d.batch_norm && (!running_mean || !running_var)) return pfail(p, NMPC_E_ARG, "running statistics missing");
    hipStream_t st = static_cast<hipStream_t>(stream);
    PTRY(p, hipSetDevice(p->device));
    PTRY(p, hipMemcpyAsync(p->theta, theta, p->n_theta * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (p->d.batch_norm) {
        const size_t n = (size_t)p->d.n_hidden * p->d.hidden * sizeof(float);
        PTRY(p, hipMemcpyAsync(p->run_mean, running_mean, n, hipMemcpyDeviceToDevice, st));
        PTRY(p, hipMemcpyAsync(p->run_var, running_var, n, hipMemcpyDeviceToDevice, st));
    }
    PTRY(p, hipMemsetAsync(p->m, 0, p->n_theta * sizeof(float), st));
    PTRY(p, hipMemsetAsync(p->v, 0, p->n_theta * sizeof(float), st));
    p->step = 0;
    return NMPC_OK;
}

int nmpc_policy_get_params(void* handle, float* theta, float* running_mean, float* running_var, void* stream) {
    Policy* p = static_cast<Policy*>(handle);
    if (!p || !theta) return pfail(p, NMPC_E_ARG, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    PTRY(p, hipSetDevice(p->device));
    PTRY(p, hipMemcpyAsync(theta, p->theta, p->n_theta * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (p->d.batch_norm && running_mean && running_var) {
        const size_t n = (size_t)p->d.n_hidden * p->d.hidden * sizeof(float);
        PTRY(p, hipMemcpyAsync(running_mean, p->run_mean, n, hipMemcpyDeviceToDevice, st));
        PTRY(p, hipMemcpyAsync(running_var, p->run_var, n, hipMemcpyDeviceToDevice, st));
    }
    return NMPC_OK;
}

int nmpc_policy_forward(void* handle, int B, const float* X, float* Y, void* stream) {
    Policy* p = static_cast<Policy*>(handle);
    if (!p) return NMPC_E_ARG;
    if (B == 0) return NMPC_OK;
    if (!X || !Y) return pfail(p, NMPC_E_ARG, "null argument");
    if (B < 0 || B > p->d.batch_max) return pfail(p, NMPC_E_ARG, "B exceeds batch_max");
    hipStream_t st = static_cast<hipStream_t>(stream);
    PTRY(p, hipSetDevice(p->device));
    forward(p, B, X, Y, false, st);
    PTRY(p, hipGetLastError());
    return NMPC_OK;
}

int nmpc_policy_train_step(void* handle, int B, const float* X, const float* Y, float lr, float* loss, float* pred,
                           void* stream) {
    Policy* p = static_cast<Policy*>(handle);
    if (!p) return NMPC_E_ARG;
    if (!X || !Y) return pfail(p, NMPC_E_ARG, "null argument");
    if (B < 1 || B > p->d.batch_max) return pfail(p, NMPC_E_ARG, "B out of range");
    if (p->d.batch_norm && B < 2) return pfail(p, NMPC_E_ARG, "BatchNorm in train mode needs B >= 2");
    if (!(lr > 0.0f)) return pfail(p, NMPC_E_ARG, "learning rate must be positive");
    hipStream_t st = static_cast<hipStream_t>(stream);
    PTRY(p, hipSetDevice(p->device));
    const int L = p->d.n_hidden, H = p->d.hidden, no = p->d.n_out;
    const bool bn = p->d.batch_norm != 0;
    forward(p, B, X, p->pred, true, st);
    if (pred) PTRY(p, hipMemcpyAsync(pred, p->pred, (size_t)B * no * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (loss) PTRY(p, hipMemsetAsync(loss, 0, sizeof(float), st));
    const size_t np_ = (size_t)B * no;
    hipLaunchKernelGGL(l1_kernel, dim3(blocks_for(np_)), dim3(256), 0, st, np_, p->pred, Y, p->dpred, loss);
    // output layer: dW = dP' a_L, db = colsum dP, d = dP W
    const float* aL = p->act + (size_t)(L - 1) * p->d.batch_max * H;
    gemm<true, true>(st, no, H, B, p->dpred, no, aL, H, p->grad + p->oW[L], H, nullptr);
    hipLaunchKernelGGL(colsum_kernel, dim3((no + 63) / 64), dim3(256), 0, st, B, no, p->dpred, p->grad + p->ob[L]);
    float* d = p->dbuf[0];
    float* dn = p->dbuf[1];
    gemm<false, true>(st, B, H, no, p->dpred, no, p->theta + p->oW[L], H, d, H, nullptr);
    for (int l = L - 1; l >= 0; --l) {
        const int fan_in = l == 0 ? p->d.n_in : H;
        const float* z = p->z + (size_t)l * p->d.batch_max * H;
        const float* a = l == 0 ? X : p->act + (size_t)(l - 1) * p->d.batch_max * H;
        const float *mu = bn ? p->mu + (size_t)l * H : nullptr, *inv = bn ? p->inv + (size_t)l * H : nullptr;
        float* dgamma = bn ? p->grad + p->og[l] : nullptr;
        float* dbeta = bn ? p->grad + p->obe[l] : p->grad + p->ob[l];
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((H + 63) / 64), dim3(256), 0, st, B, H, d, z, mu, inv,
                           bn ? p->theta + p->og[l] : nullptr, bn ? p->theta + p->obe[l] : nullptr, dgamma, dbeta);
        if (bn) {
            const size_t n = (size_t)B * H;
            hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(n)), dim3(256), 0, st, n, B, H, d, z, mu, inv,
                               p->theta + p->og[l], dgamma, dbeta);
            hipLaunchKernelGGL(colsum_kernel, dim3((H + 63) / 64), dim3(256), 0, st, B, H, d, p->grad + p->ob[l]);
        }
        gemm<true, true>(st, H, fan_in, B, d, H, a, fan_in, p->grad + p->oW[l], fan_in, nullptr);
        if (l > 0) {
            gemm<false, true>(st, B, H, H, d, H, p->theta + p->oW[l], H, dn, H, nullptr);
            float* t = d; d = dn; dn = t;
        }
    }
    p->step += 1;
    const float c1 = 1.0f - std::pow(ADAM_B1, (float)p->step), c2 = 1.0f - std::pow(ADAM_B2, (float)p->step);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(p->n_theta)), dim3(256), 0, st, p->n_theta, p->theta, p->grad,
                       p->m, p->v, lr, c1, c2);
    PTRY(p, hipGetLastError());
    return NMPC_OK;
}

}  // extern "C"
