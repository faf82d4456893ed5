// nmpc_policy.hip -- the learning update on gfx950 (C-ABI: include/nmpc_policy.h).
//
// Policy network  in -> [Linear, BatchNorm1d, ReLU] x L -> Linear -> out  (DAgger/utils/network.py:7-81)
// and one training step  L1 loss -> backward -> Adam  (DAgger/utils/train_locosafedagger.py:93-102).
// Everything is fp32, as the reference trains.  The three GEMMs of a layer -- Z = A W' (forward),
// dA = dZ W (backward data), dW = dZ' A (backward weights) -- are one LDS-tiled kernel on
// v_mfma_f32_32x32x2_f32; BatchNorm statistics, the ReLU mask, the L1 gradient, column sums and
// Adam are small HBM-bound kernels around it.
#include <hip/hip_runtime.h>

#include "nmpc_device_guard.hpp"

#include <cmath>
#include <cstdint>
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
// VEC = false: loads are guarded element-wise (K = 47, N = 12 ... are not tile multiples; rows of X
// are not 16 B aligned).  VEC = true (M, N multiples of 64, K of 16, leading dimensions multiples of
// 4): 16 B global loads without guards.  gridDim.z > 1 splits K (the weight-gradient GEMMs contract over the
// batch and have few output tiles: 64 at hidden = 512, 8 for the input and output layers): every split writes
// its partial product to split_part[z][M][N], split_reduce_kernel adds them up in split order -- no float
// atomics, so a training step is reproducible run to run (as the database statistics are).
constexpr int BM = 64, BN = 64, BK = 16, LDT = BK + 4;

// The K loop advances KS slices at a time: their 2 x KS 16 B loads per thread are issued together one
// iteration ahead, so a global-memory round trip (~1 us) is paid once per KS x 8 MFMAs instead of once
// per 8 (with single slices the loop ran at memory latency: 15 us for a GEMM whose MFMAs take 7).
constexpr int KS = 4;

template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                   const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                   int ldc, const float* __restrict__ bias, float* __restrict__ split_part) {
    __shared__ __attribute__((aligned(16))) float As[KS][BM * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[KS][BN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wm = 32 * (w >> 1), wn = 32 * (w & 1);
    const int h = lane >> 5, li = lane & 31;
    // K range of this block: whole slices of BK per split (the last split may be short or empty)
    const int kper = (((K + (int)gridDim.z - 1) / (int)gridDim.z) + BK - 1) / BK * BK;
    const int kbeg = blockIdx.z * kper, kend = kbeg + kper < K ? kbeg + kper : K;
    const bool empty_split = kbeg >= kend;                        // block-uniform: an empty split contributes zeros
    if (empty_split && gridDim.z == 1) return;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // this thread's four elements of a 64 x 16 operand slice: (row, kk .. kk+3) if k is contiguous in
    // memory, (row .. row+3, kk) if the rows are
    const int a_row = TA ? 4 * (tid & 15) : tid >> 2, a_kk = TA ? tid >> 4 : 4 * (tid & 3);
    const int b_row = TB ? 4 * (tid & 15) : tid >> 2, b_kk = TB ? tid >> 4 : 4 * (tid & 3);
    auto load_a = [&](int k0) -> f32x4 {          // slice starting at k0 (zeros past kend)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (VEC) {
            if (k0 < kend)
                v = TA ? *reinterpret_cast<const f32x4*>(A + (size_t)(k0 + a_kk) * lda + m0 + a_row)
                       : *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + a_row) * lda + k0 + a_kk);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + a_row + (TA ? e : 0), k = k0 + a_kk + (TA ? 0 : e);
                if (m < M && k < kend) v[e] = TA ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k];
            }
        }
        return v;
    };
    auto load_b = [&](int k0) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (VEC) {
            if (k0 < kend)
                v = TB ? *reinterpret_cast<const f32x4*>(B + (size_t)(k0 + b_kk) * ldb + n0 + b_row)
                       : *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + b_row) * ldb + k0 + b_kk);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + b_row + (TB ? e : 0), k = k0 + b_kk + (TB ? 0 : e);
                if (n < N && k < kend) v[e] = TB ? B[(size_t)k * ldb + n] : B[(size_t)n * ldb + k];
            }
        }
        return v;
    };
    f32x4 va[KS], vb[KS];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < KS; ++j) { va[j] = load_a(k0 + j * BK); vb[j] = load_b(k0 + j * BK); }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            if (TA) { for (int e = 0; e < 4; ++e) As[j][(a_row + e) * LDT + a_kk] = va[j][e]; }
            else *reinterpret_cast<f32x4*>(As[j] + a_row * LDT + a_kk) = va[j];
            if (TB) { for (int e = 0; e < 4; ++e) Bs[j][(b_row + e) * LDT + b_kk] = vb[j][e]; }
            else *reinterpret_cast<f32x4*>(Bs[j] + b_row * LDT + b_kk) = vb[j];
        }
    };
    constexpr int STEP = BK * KS;
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += STEP) {
        stage();
        __syncthreads();
        if (k0 + STEP < kend) fetch(k0 + STEP);                   // in flight during the MFMAs of this step
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            if (k0 + j * BK < kend) {                             // block-uniform
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(As[j] + (wm + li) * LDT + 8 * h);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(As[j] + (wm + li) * LDT + 8 * h + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs[j] + (wn + li) * LDT + 8 * h);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bs[j] + (wn + li) * LDT + 8 * h + 4);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // accumulator layout of 32x32: lane (h, j) register r holds row 8*(r/4) + 4h + r%4 of column j
    const int n = n0 + wn + li;
    if (n < N) {
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm + 8 * (r >> 2) + 4 * h + (r & 3);
            if (m < M) {
                if (gridDim.z > 1) split_part[((size_t)blockIdx.z * M + m) * N + n] = acc[r];     // summed in split order below
                else C[(size_t)m * ldc + n] = acc[r] + bv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Column reductions over the batch.  A column sum over M rows with one block per 64 features would
// leave 248 of 256 CUs idle (H = 512), so rows are split into RCHUNK chunks: partial kernels write
// part[q][chunk][f] (deterministic, no atomics), finalize kernels add the chunks up per feature.
constexpr int RCHUNK = 32;

__device__ __forceinline__ void chunk_rows(int M, int& r0, int& r1) {
    const int per = (M + RCHUNK - 1) / RCHUNK;
    r0 = blockIdx.y * per;
    r1 = r0 + per < M ? r0 + per : M;
}

// BatchNorm statistics, pass 1: sums of (z - pivot) and (z - pivot)^2 per feature and chunk, pivot = row 0
// of the feature (a shift keeps E[d^2] - E[d]^2 well conditioned when |mean| >> std)
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(int M, int H, const float* __restrict__ Z,
                                                               float* __restrict__ part) {
    __shared__ float r1s[4][64], r2s[4][64];
    const int fl = threadIdx.x & 63, f = blockIdx.x * 64 + fl, g = threadIdx.x >> 6;
    int r0, r1;
    chunk_rows(M, r0, r1);
    float s1 = 0.0f, s2 = 0.0f;
    if (f < H) {
        const float piv = Z[f];
        for (int m = r0 + g; m < r1; m += 4) { const float d = Z[(size_t)m * H + f] - piv; s1 += d; s2 += d * d; }
    }
    r1s[g][fl] = s1; r2s[g][fl] = s2;
    __syncthreads();
    if (g == 0 && f < H) {
        part[((size_t)0 * RCHUNK + blockIdx.y) * H + f] = r1s[0][fl] + r1s[1][fl] + r1s[2][fl] + r1s[3][fl];
        part[((size_t)1 * RCHUNK + blockIdx.y) * H + f] = r2s[0][fl] + r2s[1][fl] + r2s[2][fl] + r2s[3][fl];
    }
}

// Train-mode BatchNorm + ReLU with the statistics finalised in the same launch: a block owns 64
// features x one row chunk, adds up the RCHUNK partial sums of its features (pass 1 above), and
// applies them to its rows; the blocks of chunk 0 also publish mean, 1/sqrt(var + eps) for the
// backward pass and update the running statistics (momentum 0.1, unbiased variance).
__global__ __launch_bounds__(256) void bn_relu_train_kernel(int M, int H, const float* __restrict__ Z,
                                                            const float* __restrict__ part, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ out,
                                                            float* __restrict__ mu, float* __restrict__ inv,
                                                            float* __restrict__ run_mean, float* __restrict__ run_var) {
    __shared__ float smu[64], sinv[64];
    const int fl = threadIdx.x & 63, f = blockIdx.x * 64 + fl, g = threadIdx.x >> 6;
    if (g == 0 && f < H) {
        float s1 = 0.0f, s2 = 0.0f;
        for (int c = 0; c < RCHUNK; ++c) { s1 += part[((size_t)0 * RCHUNK + c) * H + f]; s2 += part[((size_t)1 * RCHUNK + c) * H + f]; }
        const float dm = s1 / (float)M;
        const float mean = Z[f] + dm;
        float var = s2 / (float)M - dm * dm;
        var = var > 0.0f ? var : 0.0f;
        const float iv = 1.0f / sqrtf(var + BN_EPS);
        smu[fl] = mean; sinv[fl] = iv;
        if (blockIdx.y == 0) {
            mu[f] = mean; inv[f] = iv;
            run_mean[f] = (1.0f - BN_MOMENTUM) * run_mean[f] + BN_MOMENTUM * mean;
            run_var[f] = (1.0f - BN_MOMENTUM) * run_var[f] + BN_MOMENTUM * var * (float)M / (float)(M > 1 ? M - 1 : 1);
        }
    }
    __syncthreads();
    if (f >= H) return;
    int r0, r1;
    chunk_rows(M, r0, r1);
    const float m_ = smu[fl], iv = sinv[fl], ga = gamma[f], be = beta[f];
    for (int m = r0 + g; m < r1; m += 4) {
        const size_t i = (size_t)m * H + f;
        const float y = (Z[i] - m_) * iv * ga + be;
        out[i] = y > 0.0f ? y : 0.0f;
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

// L1 loss: dP = sign(P - Y) / n ; loss_part[block] = sum |P - Y| / n over the block (nn.L1Loss, mean
// reduction; adam_kernel adds the blocks up); and the bias gradient of the output layer, the column sums of
// dP.  Every term of such a column sum is +-1/n, so it is accumulated as an INTEGER count of signs
// (sign_count[n_out], zero on entry, turned into count / n by adam_kernel): exact and independent of the
// order of the atomics.  With float atomics a balanced column (as many + as -: an exactly zero gradient)
// came out as 0 or as +-1 ulp depending on the order, and Adam -- whose first steps only look at the sign --
// moved that bias or not from run to run.
__global__ __launch_bounds__(256) void l1_kernel(size_t n, int n_out, const float* __restrict__ P, const float* __restrict__ Y,
                                                 float* __restrict__ dP, float* __restrict__ loss_part, int* __restrict__ sign_count) {
    __shared__ int bins[256];
    __shared__ float wsum[4];
    bins[threadIdx.x] = 0;
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float a = 0.0f;
    if (i < n) {
        const float d = P[i] - Y[i];
        a = fabsf(d) / (float)n;
        const int sgn = d > 0.0f ? 1 : d < 0.0f ? -1 : 0;
        dP[i] = (float)sgn / (float)n;
        if (sgn != 0) atomicAdd(&bins[i % n_out], sgn);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) loss_part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    if ((int)threadIdx.x < n_out && bins[threadIdx.x] != 0) atomicAdd(sign_count + threadIdx.x, bins[threadIdx.x]);
}

// Backward through ReLU and BatchNorm, pass 1: D <- D * (y > 0) in place; per feature and row chunk
// the partial sums of D (-> dbeta, or the bias gradient without BatchNorm) and of D * xhat (-> dgamma)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(int M, int H, float* __restrict__ D, const float* __restrict__ Z,
                                                             const float* __restrict__ mu, const float* __restrict__ inv,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ part) {
    __shared__ float rb[4][64], rg[4][64];
    const int fl = threadIdx.x & 63, f = blockIdx.x * 64 + fl, g = threadIdx.x >> 6;
    int r0, r1;
    chunk_rows(M, r0, r1);
    float sb = 0.0f, sg = 0.0f;
    if (f < H) {
        const bool bn = mu != nullptr;
        const float m_ = bn ? mu[f] : 0.0f, iv = bn ? inv[f] : 1.0f, ga = bn ? gamma[f] : 1.0f, be = bn ? beta[f] : 0.0f;
        for (int m = r0 + g; m < r1; m += 4) {
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
        part[((size_t)0 * RCHUNK + blockIdx.y) * H + f] = rb[0][fl] + rb[1][fl] + rb[2][fl] + rb[3][fl];
        part[((size_t)1 * RCHUNK + blockIdx.y) * H + f] = rg[0][fl] + rg[1][fl] + rg[2][fl] + rg[3][fl];
    }
}
// column sums of D[M][H], partial
__global__ __launch_bounds__(256) void colsum_partial_kernel(int M, int H, const float* __restrict__ D, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int fl = threadIdx.x & 63, f = blockIdx.x * 64 + fl, g = threadIdx.x >> 6;
    int r0, r1;
    chunk_rows(M, r0, r1);
    float s = 0.0f;
    if (f < H) for (int m = r0 + g; m < r1; m += 4) s += D[(size_t)m * H + f];
    red[g][fl] = s;
    __syncthreads();
    if (g == 0 && f < H) part[(size_t)blockIdx.y * H + f] = red[0][fl] + red[1][fl] + red[2][fl] + red[3][fl];
}
// out0[f] = sum of part[0][.][f], out1[f] = sum of part[1][.][f] (out1 may be null)
__global__ void reduce_final_kernel(int H, const float* __restrict__ part, float* __restrict__ out0, float* __restrict__ out1) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= H) return;
    float s0 = 0.0f, s1 = 0.0f;
    for (int c = 0; c < RCHUNK; ++c) {
        s0 += part[((size_t)0 * RCHUNK + c) * H + f];
        if (out1) s1 += part[((size_t)1 * RCHUNK + c) * H + f];
    }
    out0[f] = s0;
    if (out1) out1[f] = s1;
}

// pass 2 (BatchNorm only), same block shape as pass 1: adds up the partial sums (-> dgamma, dbeta,
// published by the blocks of chunk 0), applies  dz = gamma inv / M (M d - dbeta - xhat dgamma)  in
// place to its rows, and leaves the partial column sums of dz (the bias gradient in front of the
// BatchNorm -- zero up to rounding, but the reference's Adam steps on it) in part_b[chunk][f].
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(int M, int H, float* __restrict__ D, const float* __restrict__ Z,
                                                           const float* __restrict__ mu, const float* __restrict__ inv,
                                                           const float* __restrict__ gamma, const float* __restrict__ part,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ part_b) {
    __shared__ float sdg[64], sdb[64], red[4][64];
    const int fl = threadIdx.x & 63, f = blockIdx.x * 64 + fl, g = threadIdx.x >> 6;
    if (g == 0 && f < H) {
        float s0 = 0.0f, s1 = 0.0f;
        for (int c = 0; c < RCHUNK; ++c) { s0 += part[((size_t)0 * RCHUNK + c) * H + f]; s1 += part[((size_t)1 * RCHUNK + c) * H + f]; }
        sdb[fl] = s0; sdg[fl] = s1;
        if (blockIdx.y == 0) { dbeta[f] = s0; dgamma[f] = s1; }
    }
    __syncthreads();
    int r0, r1;
    chunk_rows(M, r0, r1);
    float sb = 0.0f;
    if (f < H) {
        const float m_ = mu[f], iv = inv[f], c = gamma[f] * iv / (float)M, db = sdb[fl], dg = sdg[fl];
        for (int m = r0 + g; m < r1; m += 4) {
            const size_t i = (size_t)m * H + f;
            const float xhat = (Z[i] - m_) * iv;
            const float dz = c * ((float)M * D[i] - db - xhat * dg);
            D[i] = dz;
            sb += dz;
        }
    }
    red[g][fl] = sb;
    __syncthreads();
    if (g == 0 && f < H) part_b[(size_t)blockIdx.y * H + f] = red[0][fl] + red[1][fl] + red[2][fl] + red[3][fl];
}

// torch.optim.Adam, defaults (betas 0.9 / 0.999, eps 1e-8, no weight decay); c1 = 1 - b1^t, c2 = 1 - b2^t.
// The last kernel of a step also ties up its loose ends (a launch each otherwise):
//   * the bias gradients in front of a BatchNorm arrive as partial column sums part_b[l][chunk][f]
//     (bn_bwd_apply_kernel) and are added up here by the threads that own those entries;
//   * grad and the sign counts are left zeroed for the next step (the split-K GEMMs and l1_kernel accumulate into them);
//   * block 0 adds up the per-block partial losses of l1_kernel.
struct AdamTail {
    const float* part_b;                 // nullptr without BatchNorm
    unsigned long long ob[16];           // offsets of the hidden layers' biases in theta
    int L, H;
    const float* loss_part;              // per-block partial losses of l1_kernel
    int n_loss_part;
    float* loss;                         // may be null
    int* sign_count;                     // l1_kernel's sign counts -> bias gradient of the output layer; zeroed here
    unsigned long long ob_out;           // offset of that bias in theta
    int n_out;
    float n_pred;                        // batch x n_out
};
__global__ void adam_kernel(size_t n, float* __restrict__ theta, float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float lr, float c1, float c2, const AdamTail t) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && t.loss) {
        float s = 0.0f;
        for (int b = 0; b < t.n_loss_part; ++b) s += t.loss_part[b];
        t.loss[0] = s;
    }
    if (i >= n) return;
    float gi = g[i];
    if (i >= t.ob_out && i < t.ob_out + (size_t)t.n_out) {
        gi = (float)t.sign_count[i - t.ob_out] / t.n_pred;
        t.sign_count[i - t.ob_out] = 0;
    }
    if (t.part_b) {
        for (int l = 0; l < t.L; ++l) {
            if (i >= t.ob[l] && i < t.ob[l] + (size_t)t.H) {
                const size_t f = i - t.ob[l];
                gi = 0.0f;
                for (int c = 0; c < RCHUNK; ++c) gi += t.part_b[((size_t)l * RCHUNK + c) * t.H + f];
            }
        }
    }
    g[i] = 0.0f;
    const float mi = ADAM_B1 * m[i] + (1.0f - ADAM_B1) * gi;
    const float vi = ADAM_B2 * v[i] + (1.0f - ADAM_B2) * gi * gi;
    m[i] = mi; v[i] = vi;
    theta[i] -= lr * (mi / c1) / (sqrtf(vi / c2) + ADAM_EPS);
}

// ------------------------------------------------------------------------------------------------
// Weighted sampling with replacement: fp64 inclusive prefix sums of the weights (blocks of SCAN_CHUNK
// elements: local sums -> scan of the block totals -> local scan with offset), then one inverse-CDF
// lookup per sample with a Philox-4x32-10 uniform number.
constexpr int SCAN_CHUNK = 2048;   // 256 threads x 8

__global__ __launch_bounds__(256) void scan_block_totals_kernel(const float* __restrict__ w, long long n, double* __restrict__ tot) {
    __shared__ double red[256];
    const long long base = (long long)blockIdx.x * SCAN_CHUNK;
    double s = 0.0;
    for (int e = 0; e < 8; ++e) {
        const long long i = base + (long long)threadIdx.x * 8 + e;
        if (i < n) s += (double)w[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) tot[blockIdx.x] = red[0];
}
// exclusive scan of the block totals in place, total weight to tot[nb] (one thread: nb is n / 2048)
__global__ void scan_totals_kernel(double* __restrict__ tot, long long nb) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double run = 0.0;
    for (long long b = 0; b < nb; ++b) { const double t = tot[b]; tot[b] = run; run += t; }
    tot[nb] = run;
}
__global__ __launch_bounds__(256) void scan_write_kernel(const float* __restrict__ w, long long n, const double* __restrict__ tot,
                                                         double* __restrict__ cdf) {
    __shared__ double pre[256];
    const long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)threadIdx.x * 8;
    double v[8], s = 0.0;
    for (int e = 0; e < 8; ++e) { v[e] = (base + e < n) ? (double)w[base + e] : 0.0; s += v[e]; }
    pre[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double run = 0.0; for (int t = 0; t < 256; ++t) { const double x = pre[t]; pre[t] = run; run += x; } }
    __syncthreads();
    double run = tot[blockIdx.x] + pre[threadIdx.x];
    for (int e = 0; e < 8; ++e) { run += v[e]; if (base + e < n) cdf[base + e] = run; }
}

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned& o0, unsigned& o1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1;
}

// (the total weight is read from the device: no host round trip)
__global__ void sample_kernel(const double* __restrict__ cdf, long long n, const double* __restrict__ total_ptr,
                                          int num_samples, unsigned long long seed, int* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_samples) return;
    unsigned a, b;
    philox4x32_10((unsigned)i, 0u, 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), a, b);
    const double u = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);   // 53 bits in [0, 1)
    const double target = u * total_ptr[0];
    long long lo = 0, hi = n - 1;                      // first index with cdf > target
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (cdf[mid] > target) hi = mid; else lo = mid + 1;
    }
    idx[i] = (int)lo;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, long long n_rows, int row_len, const int* __restrict__ idx,
                                   int n_idx, float* __restrict__ dst) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)n_idx * row_len) return;
    const int i = (int)(e / row_len), j = (int)(e - (size_t)i * row_len);
    const long long r = idx[i];                      // never read outside the table: a bad index shows up as NaN
    dst[e] = (r >= 0 && r < n_rows) ? src[(size_t)r * row_len + j] : NAN;
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
    float *part = nullptr;     // partial column sums [2][RCHUNK][max(hidden, n_out)]
    float *part_b = nullptr;   // partial bias-gradient sums of the hidden layers [L][RCHUNK][hidden]
    float *loss_part = nullptr;   // per-block partial losses of l1_kernel
    float *split_ws = nullptr;    // partial products of the split-K weight-gradient GEMMs [splits][M][N]
    float *sign_count = nullptr;  // int[n_out]: l1_kernel's sign counts (allocated and zeroed with the float buffers)
    bool grad_dirty = false;   // a step that did not reach adam_kernel left grad non-zero
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

// C[m][n] = sum over the splits, in split order (deterministic)
__global__ __launch_bounds__(256) void split_reduce_kernel(int M, int N, int splits, const float* __restrict__ part,
                                                           float* __restrict__ C, int ldc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * N) return;
    float acc = 0.0f;
    for (int z = 0; z < splits; ++z) acc += part[(size_t)z * M * N + i];
    const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
    C[(size_t)m * ldc + n] = acc;
}

constexpr size_t SPLIT_WS_FLOATS = (size_t)256 * BM * BN;     // at most 255 blocks of 64 x 64 take part in a split GEMM

// split_k > 1: partial products go through split_ws (SPLIT_WS_FLOATS) and are summed in split order; no bias then
template <bool TA, bool TB>
void gemm(hipStream_t st, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
          const float* bias, int split_k = 1, float* split_ws = nullptr) {
    const bool aligned = (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (reinterpret_cast<uintptr_t>(B) % 16 == 0);
    const bool vec = aligned && M % BM == 0 && N % BN == 0 && K % BK == 0 && lda % 4 == 0 && ldb % 4 == 0;
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, split_k);
    // more splits for GEMMs with very few output tiles
    while (split_k > 1 && grid.x * grid.y * grid.z < 128 && (int)grid.z * 2 * BK <= K) grid.z *= 2;
    while (grid.z > 1 && (split_ws == nullptr || (size_t)grid.z * M * N > SPLIT_WS_FLOATS)) grid.z /= 2;
    if (vec) hipLaunchKernelGGL((gemm_kernel<TA, TB, true>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias, split_ws);
    else hipLaunchKernelGGL((gemm_kernel<TA, TB, false>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias, split_ws);
    if (grid.z > 1)
        hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, st, M, N, (int)grid.z,
                           split_ws, C, ldc);
}

constexpr int SPLIT_K = 4;    // of the weight-gradient GEMMs (contraction over the batch, 64 output tiles at hidden = 512)

unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

void colsum(Policy* p, hipStream_t st, int M, int H, const float* D, float* out) {
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((H + 63) / 64, RCHUNK), dim3(256), 0, st, M, H, D, p->part);
    hipLaunchKernelGGL(reduce_final_kernel, dim3((H + 255) / 256), dim3(256), 0, st, H, p->part, out, (float*)nullptr);
}

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
                hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((H + 63) / 64, RCHUNK), dim3(256), 0, st, B, H, z, p->part);
                hipLaunchKernelGGL(bn_relu_train_kernel, dim3((H + 63) / 64, RCHUNK), dim3(256), 0, st, B, H, z, p->part,
                                   p->theta + p->og[l], p->theta + p->obe[l], o, mu, inv,
                                   p->run_mean + (size_t)l * H, p->run_var + (size_t)l * H);
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
    if (dims->n_out > 256) return pfail(nullptr, NMPC_E_ARG, "n_out <= 256");
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
        {&p->pred, Bm * dims->n_out}, {&p->dpred, Bm * dims->n_out},
        {&p->part, (size_t)2 * RCHUNK * (size_t)(H > dims->n_out ? H : dims->n_out)},
        {&p->part_b, (size_t)L * RCHUNK * H}, {&p->loss_part, (Bm * dims->n_out + 255) / 256 + 1},
        {&p->sign_count, (size_t)dims->n_out}, {&p->split_ws, SPLIT_WS_FLOATS}};
    nmpc::DeviceGuard guard(device_id);
    hipError_t e = guard.err;
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
    nmpc::DeviceGuard guard(p->device);
    float* all[] = {p->theta, p->grad, p->m, p->v, p->run_mean, p->run_var, p->mu, p->inv, p->act, p->z,
                    p->dbuf[0], p->dbuf[1], p->pred, p->dpred, p->part, p->part_b, p->loss_part, p->sign_count, p->split_ws};
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
    nmpc::DeviceGuard guard(p->device);
    PTRY(p, guard.err);
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
    nmpc::DeviceGuard guard(p->device);
    PTRY(p, guard.err);
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
    nmpc::DeviceGuard guard(p->device);
    PTRY(p, guard.err);
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
    nmpc::DeviceGuard guard(p->device);
    PTRY(p, guard.err);
    const int L = p->d.n_hidden, H = p->d.hidden, no = p->d.n_out;
    const bool bn = p->d.batch_norm != 0;
    if (p->grad_dirty) {                                    // normally both are left zero by adam_kernel
        PTRY(p, hipMemsetAsync(p->grad, 0, p->n_theta * sizeof(float), st));
        PTRY(p, hipMemsetAsync(p->sign_count, 0, (size_t)no * sizeof(int), st));
    }
    p->grad_dirty = true;
    forward(p, B, X, p->pred, true, st);
    if (pred) PTRY(p, hipMemcpyAsync(pred, p->pred, (size_t)B * no * sizeof(float), hipMemcpyDeviceToDevice, st));
    const size_t np_ = (size_t)B * no;
    hipLaunchKernelGGL(l1_kernel, dim3(blocks_for(np_)), dim3(256), 0, st, np_, no, p->pred, Y, p->dpred, p->loss_part,
                       reinterpret_cast<int*>(p->sign_count));
    // output layer: dW = dP' a_L, db = colsum dP, d = dP W
    const float* aL = p->act + (size_t)(L - 1) * p->d.batch_max * H;
    gemm<true, true>(st, no, H, B, p->dpred, no, aL, H, p->grad + p->oW[L], H, nullptr, SPLIT_K, p->split_ws);
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
        hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3((H + 63) / 64, RCHUNK), dim3(256), 0, st, B, H, d, z, mu, inv,
                           bn ? p->theta + p->og[l] : nullptr, bn ? p->theta + p->obe[l] : nullptr, p->part);
        if (bn) {
            hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((H + 63) / 64, RCHUNK), dim3(256), 0, st, B, H, d, z, mu, inv,
                               p->theta + p->og[l], p->part, dgamma, dbeta, p->part_b + (size_t)l * RCHUNK * H);
        } else {
            hipLaunchKernelGGL(reduce_final_kernel, dim3((H + 255) / 256), dim3(256), 0, st, H, p->part, dbeta, (float*)nullptr);
        }
        gemm<true, true>(st, H, fan_in, B, d, H, a, fan_in, p->grad + p->oW[l], fan_in, nullptr, SPLIT_K, p->split_ws);
        if (l > 0) {
            gemm<false, true>(st, B, H, H, d, H, p->theta + p->oW[l], H, dn, H, nullptr);
            float* t = d; d = dn; dn = t;
        }
    }
    p->step += 1;
    const float c1 = 1.0f - std::pow(ADAM_B1, (float)p->step), c2 = 1.0f - std::pow(ADAM_B2, (float)p->step);
    AdamTail tail{};
    tail.part_b = bn ? p->part_b : nullptr;
    for (int l = 0; l < L; ++l) tail.ob[l] = p->ob[l];
    tail.L = L; tail.H = H;
    tail.loss_part = p->loss_part; tail.n_loss_part = (int)blocks_for(np_); tail.loss = loss;
    tail.sign_count = reinterpret_cast<int*>(p->sign_count); tail.ob_out = p->ob[L]; tail.n_out = no; tail.n_pred = (float)np_;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(p->n_theta)), dim3(256), 0, st, p->n_theta, p->theta, p->grad,
                       p->m, p->v, lr, c1, c2, tail);
    p->grad_dirty = false;
    PTRY(p, hipGetLastError());
    return NMPC_OK;
}

int nmpc_weighted_sample(const float* weights, long long n, int num_samples, unsigned long long seed, double* scratch,
                         int* idx, void* stream) {
    if (num_samples == 0) return NMPC_OK;
    if (!weights || !scratch || !idx) return pfail(nullptr, NMPC_E_ARG, "null argument");
    if (n < 1 || n > 0x7fffffffLL || num_samples < 0) return pfail(nullptr, NMPC_E_ARG, "need 1 <= n < 2^31, num_samples >= 0");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    double* cdf = scratch;
    double* tot = scratch + n;                        // nb + 1 doubles
    nmpc::DeviceGuard guard(nmpc::device_of(weights));
    hipLaunchKernelGGL(scan_block_totals_kernel, dim3((unsigned)nb), dim3(256), 0, st, weights, n, tot);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(64), 0, st, tot, nb);
    hipLaunchKernelGGL(scan_write_kernel, dim3((unsigned)nb), dim3(256), 0, st, weights, n, tot, cdf);
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((num_samples + 255) / 256)), dim3(256), 0, st, cdf, n, tot + nb,
                       num_samples, seed, idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pfail(nullptr, NMPC_E_HIP, hipGetErrorString(e));
    return NMPC_OK;
}

int nmpc_gather_rows(const float* src, long long n_rows, int row_len, const int* idx, int n_idx, float* dst, void* stream) {
    if (n_idx == 0) return NMPC_OK;
    if (!src || !idx || !dst) return pfail(nullptr, NMPC_E_ARG, "null argument");
    if (row_len < 1 || n_idx < 0 || n_rows < 1) return pfail(nullptr, NMPC_E_ARG, "need n_rows, row_len >= 1, n_idx >= 0");
    const size_t n = (size_t)n_idx * row_len;
    nmpc::DeviceGuard guard(nmpc::device_of(src));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, n_rows, row_len, idx, n_idx, dst);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pfail(nullptr, NMPC_E_HIP, hipGetErrorString(e));
    return NMPC_OK;
}

}  // extern "C"
