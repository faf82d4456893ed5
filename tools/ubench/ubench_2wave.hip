// Micro-benchmark (diagnostic, not part of the product): do the fp32 MFMA pipe and the VALU of one
// SIMD overlap when the work comes from TWO waves?  One 512-thread block per CU: waves w and w+4
// share a SIMD (waves are dealt round-robin).  Modes per wave half (lo = waves 0..3, hi = 4..7):
//   0 idle   1 MFMA chain (4 independent accumulators)   2 VALU fma (8 independent chains)
//   3 mixed: 4 MFMA then 16 VALU per iteration (like a Riccati stage)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void work(int mode, int n, float* out) {
    const int lane = threadIdx.x & 63;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = lane * 1e-3f, y = 1.0001f;
    float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
    if (mode == 1) {
        for (int i = 0; i < n; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        }
    } else if (mode == 2) {
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                v0 = fmaf(v0, y, x); v1 = fmaf(v1, y, x); v2 = fmaf(v2, y, x); v3 = fmaf(v3, y, x);
                v4 = fmaf(v4, y, x); v5 = fmaf(v5, y, x); v6 = fmaf(v6, y, x); v7 = fmaf(v7, y, x);
            }
        }
    } else if (mode == 3) {
        for (int i = 0; i < n; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                v0 = fmaf(v0, y, x); v1 = fmaf(v1, y, x); v2 = fmaf(v2, y, x); v3 = fmaf(v3, y, x);
                v4 = fmaf(v4, y, x); v5 = fmaf(v5, y, x); v6 = fmaf(v6, y, x); v7 = fmaf(v7, y, x);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

__global__ __launch_bounds__(512) void k(float* out, unsigned long long* t, int n, int mode_lo, int mode_hi) {
    const int w = threadIdx.x >> 6;
    const int mode = (w < 4) ? mode_lo : mode_hi;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    work(mode, n, out);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 8 + w] = t1 - t0;
}

int main() {
    float* out; unsigned long long* t;
    const int blocks = 256, n = 2000;
    hipMalloc(&out, blocks * 512 * sizeof(float));
    hipMalloc(&t, blocks * 8 * sizeof(unsigned long long));
    const int cases[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {1, 2}, {3, 3}, {3, 1}, {3, 2}};
    for (auto& c : cases) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, t, n, c[0], c[1]);
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        printf("lo=%d hi=%d : cycles/iter  lo %.1f  hi %.1f\n", c[0], c[1], (double)h[0] / n, (double)h[4] / n);
    }
    return 0;
}
