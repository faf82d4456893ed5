// Micro-benchmark (diagnostic): does a bf16 MFMA overlap one wave's own VALU work on gfx950,
// and what does a 3-way bf16 split of an fp32 register cost?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define CLK() __builtin_readcyclecounter()

__device__ __forceinline__ unsigned short bf16_rn(float x) {   // round to nearest even
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ __launch_bounds__(64) void k(float* out, unsigned long long* t, int n) {
    const int lane = threadIdx.x;
    float x = out[lane] + 1.0f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (short)bf16_rn(x + i); b[i] = (short)bf16_rn(x * 0.5f + i); }
    f32x4 c = {0, 0, 0, 0}, c2 = c, c3 = c, c4 = c;
    float y1 = x, y2 = x + 1, y3 = x + 2, y4 = x + 3;
    unsigned long long t0, t1;
    t0 = CLK();
    for (int i = 0; i < n / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4, 0, 0, 0);
        }
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 0] = t1 - t0;   // /(4n)
    t0 = CLK();
    for (int i = 0; i < n / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 1] = t1 - t0;   // dependent, /(4n)
    t0 = CLK();
    for (int i = 0; i < n / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            y1 = fmaf(y1, 1.0001f, 0.5f); y2 = fmaf(y2, 1.0001f, 0.5f); y3 = fmaf(y3, 1.0001f, 0.5f);
            y4 = fmaf(y4, 1.0001f, 0.5f); y1 = fmaf(y1, 1.0001f, 0.5f); y2 = fmaf(y2, 1.0001f, 0.5f);
        }
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 2] = t1 - t0;   // MFMA + 6 VALU, /n
    t0 = CLK();
    for (int i = 0; i < n / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            y1 = fmaf(y1, 1.0001f, 0.5f); y2 = fmaf(y2, 1.0001f, 0.5f); y3 = fmaf(y3, 1.0001f, 0.5f);
            y4 = fmaf(y4, 1.0001f, 0.5f); y1 = fmaf(y1, 1.0001f, 0.5f); y2 = fmaf(y2, 1.0001f, 0.5f);
        }
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 3] = t1 - t0;   // 6 VALU, /n
    // split cost: 4 floats -> hi, mid, lo bf16 (12 values)
    f32x4 v = {y1, y2, y3, y4};
    unsigned acc = 0;
    t0 = CLK();
    for (int i = 0; i < n / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned short h = bf16_rn(v[r]);
                const float r1 = v[r] - __uint_as_float((unsigned)h << 16);
                const unsigned short m = bf16_rn(r1);
                const float r2 = r1 - __uint_as_float((unsigned)m << 16);
                const unsigned short l = bf16_rn(r2);
                acc += h + m + l;
                v[r] += 1.0f;
            }
        }
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 4] = t1 - t0;   // per tile (4 floats), /n
    out[blockIdx.x * 64 + lane] = c[0] + c2[0] + c3[1] + c4[2] + y1 + y2 + y3 + y4 + (float)acc + v[0];
}

int main() {
    const int blocks = 1024, n = 1000;
    float* out; unsigned long long* t;
    hipMalloc(&out, blocks * 64 * sizeof(float));
    hipMemset(out, 0, blocks * 64 * sizeof(float));
    hipMalloc(&t, blocks * 8 * sizeof(unsigned long long));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, t, n);
    hipDeviceSynchronize();
    static unsigned long long h[1024 * 8];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[5] = {"4 independent bf16 16x16x32 MFMA (per MFMA)", "dependent bf16 MFMA (per MFMA)",
                            "bf16 MFMA + 6 v_fma (per iteration)", "6 v_fma alone (per iteration)",
                            "3-way bf16 split of 4 floats (per tile)"};
    const double div[5] = {4.0 * n, 4.0 * n, 1.0 * n, 1.0 * n, 1.0 * n};
    for (int j = 0; j < 5; ++j) {
        double s = 0;
        for (int b = 0; b < blocks; ++b) s += (double)h[b * 8 + j];
        printf("%-48s %8.1f cycles\n", names[j], s / blocks / div[j]);
    }
    return 0;
}
