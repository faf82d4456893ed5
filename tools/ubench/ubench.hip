// Micro-benchmarks (diagnostic, not part of the product): single-wave costs on gfx950 of the
// primitives the stage sweep is made of.  One wave per CU-SIMD like the B=1024 solve.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CLK() __builtin_readcyclecounter()

__global__ __launch_bounds__(64) void k(float* out, unsigned long long* t, int n) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x;
    float x = out[lane];
    f32x4 a = {x, x + 1, x + 2, x + 3}, b = {x * 2, x * 3, x * 4, x * 5}, c = {0, 0, 0, 0}, c2 = c, c3 = c, c4 = c;
    unsigned long long t0, t1;
    // 1. dependent MFMA chain
    t0 = CLK();
    for (int i = 0; i < n; ++i) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 0] = t1 - t0;
    // 2. four independent MFMA chains
    t0 = CLK();
    for (int i = 0; i < n; ++i) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c4, 0, 0, 0);
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 1] = t1 - t0;
    c += c2 + c3 + c4;
    // 3. dependent VALU fma chain (4 per iteration)
    float y = c[0];
    t0 = CLK();
    for (int i = 0; i < n / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) y = fmaf(y, 1.0001f, 0.5f);
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 2] = t1 - t0;
    // 4. independent VALU fma (4 chains)
    float y1 = y, y2 = y + 1, y3 = y + 2, y4 = y + 3;
    t0 = CLK();
    for (int i = 0; i < n / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            y1 = fmaf(y1, 1.0001f, 0.5f); y2 = fmaf(y2, 1.0001f, 0.5f); y3 = fmaf(y3, 1.0001f, 0.5f); y4 = fmaf(y4, 1.0001f, 0.5f);
        }
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 3] = t1 - t0;
    y = y1 + y2 + y3 + y4;
    // 5. readlane + fma pairs (4 per iteration), as in the LDL' update
    float z1 = y, z2 = y * 2, z3 = y * 3, z4 = y * 4;
    t0 = CLK();
    for (int i = 0; i < n / 8; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float l1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z1), 3));
        float l2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z2), 3));
        float l3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z3), 3));
        float l4 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z4), 3));
        z1 = fmaf(-l2, 0.001f, z1); z2 = fmaf(-l3, 0.001f, z2); z3 = fmaf(-l4, 0.001f, z3); z4 = fmaf(-l1, 0.001f, z4);
      }
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 4] = t1 - t0;
    y = z1 + z2 + z3 + z4;
    // 6. LDS round trip: write b128, read b128 from another lane's slot, dependent
    f32x4 v = {y, y, y, y};
    t0 = CLK();
    for (int i = 0; i < n; ++i) {
        *reinterpret_cast<f32x4*>(&lds[lane * 20]) = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        v = *reinterpret_cast<f32x4*>(&lds[((lane + 17) & 63) * 20]);
        v[0] += 1.0f;
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 5] = t1 - t0;
    // 7. MFMA result consumed by VALU then fed back (MFMA -> VALU -> MFMA latency)
    t0 = CLK();
    for (int i = 0; i < n; ++i) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c[0] = c[0] * 0.5f;
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 6] = t1 - t0;
    // 8. MFMA interleaved with 6 independent VALU each
    t0 = CLK();
    for (int i = 0; i < n / 8; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c2, 0, 0, 0);
        y1 = fmaf(y1, 1.0001f, 0.5f); y2 = fmaf(y2, 1.0001f, 0.5f); y3 = fmaf(y3, 1.0001f, 0.5f);
        y4 = fmaf(y4, 1.0001f, 0.5f); y1 = fmaf(y1, 1.0001f, 0.5f); y2 = fmaf(y2, 1.0001f, 0.5f);
      }
    }
    t1 = CLK();
    if (lane == 0) t[blockIdx.x * 8 + 7] = t1 - t0;
    out[blockIdx.x * 64 + lane] = c[0] + c[1] + c2[0] + v[0] + v[1] + y + y1 + y2 + y3 + y4;
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
    const char* names[8] = {"dependent MFMA 16x16x4f32 (per MFMA)", "4 independent MFMA chains (per MFMA)",
                            "dependent v_fma (per op)", "4 independent v_fma (per op)", "readlane+fma pair (per pair)",
                            "LDS b128 write->read round trip", "MFMA -> VALU -> MFMA (per iteration)",
                            "MFMA + 6 VALU interleaved (per iteration)"};
    const double div[8] = {4.0 * n, 4.0 * n, 4.0 * n, 4.0 * n, 4.0 * n, 1.0 * n, 1.0 * n, 1.0 * n};  // n = 1000: 125 outer iterations x 8 (x4)
    for (int j = 0; j < 8; ++j) {
        double s = 0;
        for (int b = 0; b < blocks; ++b) s += (double)h[b * 8 + j];
        printf("%-45s %8.1f cycles\n", names[j], s / blocks / div[j]);
    }
    return 0;
}
