// Micro-benchmark (diagnostic): cost of v_fmac_f32 with a DPP row broadcast source against the
// v_readlane + v_fma pair it would replace, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define DPP(acc, x, r, I) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #I " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(r))
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* t, int n) {
    const int lane = threadIdx.x;
    float x = lane * 1e-3f, r = 1.0001f, a0 = 0.f, a1 = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {   // 12 DPP fmacs on two chains
        DPP(a0, x, r, 0); DPP(a1, x, r, 1); DPP(a0, x, r, 2); DPP(a1, x, r, 3); DPP(a0, x, r, 4); DPP(a1, x, r, 5);
        DPP(a0, x, r, 6); DPP(a1, x, r, 7); DPP(a0, x, r, 8); DPP(a1, x, r, 9); DPP(a0, x, r, 10); DPP(a1, x, r, 11);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float b0 = 0.f, b1 = 0.f;
    for (int i = 0; i < n; ++i) {   // 12 plain fmacs on two chains
#pragma unroll
        for (int j = 0; j < 6; ++j) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(b0) : "v"(x), "v"(r)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(b1) : "v"(x), "v"(r)); }
    }
    unsigned long long t2 = __builtin_readcyclecounter();
    float c0 = 0.f;
    for (int i = 0; i < n; ++i) {   // 12 readlane + fma pairs, one chain (as the old forward sweep)
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), j));
            c0 = fmaf(r, s, c0);
        }
        x += 1e-6f;
    }
    unsigned long long t3 = __builtin_readcyclecounter();
    float d0 = 0.f;
    for (int i = 0; i < n; ++i) {   // 12 DPP fmacs, ONE chain
        DPP(d0, x, r, 0); DPP(d0, x, r, 1); DPP(d0, x, r, 2); DPP(d0, x, r, 3); DPP(d0, x, r, 4); DPP(d0, x, r, 5);
        DPP(d0, x, r, 6); DPP(d0, x, r, 7); DPP(d0, x, r, 8); DPP(d0, x, r, 9); DPP(d0, x, r, 10); DPP(d0, x, r, 11);
    }
    unsigned long long t4 = __builtin_readcyclecounter();
    out[lane] = a0 + a1 + b0 + b1 + c0 + d0;
    if (lane == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; }
}
int main() {
    float* out; unsigned long long* t; const int n = 2000;
    hipMalloc(&out, 256); hipMalloc(&t, 64);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, t, n);
    hipDeviceSynchronize();
    unsigned long long h[4]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("per 12 ops: dpp-fmac two chains %.1f | plain fmac two chains %.1f | readlane+fma one chain %.1f | dpp-fmac one chain %.1f cycles\n",
           (double)h[0] / n, (double)h[1] / n, (double)h[2] / n, (double)h[3] / n);
    return 0;
}
