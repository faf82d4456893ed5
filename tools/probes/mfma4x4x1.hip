// Layout probe for v_mfma_f32_4x4x1_16b_f32 on gfx950 (sixteen 4x4 outer products per instruction):
// prints, for every lane and result register, which A lane and B lane the product came from.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    const float a = (float)(lane + 1), b = (float)(1000 * (lane + 1));
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[4 * lane + v] = d[v];
    // A broadcast: cbsz = 4 -> all sixteen blocks take the A operand of block `abid` (here 5: lanes 20..23)
    f32x4 e = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 5, 0);
    for (int v = 0; v < 4; ++v) out[256 + 4 * lane + v] = e[v];
}
int main() {
    float* d; hipMalloc(&d, 512 * sizeof(float));
    probe<<<1, 64>>>(d);
    float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int v = 0; v < 4; ++v) {
            const float expect = (float)(4 * (lane / 4) + v + 1) * (float)(1000 * (lane + 1));
            if (h[4 * lane + v] != expect) { if (bad < 8) printf("lane %d v %d: got %g expect %g\n", lane, v, h[4 * lane + v], expect); ++bad; }
        }
    printf("mismatches: %d (D[v][lane] = A[lane 4*(lane/4)+v] * B[lane])\n", bad);
    bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int v = 0; v < 4; ++v) {
            const float expect = (float)(4 * 5 + v + 1) * (float)(1000 * (lane + 1));
            if (h[256 + 4 * lane + v] != expect) { if (bad < 8) printf("bcast lane %d v %d: got %g expect %g\n", lane, v, h[256 + 4 * lane + v], expect); ++bad; }
        }
    printf("broadcast mismatches: %d (cbsz 4, abid 5: D[v][lane] = A[lane 20+v] * B[lane])\n", bad);
    return 0;
}
