// nmpc_sweep.hpp -- one Riccati stage (backward) and one rollout stage (forward) on tiles.
//
// Homogeneous form: with x~ = [dx; 1] every affine term rides inside the 16x16 tiles
//   A~ = [A d; 0 1]   B~ = [B; 0]   P~ = [P p; p' *]   Q~ = [Q q; q' 0]   S~ = [S r]   K~ = [K kff]
// so gradients, defects and feed-forward terms need no separate vector code.
//   H~xx = Q~ + A~'P~A~     H~ux = S~ + B~'P~A~     Huu = R + B~'P~B~
//   LDL' of Huu applied to [H~ux | I]  ->  Y, W  (nmpc_tile.hpp)
//   K~ = -W'Y      P~+ = H~xx - Y'Y
// (the reference reaches the same recursion through HPIPM).
#pragma once
#include "nmpc_tile.hpp"

namespace nmpc {

// Zero the rows >= n of an accumulator-layout tile.
__device__ __forceinline__ f32x4 mask_rows(f32x4 v, int lane, int n) {
    const int q = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (4 * q + r < n) ? v[r] : 0.0f;
    return v;
}

// One backward stage.  P: P~ of stage k+1 (in/out: P~ of stage k).  Aa, Ba: A~_k, B~_k; Bt = B~_k'.
// Qt, St, Rt: additive cost tiles (Q~, S~, R incl. barrier terms).  conv: 2*CTILE floats of LDS.
// hx: index of the homogeneous coordinate (= nx).  Outputs in accumulator layout:
// Kout = K~_k, Aclout = A~_k + B~_k K~_k (closed loop).  Returns false on a non-positive pivot.
template <int NU>
__device__ __forceinline__ bool backward_stage(f32x4& P, f32x4 Aa, f32x4 Ba, f32x4 Bt, f32x4 Qt,
                                               f32x4 St, f32x4 Rt, float* conv, int lane, int hx,
                                               f32x4& Kout, f32x4& Aclout) {
    const f32x4 PA = xty(P, Aa);
    const f32x4 PB = xty(P, Ba);
    // A~'(P~A~) and its bitwise transpose (P~A~)'A~ (same products, same k order): their mean is
    // exactly symmetric, which keeps P~ symmetric over the whole recursion (the tile algebra
    // uses P~ as its own transpose).
    const f32x4 Hxx = 0.5f * (xty(Aa, PA, Qt) + xty(PA, Aa, Qt));
    const f32x4 Hux = xty(Ba, PA, St);
    const f32x4 Huu = xty(Ba, PB, Rt);

    float* T0 = conv;
    float* T1 = conv + CTILE;
    lds_store_acc(T0, lane, Huu);
    lds_store_acc(T1, lane, Hux);
    wave_sync();

    // column layout: lane L -> column (L&15) of tile (L>>4): Huu | H~ux | I | unused
    const int t = lane >> 4, c = lane & 15;
    const float* src = ((t & 1) ? T1 : T0) + c * LDC;
    float col[NU];
#pragma unroll
    for (int i4 = 0; i4 < (NU + 3) / 4; ++i4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * i4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * i4 + r;
            if (i < NU) col[i] = (t < 2) ? v[r] : ((t == 2 && i == c) ? 1.0f : 0.0f);
        }
    }
#ifndef NMPC_EXP_NOELIM
    const bool ok = ldl_eliminate<NU>(col);
#else
    const bool ok = true;
#endif
    wave_sync();
    if (t == 1 || t == 2) {
        float* dst = ((t == 1) ? T1 : T0) + c * LDC;
#pragma unroll
        for (int i4 = 0; i4 < (NU + 3) / 4; ++i4) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (4 * i4 + r < NU) ? col[(4 * i4 + r < NU) ? 4 * i4 + r : 0] : 0.0f;
            *reinterpret_cast<f32x4*>(dst + 4 * i4) = v;
        }
    }
    wave_sync();
    const f32x4 Y = mask_rows(lds_load_acc(T1, lane), lane, NU);
    const f32x4 W = mask_rows(lds_load_acc(T0, lane), lane, NU);
    wave_sync();

    const f32x4 nY = -Y;
    Kout = xty(W, nY);
    f32x4 Pn = xty(nY, Y, Hxx);
    // the constant term of the cost-to-go (corner hx,hx) feeds nothing: keep it at zero
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (c == hx && 4 * (lane >> 4) + r == hx) Pn[r] = 0.0f;
    P = Pn;
    Aclout = xty(Bt, Kout, Aa);   // (B~')' K~ + A~
    return ok;
}

}  // namespace nmpc
