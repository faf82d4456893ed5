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

// Diagnostic builds (-DNMPC_STAMPS): cycle counters of the segments of one backward stage.
#ifdef NMPC_STAMPS
struct StageStamps { unsigned long long t0; unsigned long long acc[8]; };
#define SST_ARG , StageStamps& sst
#define SST_BEGIN sst.t0 = __builtin_readcyclecounter()
#define SST(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t1_ = __builtin_readcyclecounter(); \
                    sst.acc[i] += t1_ - sst.t0; sst.t0 = t1_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SST_ARG
#define SST_BEGIN
#define SST(i)
#endif

// Per-lane constants of the stage sweeps (computed once per kernel).
struct SweepLane {
    float* wb;        // write-back column of the eliminated system (groups 0/3 write to a dummy tile)
    const float* rd;  // column this lane eliminates (Huu | H~ux), dummy for groups 2/3
    bool from_lds;    // group < 2: column comes from the conversion tiles
    int ident_row;    // group 2: row of the 1 in the identity right-hand side, else -1
    bool corner[4];   // true at the (hx,hx) corner of a tile
    float rs_free[16]; // 1/sqrt(R_jj + reg): scale of an uncoupled input's row (wave-uniform)
    __device__ __forceinline__ void init(float* conv, int lane, int hx) {
        const int t = lane >> 4, c = lane & 15;
        float* T0 = conv;
        float* T1 = conv + CTILE;
        float* Td = conv + 2 * CTILE;
        wb = ((t == 1) ? T1 : (t == 2) ? T0 : Td) + c * LDC;
        rd = ((t == 0) ? T0 : (t == 1) ? T1 : Td) + c * LDC;
        from_lds = t < 2;
        ident_row = (t == 2) ? c : -1;
#pragma unroll
        for (int r = 0; r < 4; ++r) corner[r] = (c == hx && 4 * t + r == hx);
    }
};

// One backward stage on the critical path:  P~_{k+1} -> P~_k, and the factors W, Y of stage k.
//   Aa, Ba: A~_k, B~_k.  Qt, St, Rt: additive cost tiles (Q~, S~, R incl. barrier terms).
//   conv: 3*CTILE floats of LDS.  MASK: compile-time coupling mask of the inputs, or DYNAMIC_MASK
//   with the run-time mask in `coupled` (branchy fallback).
//   nc: the cost tiles of stage k-1, which do not depend on this stage: fetch() reads their LDS
//   operands right behind this stage's column loads, mfma(i) is K step i of their barrier product,
//   finish() folds the product into R and S~.  The wave is alone on its SIMD, so this is the only
//   latency hiding there is: independent work placed where the stage waits anyway.
// Outputs (accumulator layout): W = D^-1/2 L^-1, Y = D^-1/2 L^-1 H~ux; K~ = -W'Y is left to the
// caller.  Returns false on a non-positive pivot.
template <int NU, unsigned MASK, class NextCost>
__device__ __forceinline__ bool backward_stage(f32x4& P, f32x4 Aa, f32x4 Ba, f32x4 Qt, f32x4 St, f32x4 Rt,
                                               float* conv, const SweepLane& sl, int lane, unsigned coupled,
                                               f32x4& Wout, f32x4& Yout, NextCost& nc SST_ARG) {
    SST_BEGIN;
    const f32x4 PA = xty(P, Aa);
    const f32x4 PB = xty(P, Ba);
    const f32x4 Hux = xty(Ba, PA, St);
    const f32x4 Huu = xty(Ba, PB, Rt);
    SST(0);
    lds_store_acc(conv, lane, Huu);
    lds_store_acc(conv + CTILE, lane, Hux);
    wave_sync();

    // column layout: lane L -> column (L&15) of tile (L>>4): Huu | H~ux | I | unused
    float col[NU];
#pragma unroll
    for (int i4 = 0; i4 < (NU + 3) / 4; ++i4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sl.rd + 4 * i4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * i4 + r;
            if (i < NU) col[i] = sl.from_lds ? v[r] : ((i == sl.ident_row) ? 1.0f : 0.0f);
        }
    }
    nc.fetch();
    SST(1);
    // A~'(P~A~) and its bitwise transpose (P~A~)'A~ (same products, same k order): their mean is
    // exactly symmetric, which keeps P~ symmetric over the whole recursion (the tile algebra
    // uses P~ as its own transpose).
    f32x4 H1 = Qt, H2 = Qt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        H1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Aa[i], PA[i], H1, 0, 0, 0);
        H2 = __builtin_amdgcn_mfma_f32_16x16x4f32(PA[i], Aa[i], H2, 0, 0, 0);
        nc.mfma(i);
    }
    float rsf[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) rsf[j] = sl.rs_free[j];
    const bool ok = ldl_eliminate<NU, MASK>(col, coupled, rsf);
    nc.finish();
    const f32x4 Hxx = 0.5f * (H1 + H2);
    SST(2);
    wave_sync();
#pragma unroll
    for (int i4 = 0; i4 < (NU + 3) / 4; ++i4) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (4 * i4 + r < NU) ? col[(4 * i4 + r < NU) ? 4 * i4 + r : 0] : 0.0f;
        *reinterpret_cast<f32x4*>(sl.wb + 4 * i4) = v;
    }
    wave_sync();
    const f32x4 Y = mask_rows(lds_load_acc(conv + CTILE, lane), lane, NU);
    const f32x4 W = mask_rows(lds_load_acc(conv, lane), lane, NU);
    wave_sync();
    SST(3);
    f32x4 Pn = xty2(-Y, Y, Hxx);
    // the constant term of the cost-to-go (corner hx,hx) feeds nothing: keep it at zero
#pragma unroll
    for (int r = 0; r < 4; ++r) Pn[r] = sl.corner[r] ? 0.0f : Pn[r];
    P = Pn;
    Wout = W;
    Yout = Y;
    SST(4);
    return ok;
}

// NextCost of a caller that builds its cost tiles itself (dense-LQ kernel).
struct NoNextCost {
    __device__ __forceinline__ void fetch() {}
    __device__ __forceinline__ void mfma(int) {}
    __device__ __forceinline__ void finish() {}
};

// K~ = -W'Y and Acl~ = A~ + B~K~ of a finished stage (Bt = B~').
__device__ __forceinline__ void gain_tiles(f32x4 W, f32x4 Y, f32x4 Aa, f32x4 Bt, f32x4& K, f32x4& Acl) {
    K = xty(W, -Y);
    Acl = xty(Bt, K, Aa);
}

}  // namespace nmpc
