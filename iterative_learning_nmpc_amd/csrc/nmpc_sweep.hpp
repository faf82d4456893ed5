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

// LDS conversion area of the stage sweeps: four tiles of CTILE floats
//   T0: Huu, later W     T1: H~ux, later Y     Td: write-only sink     Ti: identity (read-only)
constexpr int CONV_TILES = 4;
constexpr int CONV_FLOATS = CONV_TILES * CTILE + 16;   // + rs_free[16], see SweepLane

// Per-lane constants of the stage sweeps (computed once per kernel).
struct SweepLane {
    float* wb;        // write-back column of the eliminated system (groups 0/3 write to the sink Td)
    const float* rd;  // column this lane eliminates: Huu | H~ux | I | I
    bool corner[4];   // true at the (hx,hx) corner of a tile
    float hs_col;     // 1 in the lanes of the homogeneous column (slot layout), else 0
    // 1/sqrt(R_jj + reg), the scale of an uncoupled input's row: sixteen wave-uniform constants, as
    // scalars (SGPRs) and in the LDS behind the tiles (fetched as quads with the column reads).  Which
    // copy a stage body uses is its RSF_LDS parameter: scalars are cheaper while they fit; the kernel
    // with all contact patterns spilled them (measured +1.3 % with the LDS copy there, -0.7 % here).
    const float* rs_free;
    float rs_free_s[16];
    // identity columns of the in-register layout conversion (backward_stage): what lane rows 2, 3 (the identity and
    // sink groups of the column layout) receive through the half swaps -- idx[r]: slots r and 8 + r, idy[r]: 4 + r, 12 + r
    float idx[4], idy[4];
    // fills the identity tile as well: the caller orders it (wave_sync) before the first stage
    __device__ __forceinline__ void init(float* conv, int lane, int hx, const float* rs_free_values) {
        const int t = lane >> 4, c = lane & 15;
        float* T0 = conv;
        float* T1 = conv + CTILE;
        float* Td = conv + 2 * CTILE;
        float* Ti = conv + 3 * CTILE;
        wb = ((t == 1) ? T1 : (t == 2) ? T0 : Td) + c * LDC;
        rd = ((t == 0) ? T0 : (t == 1) ? T1 : Ti) + c * LDC;
        hs_col = (c == HS) ? 1.0f : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            idx[r] = (c == (t < 2 ? r : 8 + r)) ? 1.0f : 0.0f;
            idy[r] = (c == (t < 2 ? 4 + r : 12 + r)) ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) corner[r] = (c == hx && 4 * t + r == hx);
        for (int i = lane; i < CTILE; i += 64) Ti[i] = (i % LDC == i / LDC) ? 1.0f : 0.0f;
        float* rsf = conv + CONV_TILES * CTILE;
        if (lane < 16) rsf[lane] = rs_free_values ? rs_free_values[lane] : 1.0f;
        rs_free = rsf;
#pragma unroll
        for (int j = 0; j < 16; ++j) rs_free_s[j] = rs_free_values ? rs_free_values[j] : 1.0f;
    }
};

// C + X'Y over the contraction steps 0..STEPS-1, split over two accumulators (see xty2)
template <int STEPS>
__device__ __forceinline__ void xty_pair(f32x4 X0, f32x4 Y0, f32x4& C0, f32x4 X1, f32x4 Y1, f32x4& C1) {
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        C0 = mfma4(X0[i], Y0[i], C0);
        C1 = mfma4(X1[i], Y1[i], C1);
    }
}

// Half exchanges between lane rows (16 lanes each) of two registers -- gfx950's v_permlane16_swap / v_permlane32_swap:
//   swap16(A, B): A.row1 <-> B.row0, A.row3 <-> B.row2        swap32(A, B): A.rows{2,3} <-> B.rows{0,1}
// With them the layout changes around the elimination can stay in registers instead of going through the LDS (CONV):
//   bit 0, accumulator layout -> column layout: register r of tiles T0 (Huu), T1 (H~ux) at lane row q is row 4q + r of
//   column c; column register D[4q + r] wants T0's value in lane row 0 and T1's in lane row 1 (rows 2, 3: identity columns):
//       swap16(T0[r], T1[r]) = X [t0_0 t1_0 t0_2 t1_2], Y [t0_1 t1_1 t0_3 t1_3];  swap32(X, idx) -> D[r], D[8 + r];
//       swap32(Y, idy) -> D[4 + r], D[12 + r]
//   bit 1, column layout -> accumulator layout (Y from lane row 1, W from lane row 2), a..d = D[r], D[4+r], D[8+r], D[12+r]:
//       swap32(a, c), swap32(b, d), then swap16(a, b).second = Y[r] and swap16(c, d).first = W[r].
// Measured (solves/s at B = 1024 one wave per SIMD / B = 8192 two waves): LDS both ways 2.20 M / 2.72 M; bit 0 alone 2.19 / 2.66;
// bit 1 alone 2.25 / 2.70; both 2.22 / 2.63.  The first conversion's LDS latency hides behind the A~'P~A~ MFMAs and the swaps
// are VALU time, which two waves per SIMD compete for; the second one is an exposed round trip that twelve swaps replace.  So:
// CONV = 2 for the resident variant of the QP kernel, 0 (LDS) for the lean one and the dense LQ kernel.
__device__ __forceinline__ void swap16(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap32(float& a, float& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}

// One backward stage on the critical path:  P~_{k+1} -> P~_k, and the gain tiles of stage k.
//   Aa, Ba, Bt: A~_k, B~_k, B~_k'.  Qt, St, Rt: additive cost tiles (Q~, S~, R incl. barrier terms).
//   conv: CONV_TILES*CTILE floats of LDS.  MASK: compile-time coupling mask of the inputs, or
//   DYNAMIC_MASK with the run-time mask in `coupled` (branchy fallback).
//   SLOT3: the tiles are in the slot layout (nmpc_tile.hpp) with the homogeneous coordinate at HS:
//   every product contracts over three steps; the one row the fourth step would add (row HS of
//   A~ = e_HS) is patched in with VALU adds.  Without it (dense-LQ entry) rows and columns are in
//   natural order and products take four steps.
//   NU counts logical inputs; column j of the elimination sits in lane slot_of(j) under SLOT3.
//   nc: the cost tiles of stage k-1, which do not depend on this stage: fetch() issues their LDS
//   reads, build() forms the operands, mfma(i) is K step i of their barrier product, finish() folds
//   the product into R and S~.
// The wave is alone on its SIMD, so nothing but the order of this code hides latency:
//   * chains of dependent MFMAs are issued in pairs (a dependent fp32 MFMA waits 44 cycles, an
//     independent one issues after 32);
//   * the MFMAs of A~'P~A~ sit between the LDS reads (columns, next cost operands) and their
//     first use, fenced so that the scheduler keeps them there.
// Padding rows of Huu, H~ux are exact zeros (B~ has no such columns, R and the barrier no such
// rows), so W and Y come back from the LDS with zero padding and need no masking.
// Outputs (accumulator layout): K~ = -W'Y, Acl~ = A~ + B~K~  with  W = D^-1/2 L^-1,
// Y = D^-1/2 L^-1 H~ux.  Returns false on a non-positive pivot.
template <int NU, unsigned MASK, bool SLOT3, bool RSF_LDS, unsigned NEXT_STEPS = 0xFu, int CONV = 0, class NextCost>
__device__ __forceinline__ bool backward_stage(f32x4& P, f32x4 Aa, f32x4 Ba, f32x4 Bt, f32x4 Qt, f32x4 St,
                                               f32x4 Rt, float* conv, const SweepLane& sl, int lane,
                                               unsigned coupled, f32x4& Kout, f32x4& Aclout,
                                               NextCost& nc SST_ARG) {
    constexpr int STEPS = SLOT3 ? 3 : 4;
    SST_BEGIN;
    f32x4 PA = zero4(), PB = zero4();
    xty_pair<STEPS>(P, Aa, PA, P, Ba, PB);
    if constexpr (SLOT3) PA += sl.hs_col * P;        // row HS of A~ is e_HS: P~A~[:, HS] += P~[HS, :]' (P~ symmetric)
    f32x4 Hux = St, Huu = Rt;
    xty_pair<STEPS>(Ba, PA, Hux, Ba, PB, Huu);       // row HS of B~ is zero
    SST(0);
    // column layout: lane L -> column (L&15) of tile (L>>4): Huu | H~ux | I | I
    constexpr int NQ = SLOT3 ? slot_of(NU - 1) / 4 + 1 : (NU + 3) / 4;
    f32x4 cq[4];
    if constexpr (!(CONV & 1)) {     // through the conversion tiles of the LDS
        lds_store_acc(conv, lane, Huu);
        lds_store_acc(conv + CTILE, lane, Hux);
        wave_sync();
#pragma unroll
        for (int i4 = 0; i4 < NQ; ++i4) cq[i4] = *reinterpret_cast<const f32x4*>(sl.rd + 4 * i4);
    } else {
#pragma unroll
        for (int r = 0; r < STEPS; ++r) {
            float x = Huu[r], y = Hux[r], zx = sl.idx[r], zy = sl.idy[r];
            swap16(x, y);
            swap32(x, zx);
            swap32(y, zy);
            cq[0][r] = x; cq[1][r] = y; cq[2][r] = zx; cq[3][r] = zy;
        }
    }
    // scales of the uncoupled rows (static masks with such rows only)
    constexpr bool FREE_ROWS = RSF_LDS && (MASK != DYNAMIC_MASK) && ((~MASK & ((NU < 32) ? ((1u << NU) - 1u) : ~0u)) != 0u);
    f32x4 rq[(NU + 3) / 4];
    if constexpr (FREE_ROWS) {
#pragma unroll
        for (int i4 = 0; i4 < (NU + 3) / 4; ++i4) rq[i4] = *reinterpret_cast<const f32x4*>(sl.rs_free + 4 * i4);
    }
    nc.fetch();
    // (no scheduling fence here: measured, the compiler's own placement of these LDS reads relative to the MFMAs
    // below is 0.8 % faster per solve call than pinning them in front; the fence behind the MFMAs stays)
    // H = A~'(P~A~): in exact arithmetic symmetric, in fp32 not quite -- and the tile algebra uses P~
    // as its own transpose, so the asymmetry would grow along the recursion.  H~xx = (H + H')/2 is
    // symmetric bit for bit; H' comes back through the sink tile of the LDS (the reads are covered
    // by the elimination).
    f32x4 H = Qt;
#pragma unroll
    for (int i = 0; i < STEPS; ++i) H = mfma4(Aa[i], PA[i], H);
    if constexpr (SLOT3) H[3] += PA[3];              // row HS of A~: H[HS, :] += (P~A~)[HS, :]
    __builtin_amdgcn_sched_barrier(0);
    float* Td = conv + 2 * CTILE;
    lds_store_acc(Td, lane, H);
    wave_sync();
    const f32x4 Ht = lds_load_acc_t(Td, lane);
    SST(1);
    float col[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) col[i] = SLOT3 ? cq[i / 3][i % 3] : cq[i >> 2][i & 3];
    nc.build();
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if ((NEXT_STEPS >> i) & 1u) nc.mfma(i);      // steps whose rows are all inactive in the next stage add exact zeros
    float rsf[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) rsf[j] = FREE_ROWS ? rq[j >> 2][j & 3] : sl.rs_free_s[j];
    const bool ok = ldl_eliminate<NU, MASK, SLOT3>(col, coupled, rsf);
    nc.finish();
    const f32x4 Hxx = 0.5f * (H + Ht);
    SST(2);
    f32x4 Y = zero4(), W = zero4();
    if constexpr (!(CONV & 2)) {
        wave_sync();
#pragma unroll
        for (int i4 = 0; i4 < NQ; ++i4) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = SLOT3 ? 3 * i4 + r : 4 * i4 + r;
                const bool used = (i < NU) && !(SLOT3 && r == 3);
                v[r] = used ? col[used ? i : 0] : 0.0f;
            }
            *reinterpret_cast<f32x4*>(sl.wb + 4 * i4) = v;
        }
        wave_sync();
        Y = lds_load_acc(conv + CTILE, lane);
        W = lds_load_acc(conv, lane);
        wave_sync();
    } else {
#pragma unroll
        for (int r = 0; r < STEPS; ++r) {
            float d4[4];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int i = SLOT3 ? 3 * i4 + r : 4 * i4 + r;
                d4[i4] = (i4 < NQ && i < NU) ? col[(i4 < NQ && i < NU) ? i : 0] : 0.0f;
            }
            swap32(d4[0], d4[2]);
            swap32(d4[1], d4[3]);
            swap16(d4[0], d4[1]);
            swap16(d4[2], d4[3]);
            Y[r] = d4[1];
            W[r] = d4[2];
        }
    }
    SST(3);
    // P~+ = H~xx - Y'Y on two accumulators, interleaved with the chain of K~ = -W'Y
    const f32x4 nY = -Y;
    f32x4 Pa = Hxx, Pb = zero4(), K = zero4();
    Pa = mfma4(nY[0], Y[0], Pa);
    Pb = mfma4(nY[1], Y[1], Pb);
    K = mfma4(W[0], nY[0], K);
    Pa = mfma4(nY[2], Y[2], Pa);
    K = mfma4(W[1], nY[1], K);
    if constexpr (!SLOT3) Pb = mfma4(nY[3], Y[3], Pb);
    K = mfma4(W[2], nY[2], K);
    f32x4 Pn = Pa + Pb;
    if constexpr (!SLOT3) K = mfma4(W[3], nY[3], K);
    // the constant term of the cost-to-go (corner hx,hx) feeds nothing: keep it at zero
#pragma unroll
    for (int r = 0; r < 4; ++r) Pn[r] = sl.corner[r] ? 0.0f : Pn[r];
    P = Pn;
    SST(4);
    f32x4 Ca = Aa, Cb = zero4();
    Ca = mfma4(Bt[0], K[0], Ca);
    Cb = mfma4(Bt[1], K[1], Cb);
    Ca = mfma4(Bt[2], K[2], Ca);
    if constexpr (!SLOT3) Cb = mfma4(Bt[3], K[3], Cb);
    Kout = K;
    Aclout = Ca + Cb;
    return ok;
}

// NextCost of a caller that builds its cost tiles itself (dense-LQ kernel).
struct NoNextCost {
    __device__ __forceinline__ void fetch() {}
    __device__ __forceinline__ void build() {}
    __device__ __forceinline__ void mfma(int) {}
    __device__ __forceinline__ void finish() {}
};

}  // namespace nmpc
