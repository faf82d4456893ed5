// nmpc_tile.hpp -- 16x16 fp32 tile algebra for one wavefront (gfx950 / CDNA4).
//
// One NMPC problem lives in one 64-lane wavefront.  Every matrix of the Riccati recursion
// (nx+1 <= 16, nu <= 16) is a 16x16 fp32 tile held in the *accumulator layout* of
// v_mfma_f32_16x16x4_f32: lane l = 16*q + c owns rows 4q..4q+3 of column c (4 VGPRs).
//
// The only product needed is  X'Y  and the accumulator layout feeds it with no data movement:
// with the K index permuted as k = 4q + s (s = MFMA step), lane (q,c) register s of a tile T is
// T[4q+s][c], which is exactly the B-operand of step s for "...* T" and the A-operand of step s
// for "T' * ...".  So   xty(X, Y) = X'Y   is four MFMAs on registers the wave already holds.
// (fp32-input MFMA is exact fp32, a k-ordered fmaf chain: cdna_hip_programming.md section 3.)
#pragma once
#include <hip/hip_runtime.h>

namespace nmpc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TS = 16;        // tile side
constexpr int TILE = TS * TS; // floats per tile (column-major image: element (i,c) at c*16+i)
constexpr int LDC = 20;       // LDS column stride (floats) of a conversion tile: 80 B keeps
                              // ds_read_b128 of 16 different columns conflict-free
constexpr int CTILE = TS * LDC;

// ---- slot layout of the model tiles ---------------------------------------------------------------
// A lane owns a quad of rows, and the K index of xty() runs over (quad, register): contraction step
// s takes register s of every quad.  State and input indices n < 12 are therefore placed at
//   slot(n) = 4*(n/3) + n%3        (registers 0..2 of quad n/3)
// and register 3 of every quad is padding -- except slot 3, the homogeneous coordinate.  A product
// that contracts over states or inputs then needs steps 0..2 only: 3 MFMAs instead of 4 (the
// homogeneous row of A~ = e_HS is added in the VALU, nmpc_sweep.hpp).
constexpr int HS = 3;
__host__ __device__ constexpr int slot_of(int n) { return 4 * (n / 3) + n % 3; }
__host__ __device__ constexpr int index_of(int slot) { return (slot & 3) == 3 ? -1 : 3 * (slot >> 2) + (slot & 3); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- bf16 operands (mixed-precision barrier product, nmpc_dims.precision = 1) --------------------
// The accumulator layout of a 16-row tile is also the operand layout of v_mfma_f32_16x16x16_bf16:
// lane (q, c) holds k = 4q..4q+3 of row/column c.  X'Y over all 16 rows is then ONE instruction on
// the bf16 matrix pipe instead of four fp32 steps.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x4 to_bf16x4(f32x4 v) {       // two v_cvt_pk_bf16_f32 (round to nearest even)
    const bf16x2 lo = __builtin_convertvector((f32x2){v[0], v[1]}, bf16x2);
    const bf16x2 hi = __builtin_convertvector((f32x2){v[2], v[3]}, bf16x2);
    const u32x2 p = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    return __builtin_bit_cast(s16x4, p);
}
__device__ __forceinline__ f32x4 xty_bf16(f32x4 X, f32x4 Y) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(to_bf16x4(X), to_bf16x4(Y), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}

// C + X'Y  (all three in accumulator layout)
__device__ __forceinline__ f32x4 xty(f32x4 X, f32x4 Y, f32x4 C) {
    C = __builtin_amdgcn_mfma_f32_16x16x4f32(X[0], Y[0], C, 0, 0, 0);
    C = __builtin_amdgcn_mfma_f32_16x16x4f32(X[1], Y[1], C, 0, 0, 0);
    C = __builtin_amdgcn_mfma_f32_16x16x4f32(X[2], Y[2], C, 0, 0, 0);
    C = __builtin_amdgcn_mfma_f32_16x16x4f32(X[3], Y[3], C, 0, 0, 0);
    return C;
}
__device__ __forceinline__ f32x4 xty(f32x4 X, f32x4 Y) { return xty(X, Y, zero4()); }
// the same over contraction steps 0 and 1 only (registers 2, 3 of one operand are exact zeros in every lane)
__device__ __forceinline__ f32x4 xty01(f32x4 X, f32x4 Y, f32x4 C) {
    C = __builtin_amdgcn_mfma_f32_16x16x4f32(X[0], Y[0], C, 0, 0, 0);
    C = __builtin_amdgcn_mfma_f32_16x16x4f32(X[1], Y[1], C, 0, 0, 0);
    return C;
}

// Same product with the K steps split over two accumulators: half the dependent-MFMA depth (a
// dependent fp32 MFMA costs 44 cycles against 32 issue) for a product that stands alone on the
// critical path.  X'X-type products stay bitwise symmetric (each half is).
__device__ __forceinline__ f32x4 xty2(f32x4 X, f32x4 Y, f32x4 C) {
    f32x4 D = zero4();
    C = __builtin_amdgcn_mfma_f32_16x16x4f32(X[0], Y[0], C, 0, 0, 0);
    D = __builtin_amdgcn_mfma_f32_16x16x4f32(X[1], Y[1], D, 0, 0, 0);
    C = __builtin_amdgcn_mfma_f32_16x16x4f32(X[2], Y[2], C, 0, 0, 0);
    D = __builtin_amdgcn_mfma_f32_16x16x4f32(X[3], Y[3], D, 0, 0, 0);
    return C + D;
}

// wave-uniform value of lane `src` (src must be a compile-time constant after unrolling)
__device__ __forceinline__ float bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// 1/d: v_rcp_f32 plus one Newton step (off the IEEE divide expansion, ~1 ulp)
__device__ __forceinline__ float fast_rcp(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    return fmaf(r, fmaf(-d, r, 1.0f), r);
}

// ---- global access as  uniform base + 32-bit lane offset + immediate  ----------------------------
// (the addressing mode of global_load/store with an SGPR base: no 64-bit VALU address arithmetic
// per access; `base` must be wave-uniform, `voff` and `imm` are byte offsets)
__device__ __forceinline__ float ld_f32(const float* base, unsigned voff, int imm) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)voff + imm);
}
__device__ __forceinline__ void st_f32(float* base, unsigned voff, int imm, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + (size_t)voff + imm) = v;
}

// ---- global <-> accumulator layout (tile image is column-major, 1 KiB, 16 B aligned) --------
__device__ __forceinline__ f32x4 load_tile(const float* __restrict__ t, int lane) {
    const int q = lane >> 4, c = lane & 15;
    return *reinterpret_cast<const f32x4*>(t + c * TS + 4 * q);
}
__device__ __forceinline__ void store_tile(float* __restrict__ t, int lane, f32x4 v) {
    const int q = lane >> 4, c = lane & 15;
    *reinterpret_cast<f32x4*>(t + c * TS + 4 * q) = v;
}
// transposed image of an accumulator-layout tile: 4 dword stores (64 B runs), so that a later
// load_tile() of the same memory yields the accumulator layout of the TRANSPOSE in one 16 B load
__device__ __forceinline__ void store_tile_t(float* __restrict__ t, int lane, f32x4 v) {
    const int q = lane >> 4, c = lane & 15;
    t[(4 * q + 0) * TS + c] = v[0];
    t[(4 * q + 1) * TS + c] = v[1];
    t[(4 * q + 2) * TS + c] = v[2];
    t[(4 * q + 3) * TS + c] = v[3];
}

// Ordering point between LDS accesses of different lanes of ONE wavefront.  A workgroup here is a
// single wave and the LDS serves a wave's DS instructions in issue order, so no s_barrier and no
// counter drain is needed -- only the compiler must not move DS accesses across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- LDS conversion tiles (column stride LDC) ----------------------------------------------
__device__ __forceinline__ void lds_store_acc(float* t, int lane, f32x4 v) {
    const int q = lane >> 4, c = lane & 15;
    *reinterpret_cast<f32x4*>(t + c * LDC + 4 * q) = v;
}
__device__ __forceinline__ f32x4 lds_load_acc(const float* t, int lane) {
    const int q = lane >> 4, c = lane & 15;
    return *reinterpret_cast<const f32x4*>(t + c * LDC + 4 * q);
}

// accumulator layout of the TRANSPOSE of a stored conversion tile: element (c, 4q+r)
__device__ __forceinline__ f32x4 lds_load_acc_t(const float* t, int lane) {
    const int q = lane >> 4, c = lane & 15;
    const float* p = t + (4 * q) * LDC + c;
    return f32x4{p[0], p[LDC], p[2 * LDC], p[3 * LDC]};
}

// ---- LDL' elimination in column layout ------------------------------------------------------
// Lane L < 16 holds column L of Huu (symmetric), lanes 16..31 the columns of [Hux | hu],
// lanes 32..47 the columns of I.  After the sweep every right-hand-side lane holds
//   col[j] = (D^-1/2 L^-1 rhs)[j]      i.e.  Y = D^-1/2 L^-1 [Hux|hu],  W = D^-1/2 L^-1,
// so that  Huu^-1 = W'W,  K = -W'Y,  P+ = Hxx - Y'Y  need no back substitution.
// All cross-lane traffic is v_readlane of lane j's registers (wave-uniform scalars).
// Returns false if a pivot is not positive (status "QP failure").
// `coupled`: bit j set if input j may couple with other inputs.  A pivot whose column is exactly
// diagonal (a masked-out input: zero column of B, no active constraint row) has all multipliers
// equal to zero, so its update loop is skipped -- bit-identical result.
//   MASK != DYNAMIC_MASK: the mask is a compile-time constant; inactive pivots AND inactive rows
//   vanish from the unrolled code and the elimination is straight-line code (ldl_pairs).
//   MASK == DYNAMIC_MASK: run-time mask, one wave-uniform branch per pivot.
constexpr unsigned DYNAMIC_MASK = 0xFFFFFFFFu;

// lowest set bit of mask at or above `from` (32 if none)
__host__ __device__ constexpr int next_bit(unsigned mask, int from) {
    for (int i = from; i < 32; ++i)
        if ((mask >> i) & 1u) return i;
    return 32;
}

// Static mask: the coupled pivots are eliminated TWO AT A TIME (2x2 block pivots).  The serial part
// of an LDL' is the chain  pivot -> 1/sqrt -> multipliers -> update -> next pivot ; a block pivot
// halves the number of links, and all its cross-lane reads (the 2x2 block and the two multiplier
// columns) are independent of the chain, so they issue ahead of it.  With rows a < b of the pair,
//   ra = 1/sqrt(d_a)   t = H_ba ra   rb = 1/sqrt(d_b - t^2)
//   y_a = row_a ra     y_b = (row_b - t y_a) rb                         (the rows as they leave)
//   row_i -= m_ia y_a + m_ib y_b ,  m_ia = H_ia ra ,  m_ib = (H_ib - m_ia t) rb
// which is the scalar recursion carried out for a and b at once.  A, the first pivot of the pair,
// walks over the set bits of MASK.
template <int NU, unsigned MASK, bool SLOT3, int A>
__device__ __forceinline__ void ldl_pairs(float (&col)[NU], bool& ok) {
    constexpr unsigned M = MASK & ((NU < 32) ? ((1u << NU) - 1u) : 0xFFFFFFFFu);
    if constexpr (A < NU) {
        constexpr int Bi = next_bit(M, A + 1);
        constexpr int la = SLOT3 ? slot_of(A) : A;
        if constexpr (Bi < NU) {
            constexpr int lb = SLOT3 ? slot_of(Bi) : Bi;
            const float da = bcast(col[A], la), e = bcast(col[Bi], la), db = bcast(col[Bi], lb);
            const float ra = __builtin_amdgcn_rsqf(da);
            const float t = e * ra;
            const float db2 = fmaf(-t, t, db);
            const float rb = __builtin_amdgcn_rsqf(db2);
            ok = ok && (da > 0.0f) && (db2 > 0.0f);
            const float ya = col[A] * ra;
            const float yb = fmaf(-t, ya, col[Bi]) * rb;
#pragma unroll
            for (int i = Bi + 1; i < NU; ++i) {
                if ((M >> i) & 1u) {
                    const float lia = bcast(col[i], la), lib = bcast(col[i], lb);
                    const float mia = lia * ra;
                    const float mib = fmaf(-mia, t, lib) * rb;
                    col[i] = fmaf(-mib, yb, fmaf(-mia, ya, col[i]));
                }
            }
            col[A] = ya;
            col[Bi] = yb;
            ldl_pairs<NU, MASK, SLOT3, next_bit(M, Bi + 1)>(col, ok);
        } else {   // odd one out: the last coupled pivot, nothing left to update
            const float d = bcast(col[A], la);
            ok = ok && (d > 0.0f);
            col[A] = col[A] * __builtin_amdgcn_rsqf(d);
        }
    }
}

// rs_free[j] = 1/sqrt(pivot) of input j when it is uncoupled: its pivot is then the constant
// R_jj + reg (nothing of B'PB or of the barrier reaches it), so a static mask needs neither the
// broadcast nor the v_rsq for it.
// SLOT3: column j of the system sits in lane slot_of(j) (slot layout) instead of lane j.
template <int NU, unsigned MASK, bool SLOT3>
__device__ __forceinline__ bool ldl_eliminate(float (&col)[NU], unsigned coupled, const float (&rs_free)[NU]) {
    bool ok = true;
    if constexpr (MASK != DYNAMIC_MASK) {
#pragma unroll
        for (int j = 0; j < NU; ++j)
            if (!((MASK >> j) & 1u)) col[j] = col[j] * rs_free[j];
        ldl_pairs<NU, MASK, SLOT3, next_bit(MASK, 0)>(col, ok);
    } else {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int lane_j = SLOT3 ? slot_of(j) : j;   // a constant after unrolling
            const float d = bcast(col[j], lane_j);
            ok = ok && (d > 0.0f);
            // row j leaves as col[j]/sqrt(d); the multiplier row col[j]/d is that times 1/sqrt(d)
            // again (one v_rsq per pivot and no reciprocal on the dependent chain)
            const float rs = __builtin_amdgcn_rsqf(d);
            col[j] = col[j] * rs;
            if ((coupled >> j) & 1u) {
                const float w = col[j] * rs;
#pragma unroll
                for (int i = j + 1; i < NU; ++i) {
                    const float l = bcast(col[i], lane_j);
                    col[i] = fmaf(-l, w, col[i]);
                }
            }
        }
    }
    return ok;
}

// wave reductions (64 lanes), result in every lane.  Within a row of 16 lanes four DPP exchanges (xor 1,
// xor 2 inside the quads, then the mirrored halves and the mirrored row) leave the row's result in all its
// lanes; the four rows meet through v_readlane.  ~11 instructions and ~100 cycles of dependent latency; the
// same butterfly on __shfl_xor is six ds_bpermute round trips through the LDS crossbar (~500 cycles), and
// the interior-point update runs up to six reductions per sweep on its critical path.
template <int CTRL>
__device__ __forceinline__ float dpp_exchange(float v) {
    const int x = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false));
}
template <class Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {
    v = op(v, dpp_exchange<0xB1>(v));    // quad_perm [1,0,3,2]
    v = op(v, dpp_exchange<0x4E>(v));    // quad_perm [2,3,0,1]
    v = op(v, dpp_exchange<0x141>(v));   // row_half_mirror
    v = op(v, dpp_exchange<0x140>(v));   // row_mirror
    const float r0 = bcast(v, 0), r1 = bcast(v, 16), r2 = bcast(v, 32), r3 = bcast(v, 48);
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_min(float v) { return wave_reduce(v, [](float a, float b) { return fminf(a, b); }); }
__device__ __forceinline__ float wave_max(float v) { return wave_reduce(v, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce(v, [](float a, float b) { return a + b; }); }

}  // namespace nmpc
