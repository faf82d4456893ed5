// nmpc_wb.hip -- batched NMPC solve of the whole-body model (nx = 42, nu = 30) on gfx950: a blocked Riccati /
// interior-point kernel family beside the 16x16 single-tile family of nmpc_solve.hip.
//
// Replaces, for the problem the reference actually solves (q18 + v18 + h6 | a18 + f12, solver.py:88-92,405-418),
// the per-step solve  QuadrupedAcadosSolver.solve -> acados SQP / HPIPM  (solver.py:396-403).  Two kernels per
// SQP iteration, as in nmpc_solve.hip:
//   nmpc_wb_linearize_kernel  one thread per (problem, node): foot kinematics with first and second
//        derivatives, dynamics defect and its non-trivial Jacobian blocks, the scaled dense residual Jacobian
//        Js = sqrt(W) [J | res] of the swing / contact / consistency rows as a compact record (cj_index), gradients of the
//        diagonal residuals, friction-pyramid values.
//   nmpc_wb_qp_kernel  one problem per wavefront, 48x48 homogeneous stage matrices as 3x3 tiles of 16x16 fp32 in
//        the accumulator layout of v_mfma_f32_16x16x4_f32 (nmpc_tile.hpp: X'Y on registers, no data movement):
//        prologue   Q~_k = Js'Js + diag  -- the Gauss-Newton J'WJ contraction -- on the matrix pipe, per node; the operand tiles
//                   gathered from the node's compact record through the LDS
//        phase R    backward sweep: P~A~ (A~ = I + N~), P~B~, H~ux, Huu, H~xx -- the integrator's identity blocks as lane / row
//                   shifts, the momentum rows (registers 0, 1 of tile row 2: pos_of) as two-step MFMA products; LDL' of the
//                   30x30 Huu in column layout (lane = column) applied to [Huu | I] -> W: 4x4 panels factorised on wave-uniform
//                   values, rank-1 updates of the row groups below on v_mfma_f32_4x4x1 with A broadcast (ldl_panel);
//                   Y = W H~ux, P~+ = H~xx - Y'Y, K~' = -Y'W on MFMA tiles
//        phase F    forward sweep: du = K~ dx~ row-per-lane (gain tiles -> LDS -> rows), dx+ from the model's sparse structure
//        phase I    interior point on the friction pyramids, lane = stage, barrier terms G'DG / G'v in closed form
//        phase S    step, status, write-back (warm-start shift folded in as an index map)
// Stage data that does not fit the LDS (Q~, K~' images, 6 + 6 KB per stage and sweep) streams through an HBM workspace;
// the per-stage record, the elimination columns, the transposition buffer and the gains of the last two stages live in the LDS
// (37.8 KB at N = 30: four waves per CU).  DESIGN.md 5b has the measurements behind every choice.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "../../include/nmpc.h"
#include "nmpc_wb_model.hpp"

// Contraction off from here on (as in the QP kernel of nmpc_solve.hip): every intended a*b + c of this file is an fmaf or an MFMA;
// what the backend would fuse on its own depends on the instantiation.
#pragma clang fp contract(off)

namespace nmpc {
namespace wb {

// Positions of the states in the 48-wide homogeneous vector x~ (three tiles of 16).  States 0..35 (q, v_0..v_17) sit at their own
// index: tiles 0, 1 and slots 0..3 of tile 2.  The six momentum states sit at slots 4, 5, 8, 9, 12, 13 of tile 2 and the
// homogeneous coordinate at slot 6: registers 0 and 1 of lane rows 1..3 in the accumulator layout.  A product that contracts over
// tile row 2 against an operand whose only non-zero rows are momentum rows (the dense part of N~ = A~ - I and of B~) then needs
// contraction steps 0 and 1 only: 34 of 196 MFMAs of a backward stage less than with the states in their own order.
constexpr int HX = 38;                 // position of the homogeneous coordinate of x~ = [dx; 1]
constexpr int XW = 46;                 // positions in use: 0..45
__host__ __device__ constexpr int pos_of(int s) { return s < 36 ? s : 32 + 4 * (1 + (s - 36) / 2) + ((s - 36) & 1); }      // s < 42
__host__ __device__ constexpr int state_at(int p) {      // -1: padding or the homogeneous coordinate
    return p < 36 ? p : (p < XW && (p & 3) < 2) ? 36 + 2 * ((p - 36) >> 2) + (p & 1) : -1;
}
static_assert(pos_of(36) == 36 && pos_of(37) == 37 && pos_of(38) == 40 && pos_of(39) == 41 && pos_of(40) == 44 && pos_of(41) == 45, "momentum slots");
static_assert(state_at(40) == 38 && state_at(45) == 41 && state_at(HX) == -1 && state_at(39) == -1 && state_at(46) == -1, "momentum slots");
constexpr int HXQ = (HX - 32) >> 2, HXR = (HX - 32) & 3;      // lane row and register of row HX in tile row 2
static_assert(HXR == 2, "the homogeneous row is register 2 of its lane row");

// ---- the scaled residual Jacobian Js = sqrt(W) [J | res] of a node, COMPACT -------------------------------------------------------
// Js is 30 x 46 with 381 structural non-zeros.  As a dense tile image (6 KB per node, each thread of the linearisation scattering
// its ~440 dword stores over its own image) it cost 0.6 of the linearisation's 0.78 ms per launch; it is now a 388-float record
// in the order the linearisation produces it, written in 16 B pieces, and the QP kernel's prologue gathers its operand tiles
// from the record through a per-lane index map (cj_index) -- the way the stage sweeps synthesise N~ and B~.
//   foot f, floats 88 f ..:  contact rows 3f+i (i < 3):  [6 c + i]      column q(xi_c)       c < 9  (xi = [r, theta, ql_f])
//                                                        [6 c + 3 + i]  column v(xi_c)
//                                                        [54 + i]       column HX (the scaled residual)
//                            swing row 12+f:             [57 + c], [66] column HX            ([67] padding)
//                            placement rows 22+2f+i:     [68 + 10 i + c], [68 + 10 i + 9] column HX
//   consistency, floats 352 ..:  rows 16+i: [3 i] h_lin_i, [3 i + 1] v_i, [3 i + 2] HX;  rows 19+i: [9 + 2 i] h_ang_i, [10 + 2 i] HX,
//                                [15 + 3 a + i] theta_a, [24 + 3 a + i] thetadot_a            ([33..35] padding)
constexpr int CJ_FOOT = 88, CJ_CONS = 4 * CJ_FOOT, CJ_FLOATS = CJ_CONS + 36;
constexpr int JS_FLOATS = CJ_FLOATS;
static_assert(CJ_FLOATS % 4 == 0, "records are whole 16 B pieces");
__host__ __device__ constexpr int xi_slot(int f, int qi) {      // inverse of xi_col: slot of coordinate qi in xi_f, or -1
    return qi < 6 ? qi : (qi >= 6 + 3 * f && qi < 9 + 3 * f) ? 6 + (qi - 6 - 3 * f) : -1;
}
// record index of element (row, column POSITION) of Js, or -1 for a structural zero
__host__ __device__ constexpr int cj_index(int row, int col) {
    if (row < 12) {
        const int f = row / 3, i = row % 3;
        if (col == HX) return CJ_FOOT * f + 54 + i;
        if (col < 18) { const int c = xi_slot(f, col); return c < 0 ? -1 : CJ_FOOT * f + 6 * c + i; }
        if (col < 36) { const int c = xi_slot(f, col - 18); return c < 0 ? -1 : CJ_FOOT * f + 6 * c + 3 + i; }
        return -1;
    }
    if (row < 16) {
        const int f = row - 12;
        if (col == HX) return CJ_FOOT * f + 66;
        if (col < 18) { const int c = xi_slot(f, col); return c < 0 ? -1 : CJ_FOOT * f + 57 + c; }
        return -1;
    }
    if (row < 19) {
        const int i = row - 16;
        return col == pos_of(WH + i) ? CJ_CONS + 3 * i : col == WV + i ? CJ_CONS + 3 * i + 1 : col == HX ? CJ_CONS + 3 * i + 2 : -1;
    }
    if (row < 22) {
        const int i = row - 19;
        if (col == pos_of(WH + 3 + i)) return CJ_CONS + 9 + 2 * i;
        if (col == HX) return CJ_CONS + 10 + 2 * i;
        if (col >= WQ + 3 && col < WQ + 6) return CJ_CONS + 15 + 3 * (col - WQ - 3) + i;
        if (col >= WV + 3 && col < WV + 6) return CJ_CONS + 24 + 3 * (col - WV - 3) + i;
        return -1;
    }
    if (row < 30) {
        const int f = (row - 22) / 2, i = (row - 22) % 2;
        if (col == HX) return CJ_FOOT * f + 68 + 10 * i + 9;
        if (col < 18) { const int c = xi_slot(f, col); return c < 0 ? -1 : CJ_FOOT * f + 68 + 10 * i + c; }
        return -1;
    }
    return -1;
}
static_assert(cj_index(0, 0) == 0 && cj_index(5, 18 + 9) == CJ_FOOT + 6 * 6 + 3 + 2 && cj_index(4, 6) == -1 && cj_index(13, HX) == CJ_FOOT + 66, "cj_index");
static_assert(cj_index(17, pos_of(WH + 1)) == CJ_CONS + 3 && cj_index(20, WV + 4) == CJ_CONS + 24 + 3 + 1 && cj_index(29, 2) == CJ_FOOT * 3 + 68 + 10 + 2, "cj_index");
constexpr int XT = 3, UT = 2;          // 16-wide tiles of the state (48) and input (32) dimensions
constexpr int JT = 2;                  // K tiles of the dense residual Jacobian (22 rows)
constexpr int IMG = TILE;              // floats of one tile image (column-major 16x16)
constexpr int QT_FLOATS = XT * XT * IMG, KT_FLOATS = UT * XT * IMG;

// per-node record written by the linearisation (float offsets)
constexpr int R_D = 0;                 // defect d, by POSITION: [48], zero where no state sits
constexpr int R_HQ = 48;               // d h_ang+ / d q[3..17]: [3][16]
constexpr int R_HF = 96;               // d h_ang+ / d f: [3][12]
constexpr int R_CDT = 132;             // dt c_i: d h_lin+ / d f_i = cdt_i I
constexpr int R_R = 136;               // input gradient r[30] (+2)
constexpr int R_C = 168;               // friction pyramid values c = G u - h [16]
constexpr int R_ACT = 184, R_COST = 185;
constexpr int R_ZERO = 186, R_DT = 187, R_DT2 = 224;   // constants the tile synthesis reads like any other entry
constexpr int R_GQ = 188;              // gradient of the diagonal residuals on x[0..35]
constexpr int REC = 228;

struct WbArgs {
    ModelParams mp;
    float W[NY], We[NYE];
    float reg, reg_e;
    int N, B;
    int max_sqp, n_ipm, yref_per_stage, it, shift;
    int precision;      // 0: fp32; 1: bf16 residual Jacobian, J'WJ on the bf16 matrix pipe; 2: split bf16 (hi + lo); 3: hi + mid + lo
    int pos_rows;       // 1: some foot-placement weight (W / W_e rows RY_POS.., RE_POS..) is non-zero
    float nlp_tol, mu0, sigma, s_min, gamma, tau_min;
    const float* x0;
    const float* yref;
    const float* yref_e;
    const float* params;
    float* X;
    float* U;
    int* status;
    float* stats;
    float* ws;
    const int* skip;    // nullptr, or dev [B] flag words: a problem with skip[b] & skip_mask != 0 is left untouched (nmpc_set_skip)
    int skip_mask;
};

__host__ __device__ inline int r4(int n) { return (n + 3) & ~3; }
__device__ __forceinline__ int shifted_node(int k, int shift, int N) { return (k >= 1 && k <= N - shift) ? k + shift : k; }
__device__ __forceinline__ bool shifted_stage_valid(int k, int shift, int N) { return k < N - shift; }

// arrays of the lane = stage phases, feature-major [feature][stage], odd stage stride
struct StageArr {
    int NS, dX, dU, dXp, dUp, sv, lv, total;
    __host__ __device__ explicit StageArr(int N) {
        NS = (N + 1) | 1;
        int o = 0;
        dX = o;  o += r4(NX * NS);
        dU = o;  o += r4(NU * NS);
        dXp = o; o += r4(NX * NS);
        dUp = o; o += r4(NU * NS);
        sv = o;  o += r4(NG * NS);
        lv = o;  o += r4(NG * NS);
        total = o;
    }
};
// workspace of one problem (float offsets)
struct WsLayout {
    size_t rec, js, qt, kt, arr, flag, stride;
    __host__ __device__ explicit WsLayout(int N) {
        size_t o = 0;
        rec = o; o += (size_t)(N + 1) * REC;
        js = o;  o += (size_t)(N + 1) * JS_FLOATS;
        qt = o;  o += (size_t)(N + 1) * QT_FLOATS;
        kt = o;  o += (size_t)N * KT_FLOATS;
        arr = o; o += StageArr(N).total;
        flag = o; o += 4;
        stride = (o + 63) & ~(size_t)63;
#ifdef WB_T_ODD_STRIDE       // timing build: an odd number of 256 B units per problem (do the problems' images camp on memory channels?)
        if (((stride / 64) & 1) == 0) stride += 64;
#endif
    }
};

// weight of the diagonal residual that sits on state index s < 36 (base / joint rows), stage and terminal
__device__ __forceinline__ float wdiag(const WbArgs& a, int s, bool term) {
    const int i = (s < 6) ? RY_BASE + s : (s < 18) ? RY_JOINT + (s - 6) : (s < 24) ? RY_BASE + 6 + (s - 18) : RY_JOINT + 12 + (s - 24);
    return term ? a.We[i] : a.W[i];        // base and joint rows have the same offsets in W and W_e
}
__device__ __forceinline__ int yref_of_state(int s) {
    return (s < 6) ? RY_BASE + s : (s < 18) ? RY_JOINT + (s - 6) : (s < 24) ? RY_BASE + 6 + (s - 18) : RY_JOINT + 12 + (s - 24);
}

// ------------------------------------------------------------------------------------------------------------
// Linearisation: thread t <-> (problem b, node k), k = N is the terminal node.
#ifndef WB_LIN_WAVES
#define WB_LIN_WAVES 1
#endif
__global__ __launch_bounds__(64, WB_LIN_WAVES) void nmpc_wb_linearize_kernel(const WbArgs a) {
    const int N = a.N;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)a.B * (N + 1)) return;
    const int b = (int)(t / (N + 1)), k = (int)(t - (long long)b * (N + 1));
    const WsLayout wl(N);
    float* ws = a.ws + (size_t)b * wl.stride;
    if (a.it > 0 && reinterpret_cast<const int*>(ws + wl.flag)[0]) return;
    if (a.skip && (a.skip[b] & a.skip_mask) != 0) return;
    const bool term = (k == N);
    const ModelParams& mp = a.mp;
    const float dt = mp.dt;
    const float* Xg = a.X + (size_t)b * (N + 1) * NX;
    const float* Ug = a.U + (size_t)b * N * NU;
    // alignments the vectoriser may rely on (a lane is a node: every access of this kernel is strided across the wave, so its
    // cost is the number of memory instructions -- 16 B pieces of the record instead of 210 dword stores, 8 B pieces of x, u, yref).
    // Workspace: 256 B; caller's arrays: 8 B (x, u, references: rows of 42, 30, 90 / 66 floats) and 16 B (parameters), checked by
    // the C-ABI (nmpc_api.hip, launch_wb).
    float* rec = static_cast<float*>(__builtin_assume_aligned(ws + wl.rec + (size_t)k * REC, 16));
    float* js = ws + wl.js + (size_t)k * JS_FLOATS;

    float x[NX], u[NU], p[NP];
    const float* xk = static_cast<const float*>(__builtin_assume_aligned(Xg + (size_t)shifted_node(k, a.shift, N) * NX, 8));
#pragma unroll
    for (int i = 0; i < NX; i += 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(xk + i); x[i] = v[0]; x[i + 1] = v[1]; }
    const int ks = term ? 0 : k;
    {
        // warm-start shift as an index map; in the exposed tail the contact forces are zero and the accelerations keep the
        // previous solution's values at that stage (solver.py:316-322 moves a[:, :n_warm_start] and zeroes f[:, n_warm_start:])
        const bool ok = (a.shift == 0) || shifted_stage_valid(ks, a.shift, N);
        const float* uk = static_cast<const float*>(__builtin_assume_aligned(Ug + (size_t)(ok ? ks + a.shift : ks) * NU, 8));
#pragma unroll
        for (int i = 0; i < NU; i += 2) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(uk + i);
            u[i] = (ok || i < WF) ? v[0] : 0.0f; u[i + 1] = (ok || i + 1 < WF) ? v[1] : 0.0f;
        }
    }
    const float* pg = static_cast<const float*>(__builtin_assume_aligned(a.params + ((size_t)b * (N + 1) + k) * NP, 16));
    static_assert(NX % 2 == 0 && NU % 2 == 0 && NY % 2 == 0 && NYE % 2 == 0 && NP % 4 == 0 && REC % 4 == 0, "row alignments");
#pragma unroll
    for (int i = 0; i < NP; i += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(pg + i);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[i + r] = v[r];
    }
    const int ny = term ? NYE : NY;
    const float* yr = static_cast<const float*>(__builtin_assume_aligned(
        term ? a.yref_e + (size_t)b * NYE
             : a.yref + (size_t)b * (a.yref_per_stage ? (size_t)N * NY : (size_t)NY) + (a.yref_per_stage ? (size_t)k * NY : 0), 8));
    const float* Wv = term ? a.We : a.W;
    const int r_sw = term ? RE_SWING : RY_SWING, r_ct = term ? RE_CNT : RY_CNT, r_cs = term ? RE_CONS : RY_CONS;
    const int r_ps = term ? RE_POS : RY_POS;
    (void)ny;

    // ---- kinematics
    BaseRot br;
    {
        const float th[3] = {x[WQ + 3], x[WQ + 4], x[WQ + 5]}, thd[3] = {x[WV + 3], x[WV + 4], x[WV + 5]};
        base_rotation<true>(th, thd, br);
    }
    float cost = 0.0f;
    // a run of the node's compact Jacobian record (layout: cj_index), from registers, in 16 B pieces
    float* cj = static_cast<float*>(__builtin_assume_aligned(js, 16));
    auto put_cj = [&](int off, int n, const float* v) {
#pragma unroll
        for (int i = 0; i < n; i += 4) *reinterpret_cast<f32x4*>(cj + off + i) = f32x4{v[i], v[i + 1], v[i + 2], v[i + 3]};
    };

    // a run of the record, from registers, in 16 B pieces (`off` a multiple of four)
    auto put_rec = [&](int off, const auto& v) {
        constexpr int n = (int)(sizeof(v) / sizeof(float));
        static_assert(n % 4 == 0, "record runs are whole 16 B pieces");
#pragma unroll
        for (int i = 0; i < n; i += 4) *reinterpret_cast<f32x4*>(rec + off + i) = f32x4{v[i], v[i + 1], v[i + 2], v[i + 3]};
    };
    float tau_acc[3] = {0.f, 0.f, 0.f}, F[3] = {0.f, 0.f, 0.f};
    float hfr[36], cdt[4] = {0.f, 0.f, 0.f, 0.f};      // d h_ang+ / d f [3][12] and dt c_f of the record
#pragma unroll
    for (int i = 0; i < 36; ++i) hfr[i] = 0.0f;
    float hq[3][15];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 15; ++j) hq[i][j] = 0.0f;

#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const float ql[3] = {x[WQ + 6 + 3 * f], x[WQ + 7 + 3 * f], x[WQ + 8 + 3 * f]};
        const float wl3[3] = {x[WV + 6 + 3 * f], x[WV + 7 + 3 * f], x[WV + 8 + 3 * f]};
        Leg lg;
        leg_kin<true>(mp, f, ql, wl3, lg);
        // world position, Jacobian J (3x9 wrt xi = [r, theta, ql]) and its time derivative Jd
        float Rb[3];
        mv(br.R, lg.b, Rb);
        const float pz = x[WQ + 2] + Rb[2];
        float J[3][9], Jd[3][9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) { J[i][c] = (i == c) ? 1.0f : 0.0f; Jd[i][c] = 0.0f; }
#pragma unroll
        for (int aa = 0; aa < 3; ++aa) {
            float t0[3], t1[3], t2[3];
            mv(br.Ra[aa], lg.b, t0);
            mv(br.Rad[aa], lg.b, t1);
            mv(br.Ra[aa], lg.bd, t2);
#pragma unroll
            for (int i = 0; i < 3; ++i) { J[i][3 + aa] = t0[i]; Jd[i][3 + aa] = t1[i] + t2[i]; }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float jc[3] = {lg.Jb.m[c], lg.Jb.m[3 + c], lg.Jb.m[6 + c]};
            const float jdc[3] = {lg.Jbd.m[c], lg.Jbd.m[3 + c], lg.Jbd.m[6 + c]};
            float t0[3], t1[3], t2[3];
            mv(br.R, jc, t0);
            mv(br.Rd, jc, t1);
            mv(br.R, jdc, t2);
#pragma unroll
            for (int i = 0; i < 3; ++i) { J[i][6 + c] = t0[i]; Jd[i][6 + c] = t1[i] + t2[i]; }
        }
        const float cf = p[f], peak = p[4 + f], ppz = p[8 + 3 * f + 2];
        // foot-placement rows (pos_cost, solver.py:128-137,272-273): world x, y of the foot - planned location.  Weight 0
        // outside the contact-restricted mode: the rows are then exact zeros in the image and are not rewritten
        // (a.pos_rows, wave-uniform; nmpc_set_weights has the image cleared when the rows go from weighted to unweighted)
        float fb[CJ_FOOT];      // this foot's part of the compact Jacobian record
        fb[67] = 0.0f;
        if (a.pos_rows) {
            const float px[2] = {x[WQ] + Rb[0], x[WQ + 1] + Rb[1]};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float w = Wv[r_ps + 2 * f + i], sw = sqrtf(w);
                const float res = px[i] - yr[r_ps + 2 * f + i];
                cost += 0.5f * w * res * res;
#pragma unroll
                for (int c = 0; c < 9; ++c) fb[68 + 10 * i + c] = sw * J[i][c];
                fb[68 + 10 * i + 9] = sw * res;
            }
            put_cj(CJ_FOOT * f + 68, 20, fb + 68);
        }
        // swing row: peak z_foot - ref
        {
            const float w = Wv[r_sw + f], sw = sqrtf(w);
            const float res = peak * pz - yr[r_sw + f];
            cost += 0.5f * w * res * res;
#pragma unroll
            for (int c = 0; c < 9; ++c) fb[57 + c] = sw * peak * J[2][c];
            fb[66] = sw * res;
        }
        // contact rows: c (J v + p_gain e_z (z - plane_z)) - ref
        {
            float sres[3], swc[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float vel = 0.0f;
#pragma unroll
                for (int c = 0; c < 9; ++c) vel += J[i][c] * x[WV + xi_col(f, c)];
                const float w = Wv[r_ct + 3 * f + i], sw = sqrtf(w);
                const float res = cf * (vel + (i == 2 ? mp.p_gain * (pz - ppz) : 0.0f)) - yr[r_ct + 3 * f + i];
                cost += 0.5f * w * res * res;
                sres[i] = sw * res; swc[i] = sw * cf;
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                fb[6 * c + 0] = swc[0] * Jd[0][c]; fb[6 * c + 1] = swc[1] * Jd[1][c]; fb[6 * c + 2] = swc[2] * (Jd[2][c] + mp.p_gain * J[2][c]);
                fb[6 * c + 3] = swc[0] * J[0][c];  fb[6 * c + 4] = swc[1] * J[1][c];  fb[6 * c + 5] = swc[2] * J[2][c];
            }
            fb[54] = sres[0]; fb[55] = sres[1]; fb[56] = sres[2];
        }
        put_cj(CJ_FOOT * f, 68, fb);
        if (!term) {   // momentum rows of the dynamics
            const float ff[3] = {u[WF + 3 * f], u[WF + 3 * f + 1], u[WF + 3 * f + 2]};
            float tq[3];
            cross(Rb, ff, tq);
#pragma unroll
            for (int i = 0; i < 3; ++i) { F[i] += cf * ff[i]; tau_acc[i] += cf * tq[i]; }
            // d(arm x f)/d xi_c, c = 3..8 (arm = R b does not depend on r)
#pragma unroll
            for (int c = 3; c < 9; ++c) {
                const float da[3] = {J[0][c], J[1][c], J[2][c]};
                float tc[3];
                cross(da, ff, tc);
#pragma unroll
                for (int i = 0; i < 3; ++i) hq[i][xi_col(f, c) - 3] += dt * cf * tc[i];
            }
            const float ax[9] = {0.f, -Rb[2], Rb[1], Rb[2], 0.f, -Rb[0], -Rb[1], Rb[0], 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) hfr[i * 12 + 3 * f + j] = dt * cf * ax[3 * i + j];
            cdt[f] = dt * cf;
        }
    }
    // ---- consistency rows  h - A_g(q) v,  A_g v = [m rdot ; R I_b E(theta) thetadot]
    {
        float cb[36];      // the consistency part of the compact Jacobian record
        cb[33] = cb[34] = cb[35] = 0.0f;
        const float thd[3] = {x[WV + 3], x[WV + 4], x[WV + 5]};
        const float Ib[3] = {mp.ixx, mp.iyy, mp.izz};
        float sy, cy, sx, cx;
        sincosf(x[WQ + 4], &sy, &cy);
        sincosf(x[WQ + 5], &sx, &cx);
        const M3 E = {{-sy, 0.f, 1.f, cy * sx, cx, 0.f, cx * cy, -sx, 0.f}};
        const M3 Ea[3] = {{{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}},
                          {{-cy, 0.f, 0.f, -sy * sx, 0.f, 0.f, -cx * sy, 0.f, 0.f}},
                          {{0.f, 0.f, 0.f, cy * cx, -sx, 0.f, -sx * cy, -cx, 0.f}}};
        float wbv[3], Iw[3], L[3];
        mv(E, thd, wbv);
#pragma unroll
        for (int i = 0; i < 3; ++i) Iw[i] = Ib[i] * wbv[i];
        mv(br.R, Iw, L);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            {   // linear momentum
                const float w = Wv[r_cs + i], sw = sqrtf(w);
                const float res = x[WH + i] - mp.mass * x[WV + i] - yr[r_cs + i];
                cost += 0.5f * w * res * res;
                cb[3 * i] = sw; cb[3 * i + 1] = -sw * mp.mass; cb[3 * i + 2] = sw * res;
            }
            {   // angular momentum
                const float w = Wv[r_cs + 3 + i], sw = sqrtf(w);
                const float res = x[WH + 3 + i] - L[i] - yr[r_cs + 3 + i];
                cost += 0.5f * w * res * res;
                cb[9 + 2 * i] = sw; cb[10 + 2 * i] = sw * res;
            }
        }
#pragma unroll
        for (int aa = 0; aa < 3; ++aa) {
            float t0[3], t1[3], t2[3], ew[3], iew[3], iec[3];
            mv(br.Ra[aa], Iw, t0);
            mv(Ea[aa], thd, ew);
#pragma unroll
            for (int i = 0; i < 3; ++i) { iew[i] = Ib[i] * ew[i]; iec[i] = Ib[i] * E.m[3 * i + aa]; }
            mv(br.R, iew, t1);
            mv(br.R, iec, t2);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float sw = sqrtf(Wv[r_cs + 3 + i]);
                cb[15 + 3 * aa + i] = -sw * (t0[i] + t1[i]);
                cb[24 + 3 * aa + i] = -sw * t2[i];
            }
        }
        put_cj(CJ_CONS, 36, cb);
    }
    // ---- diagonal residuals (base, joint) on x[0..35]: gradient and cost
    {
        float gq[36];
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            const float w = wdiag(a, s, term);
            const float e = x[s] - yr[yref_of_state(s)];
            gq[s] = w * e;
            cost += 0.5f * w * e * e;
        }
        put_rec(R_GQ, gq);
    }
    if (!term) {
        // input residuals: acc on a[6..17], f_reg on f
        float rr[32];
        rr[30] = rr[31] = 0.0f;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            float g = 0.0f;
            if (i >= 6 && i < 18) {
                const float w = a.W[RY_ACC + i - 6], e = u[i] - yr[RY_ACC + i - 6];
                g = w * e; cost += 0.5f * w * e * e;
            } else if (i >= WF) {
                const float w = a.W[RY_FREG + i - WF], e = u[i] - yr[RY_FREG + i - WF];
                g = w * e; cost += 0.5f * w * e * e;
            }
            rr[i] = g;
        }
        put_rec(R_R, rr);
        // dynamics defect
        const float* xn_g = static_cast<const float*>(__builtin_assume_aligned(Xg + (size_t)shifted_node(k + 1, a.shift, N) * NX, 8));
        float xn[NX], dd[48];
#pragma unroll
        for (int i = 36; i < 48; ++i) dd[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < NX; i += 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(xn_g + i); xn[i] = v[0]; xn[i + 1] = v[1]; }
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const float vn = x[WV + i] + dt * u[WA + i];
            dd[WV + i] = vn - xn[WV + i];
            dd[WQ + i] = x[WQ + i] + dt * vn - xn[WQ + i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            dd[pos_of(WH + i)] = x[WH + i] + dt * (F[i] + (i == 2 ? mp.mass * mp.gz : 0.0f)) - xn[WH + i];
            dd[pos_of(WH + 3 + i)] = x[WH + 3 + i] + dt * tau_acc[i] - xn[WH + 3 + i];
        }
        put_rec(R_D, dd);
        {
            float hq16[48];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) hq16[i * 16 + j] = j < 15 ? hq[i][j < 15 ? j : 0] : 0.0f;
            put_rec(R_HQ, hq16);
        }
        put_rec(R_HF, hfr);
        put_rec(R_CDT, cdt);
        // friction pyramid
        float fv[12], g[NG];
#pragma unroll
        for (int i = 0; i < 12; ++i) fv[i] = u[WF + i];
        gdot(mp, fv, g);
        put_rec(R_C, g);       // h = 0
        static_assert(R_ACT % 4 == 0 && R_COST == R_ACT + 1 && R_ZERO == R_ACT + 2 && R_DT == R_ACT + 3 && R_DT2 % 4 == 0, "tail of the record");
        const unsigned act = (a.n_ipm > 0) ? active_mask(p) : 0u;
        *reinterpret_cast<f32x4*>(rec + R_ACT) = f32x4{__uint_as_float(act), cost, 0.0f, dt};
        rec[R_DT2] = dt * dt;
    } else {
        rec[R_COST] = cost;
    }
}

// ------------------------------------------------------------------------------------------------------------
// tile helpers of the blocked sweeps
constexpr int LDU = 36;      // LDS column stride of the 32-row elimination columns (Huu | H~ux -> W | Y)
constexpr int LDH = 52;      // LDS column stride of the 48-row transposition buffer of H~xx
constexpr int IPMW = 57;     // LDS row of a stage's barrier-modified input terms: rt[30] at 0, Rf[4][5] at 32
constexpr int IPM_RF = 32;

constexpr int KLD = 20;                      // LDS column stride of a gain tile K~' (conflict-free 16 B row reads)
constexpr int KBUF = UT * XT * 16 * KLD;     // the six tiles of one stage
struct WbLds {
    int colU, hbuf, recb, ipm, klds, n_klds, total;
    __host__ __device__ explicit WbLds(int N) {
        int o = 0;
        colU = o; o += NG * 0 + 32 * LDU;
        hbuf = o; o += 48 * LDH;
        recb = o; o += 256;
        ipm = o;  o += r4(N * IPMW);
        // the gains of the LAST stages of a backward sweep (the first of the forward sweep) stay in the LDS -- as many as fit next to
        // three other waves of the CU (160 KB / 4): the forward sweep starts without a trip to memory and those images never leave the CU
        klds = o;
        n_klds = (o + 2 * KBUF) * 4 <= 40 * 1024 ? 2 : (o + KBUF) * 4 <= 40 * 1024 ? 1 : 0;
        if (n_klds > N) n_klds = N;
        o += n_klds * KBUF;
        total = o;
    }
};

// Eight multipliers of pivot J at once: eight v_readlane into eight different SGPRs, then the wait states a VALU read of
// a freshly written SGPR needs on gfx950 -- once per group.  Left to the compiler the elimination came out as
// readlane -> s_nop 1 -> fma on ONE reused SGPR, 435 times per stage (tools/wb_stamps.py: 7.8 k cycles per stage).
template <int J>
__device__ __forceinline__ void bcast_group(const float (&v)[8], float (&o)[8]) {
    int r0, r1, r2, r3, r4, r5, r6, r7;
    // The leading s_nop is the wait state gfx950 needs between a VALU write of a VGPR and a v_readlane of it: the
    // hazard recognizer does not look into inline assembly, and the scheduler is free to sink the last FMA of the
    // previous group to just in front of this one (found as a run-to-run varying 5e-5 error after an unrelated edit
    // had changed the schedule).
    asm volatile("s_nop 0\n\tv_readlane_b32 %0, %8, %16\n\tv_readlane_b32 %1, %9, %16\n\tv_readlane_b32 %2, %10, %16\n\t"
                 "v_readlane_b32 %3, %11, %16\n\tv_readlane_b32 %4, %12, %16\n\tv_readlane_b32 %5, %13, %16\n\t"
                 "v_readlane_b32 %6, %14, %16\n\tv_readlane_b32 %7, %15, %16\n\ts_nop 1"
                 : "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3), "=s"(r4), "=s"(r5), "=s"(r6), "=s"(r7)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "n"(J));
    o[0] = __int_as_float(r0); o[1] = __int_as_float(r1); o[2] = __int_as_float(r2); o[3] = __int_as_float(r3);
    o[4] = __int_as_float(r4); o[5] = __int_as_float(r5); o[6] = __int_as_float(r6); o[7] = __int_as_float(r7);
}
// Coupling mask of the inputs at a stage: the accelerations always couple (bits 0..17); the three force components of a
// foot couple only while the foot stands -- a swing foot has a zero column in B~, no active pyramid row and a diagonal
// cost, so its row and column of Huu are exactly diagonal: its multipliers are exact zeros and nothing below it changes.
// The elimination is instantiated for the contact patterns of a trot (the two diagonal pairs, four-foot stance, flight)
// with those rows and pivots left out at compile time (160 of the 435 multipliers of a two-foot stage), and for the full
// mask, which is valid for every pattern.
__host__ __device__ constexpr unsigned coupling_mask(unsigned stance) {
    unsigned m = 0x3FFFFu;
    for (int f = 0; f < 4; ++f) m |= ((stance >> f) & 1u) ? (0x7u << (WF + 3 * f)) : 0u;
    return m;
}
// idx-th coupled row above `after`, or -1
__host__ __device__ constexpr int coupled_row(unsigned mask, int after, int idx) {
    for (int i = after + 1; i < NU; ++i)
        if ((mask >> i) & 1u) { if (idx == 0) return i; --idx; }
    return -1;
}
__host__ __device__ constexpr int coupled_rows_above(unsigned mask, int after) {
    int n = 0;
    for (int i = after + 1; i < NU; ++i) n += (mask >> i) & 1u;
    return n;
}
// the coupled rows below pivot J, eight per group (the last group padded with its own last row)
template <int J, unsigned MASK>
__device__ __forceinline__ void ldl_update(float (&X)[NU], float wx) {
    constexpr int n = coupled_rows_above(MASK, J);
#pragma unroll
    for (int g = 0; g < (n + 7) / 8; ++g) {
        float v[8], l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            constexpr int dummy = 0; (void)dummy;
            const int idx = 8 * g + u < n ? 8 * g + u : n - 1;
            v[u] = X[coupled_row(MASK, J, idx)];
        }
        bcast_group<J>(v, l);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (8 * g + u < n) X[coupled_row(MASK, J, 8 * g + u)] = fmaf(-l[u], wx, X[coupled_row(MASK, J, 8 * g + u)]);
    }
}
template <int J, unsigned MASK>
__device__ __forceinline__ void ldl_pivots(float (&X)[NU], bool& ok, const float (&rs_free)[12]) {
    if constexpr (J < NU) {
        if constexpr ((MASK >> J) & 1u) {
            const float d = bcast(X[J], J);
            ok = ok && (d > 0.0f);
            const float rs = __builtin_amdgcn_rsqf(d);
            X[J] *= rs;
            if constexpr (coupled_rows_above(MASK, J) > 0) ldl_update<J, MASK>(X, X[J] * rs);
        } else {
            // a decoupled input (a force component of a swing foot): nothing of B~'P~B~ or of the barrier reaches its pivot, which
            // is the constant W_f_reg + reg -- its 1 / sqrt comes from the kernel's prologue (the same v_rsq of the same bits)
            // instead of a v_readlane -> v_rsq on the elimination's dependent chain
            X[J] *= rs_free[J >= WF ? J - WF : 0];
        }
        ldl_pivots<J + 1, MASK>(X, ok, rs_free);
    }
}

// ---- the same elimination with its rank-1 updates on the matrix pipe ------------------------------------------------------
// Register file: row i of [Huu | I] in register i & 3 of Xq[i >> 2] (lane = column).  v_mfma_f32_4x4x1_16b_f32 is sixteen 4 x 4
// outer products, D[v][lane] = C[v][lane] + A[lane 4 (lane / 4) + v] B[lane], and with its A-broadcast control (cbsz = 4,
// abid = g) all sixteen blocks take the A operand of block g: D[v][lane] = C[v][lane] + A[lane 4g + v] B[lane].  Huu is symmetric
// and stays so under the elimination, so the multiplier of row i under pivot J, M[i][J], is what the scaled pivot row y_J holds
// in LANE i: with A = y_J and B = -y_J ONE two-pass instruction updates the four rows of group g in all 64 columns -- no
// broadcast, no data movement -- where the scalar form spends four v_readlane and four v_fma.
// The pivots themselves are a dependent chain (v_readlane -> v_rsq -> scale -> update of the next row -> v_readlane ...): a pivot
// updates the rows up to the end of panel P + LDL_AHEAD with scalar multipliers (v_readlane of y_J), the row groups further down
// take the four pivots of a panel as four MFMAs each (tools/probes/mfma4x4x1.hip checks the operand layout on the device).
constexpr int LDL_ROWS = 32, LDL_GROUPS = LDL_ROWS / 4;
#ifndef WB_LDL_AHEAD
#define WB_LDL_AHEAD 0
#endif
constexpr int LDL_AHEAD = WB_LDL_AHEAD;
template <int J>
__device__ __forceinline__ void bcast_lanes7(float y, float (&o)[7]) {
    int r0, r1, r2, r3, r4, r5, r6;
    // leading s_nop: see bcast_group
    asm volatile("s_nop 0\n\tv_readlane_b32 %0, %7, %8\n\tv_readlane_b32 %1, %7, %9\n\tv_readlane_b32 %2, %7, %10\n\t"
                 "v_readlane_b32 %3, %7, %11\n\tv_readlane_b32 %4, %7, %12\n\tv_readlane_b32 %5, %7, %13\n\t"
                 "v_readlane_b32 %6, %7, %14\n\ts_nop 1"
                 : "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3), "=s"(r4), "=s"(r5), "=s"(r6)
                 : "v"(y), "n"(J + 1), "n"(J + 2), "n"(J + 3), "n"(J + 4), "n"(J + 5), "n"(J + 6), "n"(J + 7));
    o[0] = __int_as_float(r0); o[1] = __int_as_float(r1); o[2] = __int_as_float(r2); o[3] = __int_as_float(r3);
    o[4] = __int_as_float(r4); o[5] = __int_as_float(r5); o[6] = __int_as_float(r6);
}
template <int J>
__device__ __forceinline__ void bcast_lanes3(float y, float (&o)[7]) {
    int r0, r1, r2;
    asm volatile("s_nop 0\n\tv_readlane_b32 %0, %3, %4\n\tv_readlane_b32 %1, %3, %5\n\tv_readlane_b32 %2, %3, %6\n\ts_nop 1"
                 : "=s"(r0), "=s"(r1), "=s"(r2)
                 : "v"(y), "n"(J + 1), "n"(J + 2), "n"(J + 3));
    o[0] = __int_as_float(r0); o[1] = __int_as_float(r1); o[2] = __int_as_float(r2);
    o[3] = o[4] = o[5] = o[6] = 0.0f;
}
// the pivots `panel` (bit jj: pivot 4P + jj is coupled) applied to the row groups G .. LDL_GROUPS-1
template <int P, int G, unsigned MASK>
__device__ __forceinline__ void ldl_trailing(f32x4 (&Xq)[LDL_GROUPS], const float (&ny)[4]) {
    if constexpr (G < LDL_GROUPS) {
        if constexpr (((MASK >> (4 * G)) & 0xFu) != 0u) {      // not a group of decoupled inputs only (or past the last row)
            constexpr unsigned panel = (MASK >> (4 * P)) & 0xFu;
            if constexpr (panel & 1u) Xq[G] = __builtin_amdgcn_mfma_f32_4x4x1f32(Xq[P][0], ny[0], Xq[G], 4, G, 0);
            if constexpr (panel & 2u) Xq[G] = __builtin_amdgcn_mfma_f32_4x4x1f32(Xq[P][1], ny[1], Xq[G], 4, G, 0);
            if constexpr (panel & 4u) Xq[G] = __builtin_amdgcn_mfma_f32_4x4x1f32(Xq[P][2], ny[2], Xq[G], 4, G, 0);
            if constexpr (panel & 8u) Xq[G] = __builtin_amdgcn_mfma_f32_4x4x1f32(Xq[P][3], ny[3], Xq[G], 4, G, 0);
        }
        ldl_trailing<P, G + 1, MASK>(Xq, ny);
    }
}
#ifdef WB_T_LDL_ROWCHAIN   // timing build (tools/ab_wb.sh): the panel's pivots as a chain of row operations
template <int P, unsigned MASK>
__device__ __forceinline__ void ldl_panel(f32x4 (&Xq)[LDL_GROUPS], bool& ok, const float (&rs_free)[12]) {
    if constexpr (P < LDL_GROUPS) {
        float ny[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        auto pivot = [&](auto jc) {
            constexpr int J = 4 * P + decltype(jc)::value;
            if constexpr (J < NU) {
                if constexpr ((MASK >> J) & 1u) {
                    const float d = bcast(Xq[P][J & 3], J);
                    ok = ok && (d > 0.0f);
                    const float y = Xq[P][J & 3] * __builtin_amdgcn_rsqf(d);
                    Xq[P][J & 3] = y;
                    ny[J & 3] = -y;
                    constexpr int end = 4 * (P + LDL_AHEAD) + 3;
                    constexpr int last = end < NU - 1 ? end : NU - 1;                    // last row that takes scalar multipliers
                    constexpr unsigned below = (last > J) ? (MASK >> (J + 1)) & ((1u << (last - J)) - 1u) : 0u;
                    if constexpr (below != 0u) {
                        float l[7];
                        if constexpr (last - J > 3) bcast_lanes7<J>(y, l); else bcast_lanes3<J>(y, l);
#pragma unroll
                        for (int u = 0; u < 7; ++u)
                            if (u < last - J && ((below >> u) & 1u)) {
                                const int i = J + 1 + u;
                                Xq[i >> 2][i & 3] = fmaf(-l[u], y, Xq[i >> 2][i & 3]);
                            }
                    }
                } else {
                    Xq[P][J & 3] *= rs_free[J >= WF ? J - WF : 0];
                }
            }
        };
        pivot(std::integral_constant<int, 0>{});
        pivot(std::integral_constant<int, 1>{});
        pivot(std::integral_constant<int, 2>{});
        pivot(std::integral_constant<int, 3>{});
        ldl_trailing<P, P + 1 + LDL_AHEAD, MASK>(Xq, ny);
        ldl_panel<P + 1, MASK>(Xq, ok, rs_free);
    }
}
#else
// One panel: the 4 x 4 diagonal block of the panel's rows (ten numbers, lanes 4P .. 4P+3 of the four registers) is factorised on
// wave-uniform values -- pivot -> v_rsq -> multiplier -> next pivot, three operations per pivot and no cross-lane move on the
// chain -- and the four rows follow it as y_c = (x_c - sum_{k<c} l_ck y_k) / sqrt(d_c); then the row groups below, on the matrix pipe.
template <int P, unsigned MASK>
__device__ __forceinline__ void ldl_panel(f32x4 (&Xq)[LDL_GROUPS], bool& ok, const float (&rs_free)[12]) {
    if constexpr (P < LDL_GROUPS) {
        constexpr int J0 = 4 * P;
        constexpr unsigned pm = (MASK >> J0) & 0xFu;          // coupled rows of the panel (rows past NU - 1 have no bit)
        float blk[4][4], l[4][4], rs[4] = {1.0f, 1.0f, 1.0f, 1.0f}, ny[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c)
                blk[r][c] = (((pm >> r) & 1u) && ((pm >> c) & 1u)) ? bcast(Xq[P][r], J0 + c) : 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if ((pm >> c) & 1u) {
                float d = blk[c][c];
#pragma unroll
                for (int k = 0; k < c; ++k)
                    if ((pm >> k) & 1u) d = fmaf(-l[c][k], l[c][k], d);
                ok = ok && (d > 0.0f);
                rs[c] = __builtin_amdgcn_rsqf(d);
#pragma unroll
                for (int r = c + 1; r < 4; ++r)
                    if ((pm >> r) & 1u) {
                        float t = blk[r][c];
#pragma unroll
                        for (int k = 0; k < c; ++k)
                            if ((pm >> k) & 1u) t = fmaf(-l[r][k], l[c][k], t);
                        l[r][c] = t * rs[c];
                    }
            } else if (J0 + c < NU) {
                // a decoupled input (a force component of a swing foot): see ldl_pivots
                rs[c] = rs_free[J0 + c >= WF ? J0 + c - WF : 0];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (J0 + c < NU) {
                float v = Xq[P][c];
                if ((pm >> c) & 1u) {
#pragma unroll
                    for (int k = 0; k < c; ++k)
                        if ((pm >> k) & 1u) v = fmaf(-l[c][k], Xq[P][k], v);
                }
                v *= rs[c];
                Xq[P][c] = v;
                ny[c] = -v;
            }
        ldl_trailing<P, P + 1, MASK>(Xq, ny);
        ldl_panel<P + 1, MASK>(Xq, ok, rs_free);
    }
}
#endif

// a lane shift inside the 16-lane rows (DPP row_shr:n = 0x110 + n: lane i reads lane i - n; row_shl:n = 0x100 + n: lane i + n);
// lanes whose source falls outside their row read 0
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// the same, for the lanes of bank 0 (lanes 0..3 of every row) only; the others read 0
template <int CTRL>
__device__ __forceinline__ float dpp_row_bank0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0x1, true));
}
// sum over the 16 lanes of a lane row, in every lane of it (the first four steps of wave_reduce)
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_row<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_row<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_row<0x141>(v);     // row_half_mirror
    v += dpp_row<0x140>(v);     // row_mirror
    return v;
}
// lane row q (16 lanes) of the result takes lane row q + 1 of `a`, the last lane row takes lane row 0 of `b`: the rows 4(q+1)..
// of a tile column, continued into the next tile, moved up by one quad (ds_bpermute: the LDS crossbar, no LDS memory)
// (one crossbar move: the lane row that nobody reads `a` from -- row 0 -- offers `b` instead)
__device__ __forceinline__ float rows_up(float a, float b, int up_addr, bool first_row) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(up_addr, __float_as_int(first_row ? b : a)));
}
__device__ __forceinline__ bool n_tile_nonzero(int k, int j) { return !((k == 0 && j == 0) || (k == 1 && j == 0) || (k == 1 && j == 1)); }
__device__ __forceinline__ bool b_tile_nonzero(int k, int j) { return !(k == 0 && j == 1); }

// Diagnostic build (-DNMPC_WB_STAMPS, tools/wb_stamps.py): cycle counters of the segments of a backward stage and of
// the phases, summed per wave and left in the first Js record of the problem.  The production kernel has no stamp.
#ifdef NMPC_WB_STAMPS
#define WB_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t1_ = __builtin_readcyclecounter(); \
                         st_acc[i] += t1_ - st_t0; st_t0 = t1_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WB_STAMP(i)
#endif

// QP + step of one SQP iteration: one problem per wavefront.
__global__ __launch_bounds__(64, 1) void nmpc_wb_qp_kernel(const WbArgs a) {
#ifdef NMPC_WB_STAMPS
    unsigned long long st_t0 = __builtin_readcyclecounter(), st_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b >= a.B) return;
    if (a.skip && (a.skip[b] & a.skip_mask) != 0) return;
    const int lane = lane_id();
    const int q4 = lane >> 4, c = lane & 15;
    const int N = a.N;
    const WsLayout wl(N);
    float* ws = a.ws + (size_t)b * wl.stride;
    int* flag = reinterpret_cast<int*>(ws + wl.flag);
    if (a.it > 0 && flag[0]) return;
    const WbLds L(N);
    const StageArr SA(N);
    const int NS = SA.NS;
    float* arr = ws + wl.arr;
    float* dX = arr + SA.dX;   float* dU = arr + SA.dU;
    float* dXp = arr + SA.dXp; float* dUp = arr + SA.dUp;
    float* sv = arr + SA.sv;   float* lv = arr + SA.lv;
    float* colU = smem + L.colU; float* hbuf = smem + L.hbuf;
    float* recb = smem + L.recb; float* ipm = smem + L.ipm; float* klds = smem + L.klds;
    const float* recs = ws + wl.rec;
    float* Qimg = ws + wl.qt;
    float* Kimg = ws + wl.kt;
    float* Xg = a.X + (size_t)b * (N + 1) * NX;
    float* Ug = a.U + (size_t)b * N * NU;
    const float* x0 = a.x0 + (size_t)b * NX;
    const ModelParams& mp = a.mp;
    const float dt = mp.dt;
#define AT(arr_, k_, i_) (arr_)[(i_) * NS + (k_)]
    auto phase_sync = [&]() { __threadfence_block(); wave_sync(); };

    // ---------------------------------------------------------------- prologue: Q~_k = Js'Js + diag (+ gradient)
    // the Gauss-Newton contraction J'WJ of the dense residual rows on the matrix pipe
    float qdiag[XT], qdiag_e[XT];      // diagonal element this lane may hold in tile (I, I): weight + reg
#pragma unroll
    for (int i = 0; i < XT; ++i) {
        const int row = 16 * i + c;
        const bool mine = (c >> 2) == q4 && state_at(row) >= 0;
        qdiag[i] = mine ? (row < 36 ? wdiag(a, row < 36 ? row : 0, false) : 0.0f) + a.reg : 0.0f;
        qdiag_e[i] = mine ? (row < 36 ? wdiag(a, row < 36 ? row : 0, true) : 0.0f) + a.reg_e : 0.0f;
    }
    // The compact Jacobian record and the gradient entries of node k + 1 are requested while node k is contracted (one node per
    // iteration with its loads at the top waited a full memory latency per node).  A node's record goes through the LDS: two
    // 16 B pieces per lane in, then every lane gathers the 24 elements of its six operand tiles at precomputed indices (cj_index;
    // structural zeros read a slot that holds zero).
    float* const cjbuf = colU;                      // CJ_FLOATS + the zero slot; the elimination's columns are not in use yet
    constexpr int CJ_ZERO = CJ_FLOATS;
    static_assert(CJ_FLOATS + 4 <= 32 * LDU && CJ_FLOATS <= 2 * 64 * 4, "record buffer");
    int jIdx[JT][XT][4];
#pragma unroll
    for (int t = 0; t < JT; ++t)
#pragma unroll
        for (int j = 0; j < XT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = cj_index(16 * t + 4 * q4 + r, 16 * j + c);
                jIdx[t][j][r] = ci >= 0 ? ci : CJ_ZERO;
            }
    if (lane == 0) *reinterpret_cast<f32x4*>(cjbuf + CJ_ZERO) = zero4();
    const bool cj_second = 4 * (lane + 64) < CJ_FLOATS;      // the record is 97 pieces: lanes 0..32 carry a second one
    f32x4 cjA, cjB, gcol_n[XT];
    float grow_n[XT];
    auto request_node = [&](int k) {
        const float* js = ws + wl.js + (size_t)k * JS_FLOATS;
        const float* rec = recs + (size_t)k * REC;
        cjA = *reinterpret_cast<const f32x4*>(js + 4 * lane);
        cjB = *reinterpret_cast<const f32x4*>(js + (cj_second ? 4 * (lane + 64) : 0));
        // gradient of the diagonal residuals: column HX (lanes c == 10 of tile column 2) and row HX
#pragma unroll
        for (int i = 0; i < XT; ++i) {
            const int row0 = 16 * i + 4 * q4;
            gcol_n[i] = *reinterpret_cast<const f32x4*>(rec + R_GQ + (row0 < 36 ? row0 : 0));
        }
#pragma unroll
        for (int j = 0; j < XT; ++j) { const int col = 16 * j + c; grow_n[j] = rec[R_GQ + (col < 36 ? col : 0)]; }
    };
    request_node(0);
    for (int k = 0; k <= N; ++k) {
        const bool term = (k == N);
        f32x4 J[JT][XT], gcol[XT];
        float grow[XT];
        *reinterpret_cast<f32x4*>(cjbuf + 4 * lane) = cjA;
        if (cj_second) *reinterpret_cast<f32x4*>(cjbuf + 4 * (lane + 64)) = cjB;
#pragma unroll
        for (int i = 0; i < XT; ++i) { gcol[i] = gcol_n[i]; grow[i] = grow_n[i]; }
        request_node(k < N ? k + 1 : N);
        wave_sync();
#pragma unroll
        for (int t = 0; t < JT; ++t)
#pragma unroll
            for (int j = 0; j < XT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) J[t][j][r] = cjbuf[jIdx[t][j][r]];
        wave_sync();      // (the next node's record is written behind these reads: the LDS serves a wave in order)
        // mixed precision (BASELINE configs[4]): the scaled Jacobian rounded to bf16 -- or split into a bf16 head and a
        // bf16 tail, J = hi + lo -- and contracted on the bf16 matrix pipe with fp32 accumulation
        // (v_mfma_f32_16x16x16_bf16: one instruction per 16 residual rows where fp32 takes four steps); the accumulator
        // layout of a tile is that instruction's operand layout (nmpc_tile.hpp).  Everything downstream stays fp32.
        s16x4 Jh[JT][XT], Jm[JT][XT], Jl[JT][XT];
        if (a.precision != 0) {
            auto widen = [](s16x4 v) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = __uint_as_float(((unsigned)(unsigned short)v[r]) << 16);
                return o;
            };
#pragma unroll
            for (int t = 0; t < JT; ++t)
#pragma unroll
                for (int j = 0; j < XT; ++j) {
                    Jh[t][j] = to_bf16x4(J[t][j]);
                    const f32x4 r1 = J[t][j] - widen(Jh[t][j]);          // exact: the head's 8 bits leave a 16-bit remainder
                    Jm[t][j] = to_bf16x4(r1);                            // precision 2 calls this part "lo"
                    Jl[t][j] = to_bf16x4(r1 - widen(Jm[t][j]));          // precision 3 only: J = hi + mid + lo to 24 bits
                }
        }
        // diagonal, gradient column / row, and store -- called right behind the products of each precision branch, so
        // that the MFMA results are consumed in the block that produced them
        auto finish = [&](int i, int j, f32x4 acc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * i + 4 * q4 + r, col = 16 * j + c;
                float v = acc[r];
                if (i == j && r == (c & 3)) v += term ? qdiag_e[i] : qdiag[i];      // row == col <=> r == c - 4q
                if (col == HX && row < 36) v += gcol[i][r];
                if (row == HX && col < 36) v += grow[j];
                if (row == HX && col == HX) v = 0.0f;
                acc[r] = v;
            }
            store_tile(Qimg + (size_t)k * QT_FLOATS + (i * XT + j) * IMG, lane, acc);
        };
        if (a.precision == 0) {
#pragma unroll
            for (int i = 0; i < XT; ++i)
#pragma unroll
                for (int j = 0; j < XT; ++j) {
                    if (j > i && !term) continue;      // the sweep reads the lower tiles of a running node, all nine of the terminal one
                    f32x4 acc = zero4();
#pragma unroll
                    for (int t = 0; t < JT; ++t) acc = xty(J[t][i], J[t][j], acc);
                    finish(i, j, acc);
                }
        } else if (a.precision != 3) {
#pragma unroll
            for (int i = 0; i < XT; ++i)
#pragma unroll
                for (int j = 0; j < XT; ++j) {
                    if (j > i && !term) continue;      // the sweep reads the lower tiles of a running node, all nine of the terminal one
                    f32x4 acc = zero4();
#pragma unroll
                    for (int t = 0; t < JT; ++t) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Jh[t][i], Jh[t][j], acc, 0, 0, 0);
                        if (a.precision == 2) {      // (hi + lo)'(hi + lo) without the lo'lo term (2^-16 of the product)
                            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Jh[t][i], Jm[t][j], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Jm[t][i], Jh[t][j], acc, 0, 0, 0);
                        }
                    }
                    finish(i, j, acc);
                }
        } else {
            // three-way split: J = hi + mid + lo carries the 24 bits of the fp32 Jacobian; of the nine products the six of
            // relative size >= 2^-16 are formed (mid'lo, lo'mid, lo'lo are below fp32 rounding), smallest first, each
            // bf16 x bf16 product exact in the fp32 accumulator: the Hessian differs from the fp32 contraction by
            // accumulation order only.  Six instructions of K = 16 per 16 residual rows against four fp32 steps of K = 4.
#pragma unroll
            for (int i = 0; i < XT; ++i)
#pragma unroll
                for (int j = 0; j < XT; ++j) {
                    if (j > i && !term) continue;      // the sweep reads the lower tiles of a running node, all nine of the terminal one
                    f32x4 acc = zero4();
#pragma unroll
                    for (int t = 0; t < JT; ++t) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Jl[t][i], Jh[t][j], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Jh[t][i], Jl[t][j], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Jm[t][i], Jm[t][j], acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < JT; ++t) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Jm[t][i], Jh[t][j], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Jh[t][i], Jm[t][j], acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < JT; ++t)
                        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Jh[t][i], Jh[t][j], acc, 0, 0, 0);
                    finish(i, j, acc);
                }
        }
    }
    WB_STAMP(12);
    // cost, active rows, cold start of the interior point: s = max(-c, s_min), lam = mu0 / s
    float cost_l = 0.0f, mu_l = 0.0f;
    int nact_l = 0;
    unsigned my_act = 0u;
    for (int k = lane; k <= N; k += 64) cost_l += recs[(size_t)k * REC + R_COST];     // (N = 64: node 64 has no lane of its own)
    float s_init[NG], l_init[NG], c_init[NG], r_init[NU];      // (the first sweep's barrier terms are formed from these, below)
    {
        const float* rec = recs + (size_t)(lane < N ? lane : 0) * REC;
        if (lane < N) {
            my_act = reinterpret_cast<const unsigned*>(rec)[R_ACT];
            nact_l = __popc(my_act);
            mu_l = a.mu0 * (float)nact_l;
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) r_init[i] = rec[R_R + i];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            c_init[j] = rec[R_C + j];
            s_init[j] = fmaxf(-c_init[j], a.s_min);
            l_init[j] = a.mu0 * fast_rcp(s_init[j]);
            if (lane < N) { AT(sv, lane, j) = s_init[j]; AT(lv, lane, j) = l_init[j]; }
        }
    }
    const float cost = wave_sum(cost_l);
    const int n_act = (int)(wave_sum((float)nact_l) + 0.5f);
    float mu_sum = wave_sum(mu_l);
    const bool use_ipm = (a.n_ipm > 0) && (n_act > 0);
    const int n_sweeps = use_ipm ? a.n_ipm : 1;
    phase_sync();

    // barrier-modified input terms of stage `lane`, lane = stage: rt = r + G'(tau/s + lam + D c), Rf = G'DG per foot
    // barrier-modified input terms of stage `lane` into its LDS row, from the stage's slacks, multipliers, constraint values and
    // input gradient in registers
    auto ipm_terms = [&](float tau, const float (&sj)[NG], const float (&lj)[NG], const float (&cj)[NG], const float (&rr)[NU]) {
        if (lane < N) {
            float* row = ipm + lane * IPMW;
#pragma unroll
            for (int i = 0; i < WF; ++i) row[i] = rr[i];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                float D[4], v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool on = use_ipm && ((my_act >> (4 * f + j)) & 1u);
                    const float s = sj[4 * f + j], l = lj[4 * f + j];
                    const float is = fast_rcp(s);
                    D[j] = on ? l * is : 0.0f;
                    v[j] = on ? tau * is + l + l * is * cj[4 * f + j] : 0.0f;
                }
                row[WF + 3 * f + 0] = rr[WF + 3 * f + 0] + (v[0] - v[1]);
                row[WF + 3 * f + 1] = rr[WF + 3 * f + 1] + (v[2] - v[3]);
                row[WF + 3 * f + 2] = rr[WF + 3 * f + 2] - mp.mu * ((v[0] + v[1]) + (v[2] + v[3]));
                row[IPM_RF + 5 * f + 0] = D[0] + D[1];
                row[IPM_RF + 5 * f + 1] = D[2] + D[3];
                row[IPM_RF + 5 * f + 2] = mp.mu * mp.mu * ((D[0] + D[1]) + (D[2] + D[3]));
                row[IPM_RF + 5 * f + 3] = -mp.mu * (D[0] - D[1]);
                row[IPM_RF + 5 * f + 4] = -mp.mu * (D[2] - D[3]);
            }
        }
    };

    if (lane < N) ipm[lane * IPMW + IPMW - 1] = 0.0f;
    // the weight-derived constants of the backward sweep (loads from the argument block: once per kernel, unlike the index maps)
    float rdiag_w[UT][4], rs_free_w[12];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < UT; ++i) {
            const int uu = 16 * i + 4 * q4 + r;
            float w = 0.0f;
            if (uu >= 6 && uu < 18) w = a.W[RY_ACC + (uu >= 6 && uu < 18 ? uu - 6 : 0)];
            if (uu >= WF && uu < NU) w = a.W[RY_FREG + (uu >= WF && uu < NU ? uu - WF : 0)];
            rdiag_w[i][r] = (uu == 16 * i + c && uu < NU) ? w + a.reg : 0.0f;
        }
#pragma unroll
    for (int i = 0; i < 12; ++i) rs_free_w[i] = __builtin_amdgcn_rsqf((a.W[RY_FREG + i] + a.reg) + 0.0f);
    // barrier terms of the first sweep, from the registers of the cold start (read back from memory they cost a store -> load trip)
    ipm_terms(use_ipm ? fmaxf(a.sigma * mu_sum / (float)n_act, a.tau_min) : 0.0f, s_init, l_init, c_init, r_init);
    wave_sync();
    bool qp_ok = true;
    // What a backward sweep starts from -- terminal P~ (nine tiles), the records of stages N-1 and N-2, the lower Q~ tiles of stage
    // N-1 -- is the same for every sweep of a call and is requested at the top of the interior-point phase BEFORE it, so that the
    // sweep does not open with a memory latency.
    f32x4 P[XT][XT], Qn[XT][XT], rec_first, rec_pref;
    auto request_sweep = [&]() {
#pragma unroll
        for (int i = 0; i < XT; ++i)
#pragma unroll
            for (int j = 0; j < XT; ++j) P[i][j] = load_tile(Qimg + (size_t)N * QT_FLOATS + (i * XT + j) * IMG, lane);
        rec_first = *reinterpret_cast<const f32x4*>(recs + (size_t)(N - 1) * REC + (4 * lane < REC ? 4 * lane : 0));
        rec_pref = *reinterpret_cast<const f32x4*>(recs + (size_t)(N > 1 ? N - 2 : 0) * REC + (4 * lane < REC ? 4 * lane : 0));
#pragma unroll
        for (int i = 0; i < XT; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) Qn[i][j] = load_tile(Qimg + (size_t)(N - 1) * QT_FLOATS + (i * XT + j) * IMG, lane);
    };
    request_sweep();
    for (int ii = 0; ii < n_sweeps; ++ii) {
        const float tau = use_ipm ? fmaxf(a.sigma * mu_sum / (float)n_act, a.tau_min) : 0.0f;
        // Everything below that is derived from the lane index -- the index maps of the tile synthesis, LDS addresses of the layout
        // changes, masks -- is recomputed at the top of every sweep from a copy of the lane index the optimiser cannot see through:
        // ~150 integer operations per sweep.  Computed once per kernel (where loop-invariant code motion puts them) these ~100
        // registers stay live through the forward sweep and the interior-point phase, which then keep their own working set --
        // the gain rows of the next two stages -- in the accumulation registers and move every element back before use.
        int lane_opaque = lane;
        asm volatile("" : "+v"(lane_opaque));
        const int lane = lane_opaque, q4 = lane >> 4, c = lane & 15;
        // per-lane constants of the tile synthesis: every element of N~ = A~ - I, B~, R~ and S~ is ONE LDS read at a
        // precomputed index -- into the stage record (defect, momentum Jacobians, or one of its constant slots 0, dt, dt^2)
        // or into the stage's row of barrier-modified input terms.  (Lane predicates instead of indices cost an SGPR pair
        // per element and tile: the loop-invariant masks spilled 626 SGPRs into VGPR lanes.)
        constexpr int NT = 6, BT = 5;
        // non-zero tiles of N~: (0,1) (0,2) (1,2) (2,0) (2,1) (2,2) ; of B~: (0,0) (1,0) (1,1) (2,0) (2,1)
        int nIdx[NT][4], bIdx[BT][4], rIdx[4], sIdx[UT][4];
        float rdiag[UT][4];
        {
            constexpr int nt_i[NT] = {0, 0, 1, 2, 2, 2}, nt_j[NT] = {1, 2, 2, 0, 1, 2};
            constexpr int bt_i[BT] = {0, 1, 1, 2, 2}, bt_j[BT] = {0, 0, 1, 0, 1};
    #pragma unroll
            for (int t = 0; t < NT; ++t)
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * nt_i[t] + 4 * q4 + r, col = 16 * nt_j[t] + c;
                    int idx = R_ZERO;
                    if (row < 18 && col == row + 18) idx = R_DT;
                    const int sr = state_at(row);
                    if (col == HX && sr >= 0) idx = R_D + row;
                    if (sr >= 39 && sr < 42 && col >= 3 && col < 18) idx = R_HQ + (sr - 39) * 16 + (col - 3);
                    nIdx[t][r] = idx;
                }
    #pragma unroll
            for (int t = 0; t < BT; ++t)
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * bt_i[t] + 4 * q4 + r, uc = 16 * bt_j[t] + c;
                    int idx = R_ZERO;
                    if (uc < 18 && row == uc) idx = R_DT2;
                    // (dt at (v_16, a_16), (v_17, a_17) of tile (2,1) is applied without the matrix pipe, so that the tile's only non-zero
                    //  rows are momentum rows: contraction steps 0, 1)
                    if (uc < 18 && row == uc + 18 && !(bt_i[t] == 2 && bt_j[t] == 1)) idx = R_DT;
                    if (uc >= WF && uc < NU) {
                        const int f = uc - WF, sr = state_at(row);
                        if (sr >= 36 && sr < 39 && sr - 36 == f % 3) idx = R_CDT + f / 3;
                        if (sr >= 39 && sr < 42) idx = R_HF + (sr - 39) * 12 + f;
                    }
                    bIdx[t][r] = idx;
                }
            // input-cost tile (1,1): the barrier block of the foot, and the diagonal W_acc / W_cnt_f_reg + reg of (0,0), (1,1)
    #pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ur = 16 + 4 * q4 + r, uc = 16 + c;
                int idx = IPMW - 1;                                  // a slot that holds zero
                if (ur >= WF && ur < NU && uc >= WF && uc < NU) {
                    const int fr = ur - WF, fc = uc - WF;
                    if (fr / 3 == fc / 3) {
                        const int ar = fr % 3, ac = fc % 3;
                        const int sel = (ar == ac) ? ar : (ar + ac == 2) ? 3 : (ar + ac == 3) ? 4 : -1;
                        if (sel >= 0) idx = IPM_RF + 5 * (fr / 3) + sel;
                    }
                }
                rIdx[r] = idx;
    #pragma unroll
                for (int i = 0; i < UT; ++i) {
                    const int uu = 16 * i + 4 * q4 + r;
                    rdiag[i][r] = rdiag_w[i][r];
                    sIdx[i][r] = (c == HX - 32 && uu < NU) ? uu : IPMW - 1;      // S~: the input gradient rides in column HX
                }
            }
        }
        // constants of the identity blocks of B~ (see the backward stage): dt^2, on columns 0, 1 of a tile, on rows 0, 1 of a tile
        // 1 / sqrt of the pivot of a decoupled force input, by component: Huu[j][j] = (W_f_reg + reg) + 0 there (see ldl_pivots)
        float rs_free[12];
    #pragma unroll
        for (int i = 0; i < 12; ++i) rs_free[i] = rs_free_w[i];
        const float dt2 = dt * dt;
        const float dt2_c01 = (c < 2) ? dt2 : 0.0f, dt_c01 = (c < 2) ? dt : 0.0f;
        const int up_addr = 4 * ((lane + 16) & 63);
        const int down_addr = 4 * ((lane + 48) & 63);               // the lane one lane row (16 lanes) before this one
        const float dt_qn0 = (q4 > 0) ? dt : 0.0f;
        const bool first_row = lane < 16;
        const int x16_addr = 4 * (lane ^ 16), x32_addr = 4 * (lane ^ 32);
        const float dt_q0 = (q4 == 0) ? dt : 0.0f;
        const bool hx_col = (c == HX - 32), hx_row = (q4 == HXQ);     // HX = 38: column 6 of tile column 2; row 6 of tile row 2 = lane row 1, register 2
        // where this lane finds, in the stage record, the defects of its columns (one per tile column) ...
        int dcolIdx[XT];
    #pragma unroll
        for (int kk = 0; kk < XT; ++kk) dcolIdx[kk] = R_D + 16 * kk + c;      // (the record holds zeros where no state sits)
        float dt2_r01[4];
    #pragma unroll
        for (int r = 0; r < 4; ++r) dt2_r01[r] = (4 * q4 + r < 2) ? dt2 : 0.0f;

        // ------------------------------------------------------------ phase R: backward sweep
        // (terminal P~, the records of stages N-1 and N-2 and the Q~ tiles of stage N-1 were requested by request_sweep())
        // record of stage N-1 into the LDS; each stage prefetches the next one's (global -> registers -> LDS)
        *reinterpret_cast<f32x4*>(recb + 4 * lane) = rec_first;
        // The record of stage k-1 is requested in the middle of stage k+1, next to the Q~ tiles -- BEFORE that stage's K~ stores.
        // vmcnt retires in order: requested at the top of stage k (behind the six K~ stores of stage k+1), waiting for it meant
        // waiting for those stores to reach memory, every stage (s_waitcnt vmcnt(0) in the middle of the stage).
        wave_sync();
        // Q~ tiles (lower ones) are requested one and a half stages ahead of their use: for stage k-1 in the middle of
        // stage k.  (A timing build that reads one cache-resident image instead of each stage's own ran 10 % faster:
        // requested at the top of their own stage, the 6 KB did not arrive by the time H needs them.)
        WB_STAMP(21);          // terminal tiles, first record and Q~ tiles requested
        for (int k = N - 1; k >= 0; --k) {
            const int kn = k > 0 ? k - 1 : 0;
            WB_STAMP(19);      // the loop's back edge
            // prefetch: next record, this stage's Q~ tiles were requested ... (Q of stage k is loaded here; the
            // loads are issued first and consumed after the P~A~, P~B~ products)
            const f32x4 rec_next = rec_pref;
            f32x4 Q[XT][XT];
#pragma unroll
            for (int i = 0; i < XT; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) Q[i][j] = Qn[i][j];
            const float* rk = recb;
            const float* ik = ipm + k * IPMW;
            // contact pattern of the stage from the record's dt c_f (wave-uniform; read now: the record is replaced mid-stage)
            const unsigned pat = __builtin_amdgcn_readfirstlane((rk[R_CDT] > 0.0f ? 1u : 0u) | (rk[R_CDT + 1] > 0.0f ? 2u : 0u) |
                                                                (rk[R_CDT + 2] > 0.0f ? 4u : 0u) | (rk[R_CDT + 3] > 0.0f ? 8u : 0u));
            WB_STAMP(0);
            // ---- synthesise N~ = A~ - I and B~ tiles from the record (one LDS read per element)
            f32x4 Nt[XT][XT], Bt[XT][UT];
            {
                constexpr int nt_i[NT] = {0, 0, 1, 2, 2, 2}, nt_j[NT] = {1, 2, 2, 0, 1, 2};
                constexpr int bt_i[BT] = {0, 1, 1, 2, 2}, bt_j[BT] = {0, 0, 1, 0, 1};
                Nt[0][0] = Nt[1][0] = Nt[1][1] = zero4();
                Bt[0][1] = zero4();
                Nt[0][1] = Nt[0][2] = Nt[1][2] = Nt[2][2] = zero4();       // (their products are formed without the matrix pipe, see below)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (nt_j[t] != 2 && nt_i[t] == 2) {      // the dense tiles: momentum rows against q (2,0), (2,1)
#pragma unroll
                        for (int r = 0; r < 4; ++r) Nt[nt_i[t]][nt_j[t]][r] = rk[nIdx[t][r]];
                    }
#pragma unroll
                for (int t = 0; t < BT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Bt[bt_i[t]][bt_j[t]][r] = rk[bIdx[t][r]];
            }
            // the defect of the stage: by column of this lane (P~ d) and by row quad of this lane (d'(P~A~)); d[42], d[43] are zero
            float dcol[XT], drow[XT][4];
#pragma unroll
            for (int kk = 0; kk < XT; ++kk) {
                dcol[kk] = rk[dcolIdx[kk]];
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(rk + R_D + 16 * kk + 4 * q4);
#pragma unroll
                for (int r = 0; r < 4; ++r) drow[kk][r] = d4[r];
            }
            WB_STAMP(1);
            // ---- P~A~ = P~ + P~N~ ,  P~B~
            // Where a tile of N~ or B~ holds nothing but the integrator's identities -- N~(0,1): dt at (q_r, v_r); B~(.,0): dt^2 at
            // (q_c, a_c) and dt at (v_c, a_c) -- the product with P~ is a COLUMN SHIFT of P~ by 18 = 16 + 2 columns: a lane shift
            // by two inside the 16-lane rows of the accumulator layout (DPP row_shr / row_shl, no matrix instruction).  Every
            // output element has the same non-zero terms in the same order as the MFMA chain had (its other terms were exact
            // zeros), so the results are bit-identical: 48 of the 132 MFMAs of this segment become 48 VALU operations.
            f32x4 PA[XT][XT], PB[XT][UT];
#pragma unroll
            for (int i = 0; i < XT; ++i) {
#pragma unroll
                for (int j = 0; j < XT; ++j) {
                    f32x4 acc = P[i][j];
                    if (j == 1) {            // kk = 0: (P~ N~)[:, 18 + c] = dt P~[:, c], c <= 13
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(dt, dpp_row<0x112>(P[i][0][r]), acc[r]);      // row_shr:2
                    }
                    if (j == 2) {
                        // the tiles N~(.,2) hold dt at (q_14..q_17, v_14..v_17) -- column shifts as above -- and the defect d in the
                        // homogeneous column: (P~ N~)[:, HX] = P~ d, one column from a contraction over all 42 states.  On the
                        // matrix pipe that column costs 36 MFMAs per stage; here every lane multiplies its row quad of P~ with the
                        // defects of its columns (three fma per register) and the 16 lanes of a row add up (four DPP steps)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc[r] = __builtin_fmaf(dt, dpp_row<0x10E>(P[i][0][r]), acc[r]);              // row_shl:14: columns 14, 15 -> 32, 33
                            acc[r] = __builtin_fmaf(dt, dpp_row_bank0<0x112>(P[i][1][r]), acc[r]);        // row_shr:2, lanes 2, 3: columns 16, 17 -> 34, 35
                            float t = P[i][0][r] * dcol[0];
                            t = __builtin_fmaf(P[i][1][r], dcol[1], t);
                            t = __builtin_fmaf(P[i][2][r], dcol[2], t);
                            t = row_sum16(t);
                            acc[r] += hx_col ? t : 0.0f;
                        }
                    }
#pragma unroll
                    for (int kk = 0; kk < XT; ++kk)
                        if (n_tile_nonzero(kk, j) && !(kk == 0 && j == 1) && j != 2) {
                            static_assert(XT == 3, "the dense tiles of N~ are those of tile row 2");
                            if (kk == 2) acc = xty01(P[kk][i], Nt[kk][j], acc);      // momentum rows only: steps 0, 1
                        }
                    PA[i][j] = acc;
                }
                {   // j = 0 (inputs a_0..a_15): (P~ B~)[:, a_c] = dt^2 P~[:, c] + dt P~[:, 18 + c]
                    f32x4 acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = dt2 * P[i][0][r];
                        t = __builtin_fmaf(dt, dpp_row<0x102>(P[i][1][r]), t);      // row_shl:2: columns 18..31 of P~ onto lanes 0..13
                        t = __builtin_fmaf(dt, dpp_row<0x11E>(P[i][2][r]), t);      // row_shr:14: columns 32, 33 onto lanes 14, 15
                        acc[r] = t;
                    }
                    PB[i][0] = acc;
                }
                {   // j = 1 (a_16, a_17, f): B~(1,1) holds nothing but dt^2 at (q_16, a_16), (q_17, a_17) -- columns 16, 17 of P~ in
                    // place, scaled; the rest (dt at v_16, v_17 and the wrench map of the forces) is the tile B~(2,1)
                    f32x4 acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[r] = dt2_c01 * P[i][1][r];
                        acc[r] = __builtin_fmaf(dt_c01, dpp_row<0x102>(P[i][2][r]), acc[r]);      // dt at (v_16, a_16), (v_17, a_17): columns 34, 35 onto lanes 0, 1
                    }
                    PB[i][1] = xty01(P[2][i], Bt[2][1], acc);      // the wrench map of the forces: momentum rows only
                }
            }
            WB_STAMP(2);
            // ---- H~ux = S~ + B~'(P~A~) (stays in registers) ,  Huu = R~ + B~'(P~B~)  -> LDS columns
            // (products are written step by step over several output tiles at once: a dependent fp32 MFMA waits 40
            //  cycles for its accumulator, an independent one issues after 32 -- tools/wb_stamps.py)
            f32x4 Hux[UT][XT], Huu[UT][UT];
#pragma unroll
            for (int i = 0; i < UT; ++i) {
#pragma unroll
                for (int j = 0; j < XT; ++j) {
                    Hux[i][j] = zero4();
                    if (j == 2) {   // S~: the input gradient rides in column HX
#pragma unroll
                        for (int r = 0; r < 4; ++r) Hux[i][j][r] = ik[sIdx[i][r]];
                    }
                }
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    Huu[i][j] = zero4();
                    if (i == j) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) Huu[i][j][r] = rdiag[i][r] + (i == 1 ? ik[rIdx[r]] : 0.0f);
                    }
                }
                // kk = i: B~(i,i) is dt^2 on its diagonal (all of it for i = 0, rows a_16, a_17 for i = 1) -- B~(i,i)' X is X scaled
                // in place, the first term of every element as in the MFMA chain it replaces
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sc = (i == 0) ? dt2 : dt2_r01[r];
#pragma unroll
                    for (int j = 0; j < XT; ++j) Hux[i][j][r] = __builtin_fmaf(sc, PA[i][j][r], Hux[i][j][r]);
#pragma unroll
                    for (int j = 0; j <= i; ++j) Huu[i][j][r] = __builtin_fmaf(sc, PB[i][j][r], Huu[i][j][r]);
                }
                if (i == 0) {
                    // kk = 1, 2 for the inputs a_0..a_15: B~(1,0), B~(2,0) hold nothing but dt at (v_c, a_c) -- row 18 + c of X onto
                    // row c.  In the accumulator layout rows 18 + 4q + r are registers 2, 3 of the same lane (r = 0, 1) and
                    // registers 0, 1 one lane row (16 lanes) further on (r = 2, 3; the last lane row takes rows 32, 33 from tile
                    // row 2): one crossbar move per such register, no matrix instruction
#pragma unroll
                    for (int j = 0; j < XT; ++j) {
                        Hux[0][j][0] = __builtin_fmaf(dt, PA[1][j][2], Hux[0][j][0]);
                        Hux[0][j][1] = __builtin_fmaf(dt, PA[1][j][3], Hux[0][j][1]);
                        Hux[0][j][2] = __builtin_fmaf(dt, rows_up(PA[1][j][0], PA[2][j][0], up_addr, first_row), Hux[0][j][2]);
                        Hux[0][j][3] = __builtin_fmaf(dt, rows_up(PA[1][j][1], PA[2][j][1], up_addr, first_row), Hux[0][j][3]);
                    }
                    Huu[0][0][0] = __builtin_fmaf(dt, PB[1][0][2], Huu[0][0][0]);
                    Huu[0][0][1] = __builtin_fmaf(dt, PB[1][0][3], Huu[0][0][1]);
                    Huu[0][0][2] = __builtin_fmaf(dt, rows_up(PB[1][0][0], PB[2][0][0], up_addr, first_row), Huu[0][0][2]);
                    Huu[0][0][3] = __builtin_fmaf(dt, rows_up(PB[1][0][1], PB[2][0][1], up_addr, first_row), Huu[0][0][3]);
                }
                if (i == 1) {
                    // kk = 2, inputs a_16, a_17: dt at (v_16, a_16), (v_17, a_17) -- rows 34, 35 of X (registers 2, 3 of lane row 0 of tile
                    // row 2) onto rows a_16, a_17 (registers 0, 1 of lane row 0)
#pragma unroll
                    for (int j = 0; j < XT; ++j) {
                        Hux[1][j][0] = __builtin_fmaf(dt_q0, PA[2][j][2], Hux[1][j][0]);
                        Hux[1][j][1] = __builtin_fmaf(dt_q0, PA[2][j][3], Hux[1][j][1]);
                    }
#pragma unroll
                    for (int j = 0; j <= 1; ++j) {
                        Huu[1][j][0] = __builtin_fmaf(dt_q0, PB[2][j][2], Huu[1][j][0]);
                        Huu[1][j][1] = __builtin_fmaf(dt_q0, PB[2][j][3], Huu[1][j][1]);
                    }
                }
#pragma unroll
                for (int kk = 0; kk < XT; ++kk)
                    if (b_tile_nonzero(kk, i) && kk != i && i != 0) {
                        static_assert(XT == 3, "the one dense tile of B~ left for the matrix pipe is (2,1): momentum rows, steps 0, 1");
#pragma unroll
                        for (int st = 0; st < 2; ++st) {
#pragma unroll
                            for (int j = 0; j < XT; ++j) Hux[i][j] = mfma4(Bt[kk][i][st], PA[kk][j][st], Hux[i][j]);
#pragma unroll
                            for (int j = 0; j <= i; ++j) Huu[i][j] = mfma4(Bt[kk][i][st], PB[kk][j][st], Huu[i][j]);
                        }
                    }
                // lower tiles only; lane L > j of the elimination reads row j of column L (upper triangle), so the
                // off-diagonal tile is written a second time, transposed, into the (0,1) position
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    *reinterpret_cast<f32x4*>(colU + (16 * j + c) * LDU + 16 * i + 4 * q4) = Huu[i][j];
                    if (i != j) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) colU[(16 * i + 4 * q4 + r) * LDU + 16 * j + c] = Huu[i][j][r];
                    }
                }
            }
            WB_STAMP(3);
            // ---- H = Q~ + P~A~ + N~'(P~A~), lower tiles only; the upper ones and the symmetrisation of the diagonal
            // tiles come back transposed from the LDS: H~xx is symmetric bit for bit
            f32x4 H[XT][XT];
            // registers 2, 3 of tile row 0 of P~A~, one lane row back (lane row 0 reads lane row 3): rows q_2.. onto rows v_2.. of tile row 1
            // and rows q_14, q_15 onto v_14, v_15 of tile row 2 -- one crossbar move per register serves both
            float pa_dn2[XT], pa_dn3[XT];
#pragma unroll
            for (int j = 0; j < XT; ++j) {
                pa_dn2[j] = __int_as_float(__builtin_amdgcn_ds_bpermute(down_addr, __float_as_int(PA[0][j][2])));
                pa_dn3[j] = __int_as_float(__builtin_amdgcn_ds_bpermute(down_addr, __float_as_int(PA[0][j][3])));
            }
#pragma unroll
            for (int i = 0; i < XT; ++i) {
#pragma unroll
                for (int j = 0; j <= i; ++j) H[i][j] = Q[i][j] + PA[i][j];
                if (i == 2) {
                    // N~(.,2)'(P~A~): rows v_14..v_17 = dt x rows q_14..q_17 of P~A~ (quad 3 of tile row 0 and quad 0 of tile row 1 onto
                    // quad 0 of tile row 2), and the homogeneous row = d'(P~A~): every lane adds up its row quads against the
                    // defects of its rows (twelve fma per column tile), the four lane rows of a column meet through the crossbar
#pragma unroll
                    for (int j = 0; j < XT; ++j) {
                        const float v14 = pa_dn2[j], v15 = pa_dn3[j];
                        H[2][j][0] = __builtin_fmaf(dt_q0, v14, H[2][j][0]);
                        H[2][j][1] = __builtin_fmaf(dt_q0, v15, H[2][j][1]);
                        H[2][j][2] = __builtin_fmaf(dt_q0, PA[1][j][0], H[2][j][2]);
                        H[2][j][3] = __builtin_fmaf(dt_q0, PA[1][j][1], H[2][j][3]);
                        float t = 0.0f;
#pragma unroll
                        for (int kk = 0; kk < XT; ++kk)
#pragma unroll
                            for (int r = 0; r < 4; ++r) t = __builtin_fmaf(drow[kk][r], PA[kk][j][r], t);
                        t += __int_as_float(__builtin_amdgcn_ds_bpermute(x16_addr, __float_as_int(t)));
                        t += __int_as_float(__builtin_amdgcn_ds_bpermute(x32_addr, __float_as_int(t)));
                        H[2][j][2] += hx_row ? t : 0.0f;
                    }
                } else {
                    if (i == 1) {
                        // N~(0,1)'(P~A~): N~(0,1) holds nothing but dt at (q_c, v_c), c < 14 -- rows v_c = 18 + c of H take dt x row c of P~A~:
                        // rows 4q, 4q+1 of tile row 0 onto rows 4q+2, 4q+3 of tile row 1 (registers 0, 1 -> 2, 3 of the same lane), rows
                        // 4q+2, 4q+3 onto rows 4(q+1), 4(q+1)+1 (registers 2, 3 -> 0, 1 one lane row further on: one crossbar move each)
#pragma unroll
                        for (int j = 0; j <= 1; ++j) {
                            const float d2 = pa_dn2[j], d3 = pa_dn3[j];
                            H[1][j][0] = __builtin_fmaf(dt_qn0, d2, H[1][j][0]);
                            H[1][j][1] = __builtin_fmaf(dt_qn0, d3, H[1][j][1]);
                            H[1][j][2] = __builtin_fmaf(dt, PA[0][j][0], H[1][j][2]);
                            H[1][j][3] = __builtin_fmaf(dt, PA[0][j][1], H[1][j][3]);
                        }
                    }
#pragma unroll
                    for (int kk = 0; kk < XT; ++kk)
                        if (n_tile_nonzero(kk, i) && kk == 2) {      // tile row 2 of N~: momentum rows only, contraction steps 0, 1
#pragma unroll
                            for (int st = 0; st < 2; ++st)
#pragma unroll
                                for (int j = 0; j <= i; ++j) H[i][j] = mfma4(Nt[kk][i][st], PA[kk][j][st], H[i][j]);
                        }
                }
#pragma unroll
                for (int j = 0; j <= i; ++j) *reinterpret_cast<f32x4*>(hbuf + (16 * j + c) * LDH + 16 * i + 4 * q4) = H[i][j];
            }
            // next stage's record replaces this one's (all of its reads are issued above)
            *reinterpret_cast<f32x4*>(recb + 4 * lane) = rec_next;
            wave_sync();
            WB_STAMP(4);
            // ---- LDL' of Huu in column layout applied to the identity: W = D^-1/2 L^-1
            // lanes 0..31: column `lane` of Huu; lanes 32..63: column lane-32 of I; row i in register i & 3 of Xq[i >> 2] (ldl_panel).
            // The right-hand side H~ux is NOT carried through the elimination (as a second column per lane the compiler split it into
            // a pass of its own and parked all 435 multipliers in spilled SGPRs for it): Y = W H~ux is formed on the matrix pipe below.
            f32x4 Xq[LDL_GROUPS];
            {
                const bool is_h = lane < 32;
                const float* pu = colU + (is_h ? lane : 0) * LDU;
#pragma unroll
                for (int i4 = 0; i4 < LDL_GROUPS; ++i4) {
                    const f32x4 vu = *reinterpret_cast<const f32x4*>(pu + 4 * i4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 4 * i4 + r;
                        Xq[i4][r] = (i < NU) ? (is_h ? vu[r] : ((lane - 32 == i) ? 1.0f : 0.0f)) : 0.0f;
                    }
                }
            }
            bool ok = true;
            {
                if (pat == 0x9u) ldl_panel<0, coupling_mask(0x9u)>(Xq, ok, rs_free);
                else if (pat == 0x6u) ldl_panel<0, coupling_mask(0x6u)>(Xq, ok, rs_free);
                else if (pat == 0x0u) ldl_panel<0, coupling_mask(0x0u)>(Xq, ok, rs_free);
                else ldl_panel<0, coupling_mask(0xFu)>(Xq, ok, rs_free);
            }
            qp_ok = qp_ok && ok;
#pragma unroll
            for (int i = 0; i < XT; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) Qn[i][j] = load_tile(Qimg + (size_t)
#ifdef WB_T_ONEQ      // timing build: every stage reads the Q~ tiles of stage 0 (cache-resident; results are wrong)
                                                    (a.B < 0 ? kn : 0)
#else
                                                    kn
#endif
                                                    * QT_FLOATS + (i * XT + j) * IMG, lane);
            rec_pref = *reinterpret_cast<const f32x4*>(recs + (size_t)(k > 1 ? k - 2 : 0) * REC + (4 * lane < REC ? 4 * lane : 0));
            WB_STAMP(5);
            // transposed tiles: Ht[i][j] = (lower tile (i,j))', i >= j
            f32x4 Ht[XT][XT];
#pragma unroll
            for (int i = 0; i < XT; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    const float* ph = hbuf + (16 * j + 4 * q4) * LDH + 16 * i + c;      // element (4q+r, c) of the transpose = H[16i+c][16j+4q+r]
                    Ht[i][j] = f32x4{ph[0], ph[LDH], ph[2 * LDH], ph[3 * LDH]};
                }
            wave_sync();
            // W columns back into the LDS (lanes 32..63 -> column lane-32), then as tiles: W and W'
            {
                float* pw = colU + (lane >= 32 ? lane - 32 : 0) * LDU;
#pragma unroll
                for (int i4 = 0; i4 < 8; ++i4) {
                    f32x4 vw;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 4 * i4 + r;
                        vw[r] = (i < NU) ? Xq[i4][r] : 0.0f;
                    }
                    if (lane >= 32) *reinterpret_cast<f32x4*>(pw + 4 * i4) = vw;
                }
            }
            wave_sync();
            f32x4 Wt[UT][UT], WT[UT][UT];      // W[i][j] (rows 16i.., columns 16j..) and its transpose W'[j][i] = WT[i][j]
#pragma unroll
            for (int i = 0; i < UT; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    Wt[i][j] = *reinterpret_cast<const f32x4*>(colU + (16 * j + c) * LDU + 16 * i + 4 * q4);
                    const float* pt = colU + (16 * j + 4 * q4) * LDU + 16 * i + c;      // element (4q+r, c) of W[i][j]' = W[16i+c][16j+4q+r]
                    WT[i][j] = f32x4{pt[0], pt[LDU], pt[2 * LDU], pt[3 * LDU]};
                }
            Wt[0][1] = zero4();
            wave_sync();
            WB_STAMP(6);
            // Y = W H~ux:  Y[i][j] = sum_{kk <= i} W[i][kk] H~ux[kk][j] = sum xty(W[i][kk]', H~ux[kk][j])
            f32x4 Y[UT][XT];
#pragma unroll
            for (int i = 0; i < UT; ++i)
#pragma unroll
                for (int j = 0; j < XT; ++j) Y[i][j] = zero4();
#pragma unroll
            for (int kk = 0; kk < UT; ++kk)
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int i = kk; i < UT; ++i)
#pragma unroll
                        for (int j = 0; j < XT; ++j) Y[i][j] = mfma4(WT[i][kk][st], Hux[kk][j][st], Y[i][j]);
            WB_STAMP(7);
            // ---- P~+ = H~xx - Y'Y (lower tiles; the upper ones are their transposes through the LDS) ,  K~ = -W'Y
            f32x4 nY[UT][XT];
#pragma unroll
            for (int i = 0; i < UT; ++i)
#pragma unroll
                for (int j = 0; j < XT; ++j) nY[i][j] = -Y[i][j];
#pragma unroll
            for (int i = 0; i < XT; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) P[i][j] = (i == j) ? 0.5f * (H[i][j] + Ht[i][j]) : H[i][j];
#pragma unroll
            for (int kk = 0; kk < UT; ++kk)
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int i = 0; i < XT; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j) P[i][j] = mfma4(nY[kk][i][st], Y[kk][j][st], P[i][j]);
            WB_STAMP(16);      // P~+ products
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c == HX - 32 && 32 + 4 * q4 + r == HX) P[2][2][r] = 0.0f;
#pragma unroll
            for (int i = 0; i < XT; ++i)
#pragma unroll
                for (int j = 0; j < i; ++j) *reinterpret_cast<f32x4*>(hbuf + (16 * j + c) * LDH + 16 * i + 4 * q4) = P[i][j];
            wave_sync();
#pragma unroll
            for (int i = 0; i < XT; ++i)
#pragma unroll
                for (int j = 0; j < i; ++j) {
                    const float* ph = hbuf + (16 * j + 4 * q4) * LDH + 16 * i + c;
                    P[j][i] = f32x4{ph[0], ph[LDH], ph[2 * LDH], ph[3 * LDH]};
                }
            WB_STAMP(17);      // mirror of P~+ through the LDS
            {
                f32x4 Kt[UT][XT];
#pragma unroll
                for (int i = 0; i < UT; ++i)
#pragma unroll
                    for (int j = 0; j < XT; ++j) Kt[i][j] = zero4();
#pragma unroll
                for (int kk = 0; kk < UT; ++kk)
#pragma unroll
                    for (int st = 0; st < 4; ++st)
#pragma unroll
                        for (int i = 0; i <= kk; ++i)                                   // W is lower triangular
#pragma unroll
                            for (int j = 0; j < XT; ++j) Kt[i][j] = mfma4(nY[kk][j][st], Wt[kk][i][st], Kt[i][j]);      // K~' tiles: states x inputs
#pragma unroll
                for (int i = 0; i < UT; ++i)
#pragma unroll
                    for (int j = 0; j < XT; ++j) {
#ifdef WB_T_NOKSTORE     // timing build (tools/ab_wb.sh): no gain stores -- results are wrong, the time tells what the stores cost
                        if (a.B < 0)
#endif
                        if (k >= L.n_klds) store_tile(Kimg + (size_t)k * KT_FLOATS + (i * XT + j) * IMG, lane, Kt[i][j]);
                        else *reinterpret_cast<f32x4*>(klds + k * KBUF + ((i * XT + j) * 16 + c) * KLD + 4 * q4) = Kt[i][j];
                    }
            }
            WB_STAMP(18);      // K~ products and stores
        }
        WB_STAMP(8);
        phase_sync();
        // ------------------------------------------------------------ phase F: forward sweep
        // du = K~ dx~ row per lane, dx+ from the model's sparse structure -- with dx~ and the force part of du WAVE-UNIFORM in
        // scalar registers (v_readlane broadcasts) and the two lane-shifted terms of the kinematic rows through one crossbar
        // move: no LDS round trip on the stage-to-stage chain (the version that kept dx~, du in the LDS spent 4.4 k cycles
        // per stage in three dependent write -> fence -> read trips; this one is bit-identical to it).  Gain tiles are
        // requested three stages ahead (registers -> LDS -> rows), the stage record two stages ahead (registers -> LDS).
        float* oX = use_ipm ? dXp : dX;
        float* oU = use_ipm ? dUp : dU;
        // lanes are POSITIONS of x~ (pos_of): lane p holds dx of the state that sits there, lane HX the homogeneous 1, the others 0
        const int sl = state_at(lane);
        float xcur = 0.0f;
        if (sl >= 0) xcur = x0[sl >= 0 ? sl : 0] - Xg[sl >= 0 ? sl : 0];          // node 0 is not moved by the warm-start shift
        if (lane == HX) xcur = 1.0f;
        if (sl >= 0) AT(oX, 0, sl) = xcur;
        float* const rbuf[2] = {recb, hbuf + 1024};         // stage records, double-buffered (hbuf is free in this phase)
        // The gain images K~' (states x inputs) come in as TILES -- six 16 B loads per lane and stage, every one a contiguous KiB
        // across the wave -- two stages before the stage that writes them into the LDS (column stride 20 floats: conflict-free),
        // where every lane reads its row of K~ (a column of K~') a stage later.  A lane reading its row straight from memory
        // (twelve 16 B loads, 64 B apart between lanes: 64 separate lines per instruction) kept the CU's address path busy for
        // most of a stage with all four waves in this phase, and a deeper prefetch made that worse, not better.
        static_assert(KBUF <= 32 * LDU + 1024 && 1024 + 256 <= 48 * LDH, "gain tiles and the second record buffer fit the elimination's LDS");
        float* const kbuf = colU;                           // (spans colU and the head of hbuf: both free in this phase)
        const unsigned rec_lane = (4 * lane < REC) ? 4 * lane : 0;
        auto load_rec = [&](int k) { return *reinterpret_cast<const f32x4*>(recs + (size_t)(k < N ? k : N - 1) * REC + rec_lane); };
        *reinterpret_cast<f32x4*>(rbuf[0] + 4 * lane) = load_rec(0);
        f32x4 rec_ahead = load_rec(1);                      // the record of stage k + 1 while stage k runs
        wave_sync();
        const int urow = lane < NU ? lane : 0;
        auto load_ktiles = [&](int k, f32x4 (&t)[UT * XT]) {
            const float* Kk = Kimg + (size_t)(k < N ? k : N - 1) * KT_FLOATS;
#pragma unroll
            for (int i = 0; i < UT * XT; ++i) t[i] = load_tile(Kk + i * IMG, lane);
        };
        auto put_ktiles = [&](const f32x4 (&t)[UT * XT]) {
#pragma unroll
            for (int i = 0; i < UT * XT; ++i) *reinterpret_cast<f32x4*>(kbuf + (i * 16 + c) * KLD + 4 * q4) = t[i];
        };
        const int krow_off = ((urow >> 4) * XT * 16 + (urow & 15)) * KLD;      // tiles (urow >> 4, 0..2), column urow & 15
        auto get_krow = [&](const float* buf, float (&row)[48]) {
            const float* krow = buf + krow_off;
#pragma unroll
            for (int j4 = 0; j4 < 12; ++j4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(krow + (j4 >> 2) * 16 * KLD + 4 * (j4 & 3));
#pragma unroll
                for (int r = 0; r < 4; ++r) row[4 * j4 + r] = v[r];
            }
        };
        const int fi = sl >= 0 ? sl : 0;
        const int hr = (fi >= 39) ? fi - 39 : 0, lr = (fi >= 36 && fi < 39) ? fi - 36 : 0;
        const int x_addr = 4 * (lane < 18 ? lane + 18 : (lane < 36 ? lane - 18 : lane));      // q rows take dx of their v row, v rows du of their q row
        // tiles on their way: requested two stages before their LDS write.  ktA holds those of the even stages, ktB of the odd ones;
        // the first n_klds stages find their gains in the LDS where the backward sweep left them
        f32x4 ktA[UT * XT], ktB[UT * XT];
        const int KL = L.n_klds;                            // 0, 1 or 2
        if (KL == 0) {
            load_ktiles(0, ktA);
            load_ktiles(1, ktB);
            put_ktiles(ktA);
            load_ktiles(2, ktA);
        } else if (KL == 1) {
            load_ktiles(1, ktB);
            load_ktiles(2, ktA);
        } else {
            load_ktiles(2, ktA);
            load_ktiles(3, ktB);
        }
        wave_sync();
        // What the interior-point phase reads that this sweep does not write -- slacks, multipliers, constraint values, the input
        // gradient of the next sweep's barrier terms, lane = stage -- is requested three stages before the sweep ends: the phase then
        // opens with its inputs in registers instead of a trip to memory (7 k cycles per sweep).
        const bool ipm_more = ii + 1 < n_sweeps;
        const int ipm_k = lane < N ? lane : 0;
        float ipm_s[NG], ipm_l[NG], ipm_c[NG], ipm_rr[NU];
#pragma unroll
        for (int j = 0; j < NG; ++j) ipm_s[j] = ipm_l[j] = ipm_c[j] = 0.0f;
#pragma unroll
        for (int i = 0; i < NU; ++i) ipm_rr[i] = 0.0f;
        auto request_ipm_inputs = [&]() {
            const unsigned k4 = 4u * (unsigned)ipm_k;
            const float* rec = recs + (size_t)ipm_k * REC;
#pragma unroll
            for (int j = 0; j < NG; ++j) { ipm_c[j] = rec[R_C + j]; ipm_s[j] = ld_f32(sv + j * NS, k4, 0); ipm_l[j] = ld_f32(lv + j * NS, k4, 0); }
#pragma unroll
            for (int i = 0; i < NU; ++i) ipm_rr[i] = ipm_more ? rec[R_R + i] : 0.0f;
        };
        const int ipm_request_stage = N > 3 ? N - 3 : 0;
        auto fwd_stage = [&](int k, f32x4 (&kt)[UT * XT]) {      // kt: the tiles of stage k + 1, refilled with those of k + 3
            if (use_ipm && k == ipm_request_stage) request_ipm_inputs();
            const float* rk = rbuf[k & 1];
            const f32x4 rec_next = rec_ahead;                // stage k + 1's record, requested a stage ago
            rec_ahead = load_rec(k + 2);
            // what this lane needs of the stage record (not on the dependent chain: the record has been in the LDS for a stage)
            const float d_i = rk[R_D + (lane < 48 ? lane : 0)];      // (the record's defect is stored by position)
            f32x4 hq4[4], hf4[3];
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) hq4[v4] = *reinterpret_cast<const f32x4*>(rk + R_HQ + hr * 16 + 4 * v4);
#pragma unroll
            for (int v4 = 0; v4 < 3; ++v4) hf4[v4] = *reinterpret_cast<const f32x4*>(rk + R_HF + hr * 12 + 4 * v4);
            const f32x4 cd = *reinterpret_cast<const f32x4*>(rk + R_CDT);
            float row[48];
            get_krow(k < KL ? klds + k * KBUF : kbuf, row);
            float dxs[XW];
#pragma unroll
            for (int j = 0; j < XW; ++j) dxs[j] = bcast(xcur, j);
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
            for (int j4 = 0; j4 < 12; ++j4) {
                a0 = fmaf(row[4 * j4], dxs[4 * j4], a0);
                a1 = fmaf(row[4 * j4 + 1], dxs[4 * j4 + 1], a1);
                if (4 * j4 + 2 < XW) a2 = fmaf(row[4 * j4 + 2 < XW ? 4 * j4 + 2 : 0], dxs[4 * j4 + 2 < XW ? 4 * j4 + 2 : 0], a2);
                if (4 * j4 + 3 < XW) a3 = fmaf(row[4 * j4 + 3 < XW ? 4 * j4 + 3 : 0], dxs[4 * j4 + 3 < XW ? 4 * j4 + 3 : 0], a3);
            }
            const float du = (a0 + a1) + (a2 + a3);
            WB_STAMP(13);
            if (lane < NU) AT(oU, k, lane) = du;
            float duf[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) duf[j] = bcast(du, WF + j);
            const float other = __int_as_float(__builtin_amdgcn_ds_bpermute(x_addr, __float_as_int(lane >= 18 ? xcur : du)));
            float xn = xcur + d_i;
            // kinematic rows: q+ = q + dt v + dt^2 a, v+ = v + dt a
            const float t_q = dt * other + dt * dt * du;
            const float t_v = dt * other;
            xn += (fi < 18) ? t_q : (fi < 36) ? t_v : 0.0f;
            // momentum rows 36..41 (every lane runs the code on a clamped row; six lanes keep the result):
            // linear rows: sum_f cdt_f duf[3f + i-36]; angular rows: Hq[i-39] . dx[3..17] + Hf[i-39] . duf
            float acc_a = 0.0f, acc_l = 0.0f;
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * v4 + r < 15) acc_a = fmaf(hq4[v4][r], dxs[3 + 4 * v4 + r], acc_a);
#pragma unroll
            for (int v4 = 0; v4 < 3; ++v4)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_a = fmaf(hf4[v4][r], duf[4 * v4 + r], acc_a);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const float sel = lr == 0 ? duf[3 * f] : lr == 1 ? duf[3 * f + 1] : duf[3 * f + 2];
                acc_l = fmaf(cd[f], sel, acc_l);
            }
            xn += (fi >= 39) ? acc_a : (fi >= 36) ? acc_l : 0.0f;
            WB_STAMP(14);
            if (sl >= 0) AT(oX, k + 1, sl) = xn;
            xcur = (sl >= 0) ? xn : (lane == HX ? 1.0f : 0.0f);
            *reinterpret_cast<f32x4*>(rbuf[(k + 1) & 1] + 4 * lane) = rec_next;
            if (k + 1 >= KL) {
                put_ktiles(kt);                              // (this stage's row reads were issued before: the LDS serves a wave in order)
                load_ktiles(k + 3, kt);
            }
            wave_sync();
            WB_STAMP(15);
        };
        {
            int k = 0;
            for (; k + 2 <= N; k += 2) {
                fwd_stage(k, ktB);
                fwd_stage(k + 1, ktA);
            }
            if (k < N) fwd_stage(k, ktB);
        }
        phase_sync();
        WB_STAMP(9);
        // ------------------------------------------------------------ phase I: interior-point update, lane = stage
        // Everything the phase reads from memory is requested in one go at its top -- its own inputs, the first pieces of the
        // step blend, the input gradient of the next sweep's barrier terms -- and nothing in it waits for a store: written
        // serially (load, compute, store, fence, load ...) the phase took six memory latencies per sweep, 15 % of the kernel
        // (tools/wb_stamps.py, slot 20).  Addresses are uniform base + lane offset (one VGPR for all of them).
        if (use_ipm) {
            const bool more = ipm_more;
            const bool live = lane < N;
            const int k = ipm_k;
            const unsigned k4 = 4u * (unsigned)k;
            float duf[12], g[NG];
            float (&s)[NG] = ipm_s; float (&l)[NG] = ipm_l; float (&cc)[NG] = ipm_c; float (&rr)[NU] = ipm_rr;
#pragma unroll
            for (int i = 0; i < 12; ++i) duf[i] = ld_f32(dUp + (WF + i) * NS, k4, 0);
            // step <- step + ap (new step - step): [dX | dU] and [dXp | dUp] have the same layout (StageArr), one pass over both
            // in 16 B pieces, nine pieces per lane in flight
            const int nvec = (SA.dXp - SA.dX) >> 2;
            f32x4* d4 = reinterpret_cast<f32x4*>(dX);
            const f32x4* n4 = reinterpret_cast<const f32x4*>(dXp);
            constexpr int BL = 9;      // N <= 30: the whole blend in one round (559 pieces)
            f32x4 dv[BL], nv[BL];
            auto blend_request = [&](int i0) {
#pragma unroll
                for (int u = 0; u < BL; ++u) {
                    const int i = i0 + 64 * u;
                    const int ic = i < nvec ? i : i0;
                    nv[u] = n4[ic];
                    dv[u] = (ii == 0) ? zero4() : d4[ic];
                }
            };
            blend_request(lane);
            request_sweep();
            gdot(mp, duf, g);
            // (the steps ds, dl are formed twice -- for the step lengths and for the update -- rather than kept: 32 registers)
            float rp = 0.0f, rd = 0.0f;
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const float is = fast_rcp(s[j]);
                const float dsj = -(g[j] + cc[j]) - s[j];
                const float dlj = tau * is - l[j] - l[j] * is * dsj;
                const bool on = live && ((my_act >> j) & 1u);
                rp = on ? fmaxf(rp, -dsj * is) : rp;
                rd = on ? fmaxf(rd, -dlj * __builtin_amdgcn_rcpf(l[j])) : rd;
            }
            WB_STAMP(22);      // inputs arrived, step lengths per lane
            const float rpm = wave_max(rp), rdm = wave_max(rd);
            const float ap = rpm > a.gamma ? a.gamma / rpm : 1.0f;
            const float ad = rdm > a.gamma ? a.gamma / rdm : 1.0f;
            float m_l = 0.0f;
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const float is = fast_rcp(s[j]);
                const float dsj = -(g[j] + cc[j]) - s[j];
                const float dlj = tau * is - l[j] - l[j] * is * dsj;
                const bool on = live && ((my_act >> j) & 1u);
                s[j] += on ? ap * dsj : 0.0f;
                l[j] += on ? ad * dlj : 0.0f;
                m_l += on ? s[j] * l[j] : 0.0f;
            }
            if (live) {
#pragma unroll
                for (int j = 0; j < NG; ++j) { st_f32(sv + j * NS, k4, 0, s[j]); st_f32(lv + j * NS, k4, 0, l[j]); }
            }
            mu_sum = wave_sum(m_l);
            WB_STAMP(23);      // reductions, update, slack / multiplier stores
            for (int i0 = lane; i0 < nvec; i0 += 64 * BL) {
#pragma unroll
                for (int u = 0; u < BL; ++u) {
                    const int i = i0 + 64 * u;
                    if (i < nvec) d4[i] = dv[u] + ap * (nv[u] - dv[u]);
                }
                if (i0 + 64 * BL < nvec) blend_request(i0 + 64 * BL);
            }
            // (no fence here: the next reader of dX, dU across lanes is the step phase, behind its own phase_sync; the slacks and
            //  multipliers are read back by the lane that wrote them)
            if (more) {
                ipm_terms(fmaxf(a.sigma * mu_sum / (float)n_act, a.tau_min), s, l, cc, rr);
                wave_sync();
            }
        } else {
            request_sweep();
        }
        WB_STAMP(20);      // interior-point update of this sweep
    }
    WB_STAMP(10);
    phase_sync();
    // ---------------------------------------------------------------- phase S: step
    float sn_l = 0.0f;
    bool bad_l = false;
    for (int k = lane; k <= N; k += 64) {
        for (int i = 0; i < NX; ++i) { const float v = AT(dX, k, i); bad_l = bad_l || !(fabsf(v) <= 1e30f); sn_l = fmaxf(sn_l, fabsf(v)); }
        if (k < N)
            for (int i = 0; i < NU; ++i) { const float v = AT(dU, k, i); bad_l = bad_l || !(fabsf(v) <= 1e30f); sn_l = fmaxf(sn_l, fabsf(v)); }
    }
    const float stepn = wave_max(sn_l);
    const bool bad = __any(bad_l);
    int status = NMPC_STATUS_MAXITER;
    bool finished = false;
    if (bad) { status = NMPC_STATUS_NAN; finished = true; }
    else if (!qp_ok) { status = NMPC_STATUS_QP; finished = true; }
    else if (a.nlp_tol > 0.0f && stepn < a.nlp_tol) { status = NMPC_STATUS_OK; finished = true; }
    // new iterate = previous one (read through the warm-start shift) + step; all reads before the first write
    if (!bad || a.shift > 0) {
        constexpr int PRE = 8;
        const float sc = bad ? 0.0f : 1.0f;
        const int n_x = (N + 1) * NX, n_u = N * NU;
        for (int base = 0; base < n_x; base += 64 * PRE) {
            float v[PRE];
#pragma unroll
            for (int uu = 0; uu < PRE; ++uu) {
                const int e = base + 64 * uu + lane, ec = e < n_x ? e : 0;
                const int k = ec / NX, i = ec - k * NX;
                v[uu] = Xg[ec + (shifted_node(k, a.shift, N) - k) * NX] + (bad ? 0.0f : sc * AT(dX, k, i));
            }
            if (a.shift > 0) phase_sync();      // a shifted read may be another lane's write target
#pragma unroll
            for (int uu = 0; uu < PRE; ++uu) {
                const int e = base + 64 * uu + lane;
                if (e < n_x) Xg[e] = v[uu];
            }
            if (a.shift > 0) phase_sync();
        }
        for (int base = 0; base < n_u; base += 64 * PRE) {
            float v[PRE];
#pragma unroll
            for (int uu = 0; uu < PRE; ++uu) {
                const int e = base + 64 * uu + lane, ec = e < n_u ? e : 0;
                const int k = ec / NU, i = ec - k * NU;
                const bool okk = a.shift == 0 || shifted_stage_valid(k, a.shift, N);
                const float t = Ug[ec + (okk ? a.shift * NU : 0)];
                v[uu] = ((okk || i < WF) ? t : 0.0f) + (bad ? 0.0f : sc * AT(dU, k, i));
            }
            if (a.shift > 0) phase_sync();
#pragma unroll
            for (int uu = 0; uu < PRE; ++uu) {
                const int e = base + 64 * uu + lane;
                if (e < n_u) Ug[e] = v[uu];
            }
            if (a.shift > 0) phase_sync();
        }
    }
#ifdef NMPC_WB_STAMPS
    WB_STAMP(11);
    if (lane == 0) for (int i = 0; i < 24; ++i) (ws + wl.js)[i] = (float)st_acc[i];
#endif
    if (lane == 0) {
        flag[0] = finished ? 1 : 0;
        if (a.status) a.status[b] = status;
        if (a.stats) {
            a.stats[4 * b + 0] = cost;
            a.stats[4 * b + 1] = stepn;
            a.stats[4 * b + 2] = 1.0f;
            a.stats[4 * b + 3] = (float)(a.it + 1);
        }
    }
#undef AT
}

}  // namespace wb
}  // namespace nmpc
