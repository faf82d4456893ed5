// nmpc_solve.hip -- batched NMPC solve on gfx950: two kernels per SQP iteration (one in the two-wave variant, which
// linearises inside the QP kernel: qp_linearizes_itself).
//
// Replaces the reference's per-step solve (mpc_controller/utils/solver.py:396-403 ->
// acados SQP / HPIPM):
//   nmpc_linearize_kernel  one thread per (problem, stage): dynamics, analytic A,B, defects,
//                          gradients, constraint values -> workspace (fully parallel, any
//                          register budget, no cross-lane traffic)
//   nmpc_qp_kernel         one problem per wavefront: interior-point loop of Riccati sweeps
//        phase R (serial stages)  backward sweep on 16x16 tiles: MFMA products + readlane LDL'
//        phase F (serial stages)  rollout of the closed loop (row-per-lane VALU mat-vec)
//        phase I (lane = stage)   interior-point step: slacks, multipliers, fraction to boundary
//        phase S (lane = stage)   step (optionally l1-merit backtracking), status, write back
// Splitting the linearisation off keeps the sweep kernel's register allocation small (the
// Jacobian code wants hundreds of VGPRs; fused, it forced the MFMA accumulators of the sweeps
// through AGPR<->VGPR copies and pinned the kernel at one wave per SIMD).
// Stage matrices A~, B~, K~, Acl~ live in an HBM/L2 workspace as compact images (640-768 B);
// trajectories, gradients and IPM state of a problem live in LDS (39.6 KB at N = 50) inside
// nmpc_qp_kernel -- or, in its lean variant for batches beyond one wave per SIMD, partly in the workspace (Lds).
// Kernel variants (template flags): LDS layout, precision of the barrier product, set of contact
// patterns with a static stage body.  DESIGN.md 5 has the measurements behind each choice.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/nmpc.h"
#include "nmpc_models.hpp"
#include "nmpc_sweep.hpp"

namespace nmpc {

struct SolveArgs {
    ModelParams mp;
    float W[32];    // stage weights [nx+nu]
    float rs_free[16];   // 1/sqrt(W_u + reg): row scale of an input that is decoupled at a stage
    float We[16];   // terminal weights [nx]
    float reg, reg_e;
    int N, B;
    int max_sqp, n_ipm, line_search, yref_per_stage;
    int it;               // SQP iteration this launch pair belongs to
    int shift;            // warm-start shift folded into this launch pair's reads of X, U (0: none)
    float nlp_tol, mu0, sigma, s_min, gamma, tau_min, rho;
    const float* x0;
    const float* yref;
    const float* yref_e;
    const float* params;
    float* X;
    float* U;
    int* status;
    float* stats;
    float* ws;            // workspace, WsLayout per problem
    float* dbg;           // diagnostic builds: [B][8] phase cycle counts, else unused
    const int* skip;      // nullptr, or dev [B] flag words: a problem with skip[b] & skip_mask != 0 is left untouched
    int skip_mask;        // (terminated rollouts, nmpc_set_skip)
};

__device__ __forceinline__ bool skipped(const SolveArgs& a, int b) { return a.skip && (a.skip[b] & a.skip_mask) != 0; }

__host__ __device__ inline int round4(int n) { return (n + 3) & ~3; }

// Warm-start shift as an index map (solver.py:304-322, same result as nmpc_shift_kernel): node 0 and
// the last `shift` nodes of X keep their values, nodes 1..N-shift take those `shift` nodes later; U
// moves up by `shift` stages and its tail is zero.  With it the shift costs no launch and no pass
// over memory: the kernels of the first SQP iteration read the previous solution through the map.
__device__ __forceinline__ int shifted_node(int k, int shift, int N) {
    return (k >= 1 && k <= N - shift) ? k + shift : k;
}
__device__ __forceinline__ bool shifted_stage_valid(int k, int shift, int N) { return k < N - shift; }

// Compact stage images in the workspace (tile strides are multiples of 128 B so an image starts
// on a cache line).  Images hold logical indices; the slot layout of the tiles (nmpc_tile.hpp)
// exists only in registers and in the column index of the K~/Acl~ rows:
//   A~ : columns 0..nx (A | d), rows 0..nx-1, column-major, column stride SA = round4(nx) floats
//        (row HS = e_HS is synthesised at load)
//   B~ : columns 0..nu-1, rows 0..nx-1, same column stride
//   K~ : rows 0..nu-1 of [K kff], row-major, 16 floats per row indexed by slot (what the forward
//        sweep multiplies with); rows padded to a multiple of 3
//   Acl~: rows 0..nx-1 of A~ + B~K~, same format
template <class M>
struct TileGeom {
    static_assert(M::NX <= 12 && M::NU <= 12, "slot layout: at most 12 states and 12 inputs");
    static constexpr int SA = (M::NX + 3) & ~3;                       // column stride of A~/B~ images
    static constexpr int RQ = SA / 4;                                 // row quads stored per column
    static constexpr int KQ = (M::NU + 2) / 3;                        // row triples of a K~ image
    static constexpr int CQ = (M::NX + 2) / 3;                        // row triples of an Acl~ image
    static constexpr int A_FLOATS = (((M::NX + 1) * SA) + 31) & ~31;
    static constexpr int B_FLOATS = ((M::NU * SA) + 31) & ~31;
    static constexpr int K_FLOATS = ((3 * KQ * TS) + 31) & ~31;
    static constexpr int C_FLOATS = ((3 * CQ * TS) + 31) & ~31;
};

typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));

// Per-lane byte offsets into a stage image, computed once per kernel.  Every access of the stage
// sweeps is  uniform image base + lane offset + immediate  (ld_f32/st_f32): no per-access address
// arithmetic in the VALU.  Loads come in two halves so that a prefetch stays a prefetch: the raw
// load is issued a stage ahead (clamped in-bounds addresses), *_fix masks the padding when the tile
// is consumed.  (Masking at the load would make the wave wait for it on the spot.)
// Lane (q, c) of a tile holds row slots 4q..4q+3 = logical rows 3q..3q+2 (+ padding) of column slot c.
template <class M>
struct ImageLane {
    using G = TileGeom<M>;
    unsigned a_off, b_off, t_off;   // A~ and B~ (three consecutive rows of a column), B~' (3 dwords)
    bool a_ok, b_ok, t_ok;
    __device__ __forceinline__ void init(int lane) {
        const int q = lane >> 4, c = lane & 15;
        const int jc = (c == HS) ? M::NX : index_of(c);       // logical column of A~ (d sits at HS)
        a_ok = (jc >= 0) && (jc <= M::NX) && (3 * q < M::NX);
        b_ok = (index_of(c) >= 0) && (index_of(c) < M::NU) && (3 * q < M::NX);
        a_off = 4u * ((a_ok ? jc : 0) * G::SA + (a_ok ? 3 * q : 0));
        b_off = 4u * ((b_ok ? index_of(c) : 0) * G::SA + (b_ok ? 3 * q : 0));
        // element (slot(u), slot(x)) of B~' is B~[x][u]: image column u = 3q+r, row x
        t_ok = (index_of(c) >= 0) && (index_of(c) < M::NX) && (3 * q < M::NU);
        t_off = 4u * ((t_ok ? 3 * q : 0) * G::SA + (t_ok ? index_of(c) : 0));
    }
    __device__ __forceinline__ f32x4 load3(const float* img, unsigned off) const {
        const f32x3u v = *reinterpret_cast<const f32x3u*>(reinterpret_cast<const char*>(img) + (size_t)off);
        return f32x4{v[0], v[1], v[2], 0.0f};
    }
    __device__ __forceinline__ f32x4 load_A(const float* img) const { return load3(img, a_off); }
    __device__ __forceinline__ f32x4 load_B(const float* img) const { return load3(img, b_off); }
    __device__ __forceinline__ f32x4 load_Bt(const float* img) const {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = ld_f32(img, t_off, 4 * r * G::SA);
        o[3] = 0.0f;
        return o;
    }
    // masks of the raw loads: rows/columns beyond the model's dimensions are zero; A~[HS][HS] = 1
    __device__ __forceinline__ f32x4 fix_A(f32x4 v, int lane) const {
        const int q = lane >> 4, c = lane & 15;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = (a_ok && 3 * q + r < M::NX) ? v[r] : 0.0f;
        o[3] = (q == 0 && c == HS) ? 1.0f : 0.0f;
        return o;
    }
    __device__ __forceinline__ f32x4 fix_B(f32x4 v, int lane) const {
        const int q = lane >> 4;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = (b_ok && 3 * q + r < M::NX) ? v[r] : 0.0f;
        o[3] = 0.0f;
        return o;
    }
    __device__ __forceinline__ f32x4 fix_Bt(f32x4 v, int lane) const {
        const int q = lane >> 4;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 3; ++r) o[r] = (t_ok && 3 * q + r < M::NU) ? v[r] : 0.0f;
        o[3] = 0.0f;
        return o;
    }
};
// Row-major store of the logical rows of an accumulator-layout tile (registers 0..2 of the first NQ
// quads), 16 floats per row, into the image of stage k of an image array.  Lanes of the other quads
// write to the array's scratch image (slot N) instead, so the store needs no branch.  `off` walks
// down with the stage.
struct RowStoreLane {
    unsigned off, step;
    __device__ __forceinline__ void init(int lane, int nq, int image_floats, int k_first, int k_scratch) {
        const int q = lane >> 4, c = lane & 15;
        const bool valid = q < nq;
        off = 4u * ((valid ? k_first : k_scratch) * image_floats + (valid ? 3 * q * TS : 0) + c);
        step = valid ? 4u * image_floats : 0u;
    }
    __device__ __forceinline__ void store(float* images, f32x4 v) {
#pragma unroll
        for (int r = 0; r < 3; ++r) st_f32(images, off, 4 * r * TS, v[r]);
        off -= step;
    }
};

// Arrays of the lane = stage phases (trajectories, steps, slacks, multipliers).  Two layouts:
//   LDS (resident variant): FEATURE-major, [feature][stage] with an odd stage stride NS, so consecutive lanes touch
//       consecutive banks;
//   workspace (lean variant): STAGE-major, [stage][16]: the lane that owns a stage moves its whole row with three or
//       four 16 B accesses (a feature-major row costs one 4 B access per feature: 100 instead of 35 memory instructions
//       per lane and interior-point iteration), and the forward sweep's row stores of a stage land in one cache line.
// Offsets in floats from the start of the group; nX/nU/nG = extent of a state / input / constraint-row array for the
// element-wise passes (padding of the stage-major rows is zero and stays zero).
template <class M>
struct StageArrays {
    int NS;   // feature-major: stage stride (odd, >= N+1)
    int nX, nU, nG;
    int Xs, Us, dX, dU, dXp, dUp, sv, lv, total;
    __host__ __device__ StageArrays(int N, bool stage_major) {
        NS = (N + 1) | 1;
        nX = stage_major ? (N + 1) * TS : M::NX * NS;
        nU = stage_major ? (N + 1) * TS : M::NU * NS;
        nG = stage_major ? (N + 1) * TS : M::NG * NS;
        int o = 0;
        Xs = o;  o += round4(nX);
        Us = o;  o += round4(nU);
        dX = o;  o += round4(nX);
        dU = o;  o += round4(nU);
        dXp = o; o += round4(nX);
        dUp = o; o += round4(nU);
        sv = o;  o += round4(nG);
        lv = o;  o += round4(nG);
        o += 4;                                   // sink for the idle lanes of the forward sweep
        total = o;
    }
};

// Workspace of one problem (float offsets): stage images A~[N], B~[N], K~[N+1], Acl~[N+1]
// (the extra slot is scratch for the software pipeline), then what the linearisation hands to the
// QP kernel: gradients q[N+1][nx], r[N][nu], constraint values c[N][ng], masks, stage costs, and
// the problem's "finished" flag.
template <class M>
struct WsLayout {
    size_t At, Bt, Kt, Ct, q, r, c, act, umk, cost, flag, lean, stride;
    __host__ __device__ explicit WsLayout(int N) {
        size_t o = 0;
        At = o; o += (size_t)N * TileGeom<M>::A_FLOATS;
        Bt = o; o += (size_t)N * TileGeom<M>::B_FLOATS;
        Kt = o; o += (size_t)(N + 1) * TileGeom<M>::K_FLOATS;
        Ct = o; o += (size_t)(N + 1) * TileGeom<M>::C_FLOATS;
        q = o; o += round4((N + 1) * M::NX);
        r = o; o += round4(N * M::NU);
        c = o; o += round4(N * M::NG);
        act = o; o += round4(N);
        umk = o; o += round4(N);
        cost = o; o += round4(N + 1);
        flag = o; o += 4;
        lean = o; o += StageArrays<M>(N, true).total;     // lane = stage arrays of the lean-LDS kernel variant (stage-major)
        stride = (o + 63) & ~(size_t)63;
    }
};
constexpr int N_LANE_STAGES = 4;   // horizon limit of the lane = stage loops: N <= 256
// true if nmpc_qp_kernel<M, LEAN, ..> linearises its problem itself (then nmpc_linearize_kernel is not launched)
__host__ __device__ constexpr bool qp_linearizes_itself(bool lean, int N) { return lean && N < 64; }

// Diagnostic build only (-DNMPC_STAMPS, tools/phase_shares.py): per-phase cycle counters written to
// a buffer of their own; the production kernel contains no stamp.
#ifdef NMPC_STAMPS
#define STAMP_DECL unsigned long long st_t0 = __builtin_readcyclecounter(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
    StageStamps sst = {0, {0, 0, 0, 0, 0, 0, 0, 0}}
#define SST_PASS , sst
#define SST_TILES(i) SST(i)
#define STAMP(i) do { const unsigned long long st_t1 = __builtin_readcyclecounter(); st_acc[i] += st_t1 - st_t0; st_t0 = st_t1; } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define SST_PASS
#define SST_TILES(i)
#endif

// LDS layout of one problem.  Two variants of the QP kernel:
//   resident (LEAN = false): stage arrays + sweep operands + conversion tiles, 39.6 KB at N = 50 ->
//       4 waves per CU, one per SIMD: the choice while the batch fits that many waves (B <= 1024 on an MI355X);
//   lean (LEAN = true): the stage arrays live in the workspace (stage-major rows), 17.8 KB -> 8 waves per CU, two per
//       SIMD: the choice for larger batches (the second wave fills the first one's dependency stalls: 2.7 M solves/s
//       at B = 8192 against 2.3 M), for horizons whose resident layout does not fit the LDS, and NMPC_QP_VARIANT=lean.
// Same arithmetic in the same order: results are bit-identical (tests/test_gpu_parity.py).  DESIGN.md 7.
template <class M, bool LEAN>
struct Lds {
    int arr, qv, rv, gsq, gvt, act, umk, conv, dx0, total;   // float offsets
    __host__ __device__ explicit Lds(int N) {
        const int NS = (N + 1) | 1;
        int o = 0;
        arr = o; o += LEAN ? 0 : StageArrays<M>(N, false).total;
        // sweep operands are STAGE-major, 16 floats per stage: the four tile registers of a lane
        // are one 16 B read (only the stage sweeps touch them after they are written)
        qv = o;  o += (N + 1) * TS;
        rv = o;  o += N * TS;
        gsq = o; o += N * TS;
        gvt = o; o += N * TS;
        act = o; o += round4(NS);
        umk = o; o += round4(NS);
        conv = o; o += CONV_FLOATS;
        dx0 = o; o += LEAN ? TS : 0;              // lean: x0 - X[0], read at the start of every forward sweep
        total = o;
    }
};

// ------------------------------------------------------------------------------------------------
// Linearisation: thread t <-> (problem b, stage k), k = N is the terminal stage.
// A thread owns a whole 1 KiB tile, so storing it directly would make every wave store touch 64
// different tiles with 16 B each.  Columns therefore go through a 4 KiB LDS stage: the block
// re-distributes one column of 64 stages so that the lanes of a quad group write the contiguous
// bytes of a stage's column (16 tiles per store instruction instead of 64).  The block is ONE wave:
// the LDS hand-over is ordered by issue (wave_sync), not by s_barrier -- __syncthreads would also drain
// the column stores of the previous flush (vmcnt(0)) 25 times per thread (measured: 29.8 -> 24.8 us
// together with requesting the reference row and the next node up front).
// (a device function: the kernel below runs it with thread t <-> (problem, node); the two-wave variant of the QP kernel runs
//  it itself, lane = node, in front of its prologue when the horizon fits one wave -- qp_linearizes_itself)
constexpr int LIN_LDS_FLOATS = 64 * 16 + 2 * 64 * 2;      // column stage + the 2 x 64 tile pointers
template <class M>
__device__ __forceinline__ void linearize_node(const SolveArgs& a, bool in_range, int b_in, int k_in, float* lds) {
#pragma clang fp contract(off)      // as the model functions it calls (nmpc_models.hpp): one rounding behaviour in every caller
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NG = M::NG, NY = NX + NU;
    using G = TileGeom<M>;
    float* stage_col = lds;
    float** tile_of0 = reinterpret_cast<float**>(lds + 64 * 16);
    float** tile_of1 = tile_of0 + 64;
    float** tile_of[2] = {tile_of0, tile_of1};       // A~ / B~ tile of each thread's stage (nullptr: none)
    const int N = a.N;
    const int tid = threadIdx.x;
    const int b = b_in;                   // a valid problem index also where in_range is false (idle lanes only read it)
    const int k = in_range ? k_in : N;
    const WsLayout<M> wl(N);
    float* ws = a.ws + (size_t)b * wl.stride;
    const bool live = in_range && !(a.it > 0 && reinterpret_cast<const int*>(ws + wl.flag)[0]) && !skipped(a, b);
    const bool stage = live && k < N;       // this thread linearises a shooting interval
    const int ks = stage ? k : 0;
    const float* Xg = a.X + (size_t)b * (N + 1) * NX;
    const float* Ug = a.U + (size_t)b * N * NU;
    tile_of[0][tid] = stage ? ws + wl.At + (size_t)k * G::A_FLOATS : nullptr;
    tile_of[1][tid] = stage ? ws + wl.Bt + (size_t)k * G::B_FLOATS : nullptr;
    float x[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = Xg[(size_t)shifted_node(live ? k : 0, a.shift, N) * NX + i];
    if (live && k == N) {   // terminal gradient and cost
        const float* yre = a.yref_e + (size_t)b * NX;
        float cst = 0.0f;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const float e = x[i] - yre[i];
            ws[wl.q + (size_t)N * NX + i] = a.We[i] * e;
            cst += 0.5f * a.We[i] * e * e;
        }
        ws[wl.cost + N] = cst;
    }
    float u[NU], xn[NX], p[NP > 0 ? NP : 1];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const float v = Ug[(size_t)(shifted_stage_valid(ks, a.shift, N) ? ks + a.shift : ks) * NU + i];
        u[i] = (a.shift == 0 || shifted_stage_valid(ks, a.shift, N)) ? v : 0.0f;
    }
    const float* pg = a.params + ((size_t)b * (N + 1) + ks) * NP;
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = pg[i];
    // everything else the thread will need from memory, requested now: the loads overlap the Jacobians
    // instead of costing a round trip each where they are used
    float x_next[NX], yv[NY];
#pragma unroll
    for (int i = 0; i < NX; ++i) x_next[i] = Xg[(size_t)shifted_node(ks + 1, a.shift, N) * NX + i];
    const float* yk = a.yref + (size_t)b * (a.yref_per_stage ? (size_t)N * NY : (size_t)NY) +
                      (a.yref_per_stage ? (size_t)ks * NY : 0);
#pragma unroll
    for (int i = 0; i < NY; ++i) yv[i] = yk[i];
    wave_sync();
    // every thread of the block takes part in every column flush (idle threads carry a null tile)
    auto flush = [&](int which, int j, const float (&v)[16]) {
#ifdef LIN_T_NOFLUSH   // timing build (tools/lin_timing.sh): no hand-over, the column only stays alive
        float sacc = 0.0f;
        for (int i = 0; i < 16; ++i) sacc += v[i];
        if (sacc == 123.456f) stage_col[tid] = sacc;
        return;
#endif
        f32x4* mine = reinterpret_cast<f32x4*>(stage_col + tid * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) mine[i] = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
        wave_sync();
        const int quad = tid & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = (tid >> 2) + 16 * i;
            float* tile = tile_of[which][s];
#ifdef LIN_T_NOSTORE   // timing build: hand-over without the column stores
            if (tile != nullptr && quad < G::RQ && a.B < 0)
#else
            if (tile != nullptr && quad < G::RQ)
#endif
                *reinterpret_cast<f32x4*>(tile + j * G::SA + 4 * quad) =
                    *reinterpret_cast<const f32x4*>(stage_col + s * 16 + 4 * quad);
        }
        wave_sync();
    };
    auto emitA = [&](int j, const float (&colv)[NX]) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (i < NX) ? colv[i < NX ? i : 0] : 0.0f;
        flush(0, j, v);
    };
    auto emitB = [&](int j, const float (&colv)[NX]) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (i < NX) ? colv[i < NX ? i : 0] : 0.0f;
        flush(1, j, v);
    };
    M::linearize(a.mp, x, u, p, xn, emitA, emitB);
    {   // defect column  [d; 1]
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            v[i] = (i < NX) ? xn[i < NX ? i : 0] - x_next[i < NX ? i : 0] : 0.0f;
        flush(0, NX, v);
    }
#ifdef LIN_T_NOTAIL    // timing build: no gradients / constraint values / masks
    if (a.B > 0) return;
#endif
    if (!stage) return;
    float cst = 0.0f;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const float e = x[i] - yv[i];
        ws[wl.q + (size_t)k * NX + i] = a.W[i] * e;
        cst += 0.5f * a.W[i] * e * e;
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const float e = u[i] - yv[NX + i];
        ws[wl.r + (size_t)k * NU + i] = a.W[NX + i] * e;
        cst += 0.5f * a.W[NX + i] * e * e;
    }
    ws[wl.cost + k] = cst;
    reinterpret_cast<unsigned*>(ws + wl.umk)[k] = M::input_mask(a.mp, p);
    reinterpret_cast<unsigned*>(ws + wl.act)[k] = (a.n_ipm > 0) ? M::active_mask(a.mp, p) : 0u;
    float g[NG];
    M::gdot(a.mp, u, g);
#pragma unroll
    for (int j = 0; j < NG; ++j) ws[wl.c + (size_t)k * NG + j] = g[j] - M::h(a.mp, j);
}

template <class M>
__global__ __launch_bounds__(64) void nmpc_linearize_kernel(const SolveArgs a) {
    __shared__ __attribute__((aligned(16))) float lin_lds[LIN_LDS_FLOATS];
    const int N = a.N;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = t < (long long)a.B * (N + 1);
    const int b = in_range ? (int)(t / (N + 1)) : 0;
    const int k = in_range ? (int)(t - (long long)b * (N + 1)) : N;
    linearize_node<M>(a, in_range, b, k, lin_lds);
}

// Element-wise pass over n floats, lane-strided, with PRE loads per lane in flight before the first
// result is stored: ld(i) computes the value of element i from its loads, st(i, v) consumes it.
// (In the lean variant these are memory round trips; a plain loop would serialise them, since the
// compiler must assume that a store may alias the next iteration's loads.)
template <int PRE, class Load, class Store>
__device__ __forceinline__ void batched(int n, int lane, Load&& ld, Store&& st) {
    for (int base = 0; base < n; base += 64 * PRE) {
        float v[PRE];
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int i = base + 64 * u + lane;
            v[u] = ld(i < n ? i : 0);
        }
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int i = base + 64 * u + lane;
            if (i < n) st(i, v[u]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// QP + step of one SQP iteration: one problem per wavefront.
// LEAN: LDS layout variant (Lds).  BF16B: the barrier product Gs'[Gs | vt] on the bf16 matrix pipe
// (nmpc_dims.precision = 1, BASELINE configs[4]); everything else stays fp32.
// ALLV: the stage loop dispatches over every static variant the model lists (all contact patterns);
// otherwise over the model's short list (common_variant) with the run-time fallback for the rest.
// Two kernels rather than one loop with everything: measured, the mere presence of the other
// variants in the kernel costs the common ones 2.5 % (code layout, register allocation).
// Registers: 234 VGPRs (resident) / 256 with 16 values in scratch, none of them in a stage loop (lean); no AGPRs
// (profiles/r02_isa_resources.md).  The resident variant is pinned to ONE wave per SIMD (amdgpu_waves_per_eu): its LDS image lets
// four waves onto a CU, and with two allowed per SIMD the hardware co-locates them and leaves SIMDs idle once a batch runs in
// rounds (measured 1.91 M instead of 2.30 M solves/s at B = 8192).  build.sh passes -amdgpu-mfma-vgpr-form so that the MFMAs keep
// their VGPR operands.  What kept both variants above 256 registers until round 2 was the interior-point phase, not the sweeps:
// its loads were interleaved with their uses; a stage's rows are now requested up front (ld_row / st_row).
// Floating-point contraction is OFF in the body of this kernel (the tile algebra in nmpc_sweep.hpp / nmpc_tile.hpp is MFMA and
// explicit fma; this is about the lane = stage arithmetic) and every a*b + c that is meant as one operation is written as fmaf:
// left to the compiler, which products get fused into the following add depends on the instantiation -- the resident and the
// lean variant, compiled from the same expressions, came out 1e-5 apart after an unrelated change to their loads.
#pragma clang fp contract(off)
#ifndef NMPC_RES_WAVES
#define NMPC_RES_WAVES 1      // diagnostic builds (tools/occupancy_probe.py) compile the resident variant for two waves per SIMD
#endif
template <class M, bool LEAN, bool BF16B, bool ALLV>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LEAN ? 2 : NMPC_RES_WAVES, LEAN ? 2 : NMPC_RES_WAVES)))
void nmpc_qp_kernel(const SolveArgs a) {
    constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NG = M::NG, NY = NX + NU;
    using G = TileGeom<M>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b >= a.B) return;
    if (skipped(a, b)) return;                   // e.g. a terminated rollout: X, U, status stay as they are
    const int lane = lane_id();
    const int q4 = lane >> 4, c = lane & 15;
    const int N = a.N;
    const WsLayout<M> wl(N);
    float* ws = a.ws + (size_t)b * wl.stride;
    int* flag = reinterpret_cast<int*>(ws + wl.flag);
    if (a.it > 0 && flag[0]) return;             // finished in an earlier iteration
    const Lds<M, LEAN> L(N);
    const StageArrays<M> SA_(N, LEAN);
    const int NS = SA_.NS;
    // the stage arrays: LDS in the resident variant, workspace in the lean one (same code either way;
    // the variant is a template parameter so that every pointer has ONE address space)
    float* arr = LEAN ? ws + wl.lean : smem + L.arr;
    float* Xs = arr + SA_.Xs;   float* Us = arr + SA_.Us;
    float* dX = arr + SA_.dX;   float* dU = arr + SA_.dU;
    float* dXp = arr + SA_.dXp; float* dUp = arr + SA_.dUp;
    float* sv = arr + SA_.sv;   float* lv = arr + SA_.lv;
    float* idle_sink = arr + SA_.total - 4;
    float* qv = smem + L.qv;   float* rv = smem + L.rv;
    float* gsq = smem + L.gsq; float* gvt = smem + L.gvt;
    // ordering point between phases that exchange data between lanes: LDS traffic of a single wave
    // is ordered by issue; the workspace needs the stores drained first
    auto phase_sync = [&]() {
        if constexpr (LEAN) __threadfence_block();
        wave_sync();
    };
    unsigned* actm = reinterpret_cast<unsigned*>(smem + L.act);
    unsigned* umask = reinterpret_cast<unsigned*>(smem + L.umk);
    float* conv = smem + L.conv;
#define AT(arr, k, i) (arr)[LEAN ? (k) * TS + (i) : (i) * NS + (k)]
    // the row of stage k of a stage array (12 or 16 features): 16 B accesses in the stage-major layout
    auto ld_row = [&](const float* arr_, int k, auto& out) {
        constexpr int n = (int)(sizeof(out) / sizeof(float));
        if constexpr (LEAN) {
#pragma unroll
            for (int g = 0; g < (n + 3) / 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(arr_ + k * TS + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r < n) out[4 * g + r] = v[r];
            }
        } else {
#pragma unroll
            for (int i = 0; i < n; ++i) out[i] = AT(arr_, k, i);
        }
    };
    auto st_row = [&](float* arr_, int k, const auto& in) {     // (the padding of a stage-major row is rewritten as zero)
        constexpr int n = (int)(sizeof(in) / sizeof(float));
        if constexpr (LEAN) {
#pragma unroll
            for (int g = 0; g < (n + 3) / 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (4 * g + r < n) ? in[4 * g + r < n ? 4 * g + r : 0] : 0.0f;
                *reinterpret_cast<f32x4*>(arr_ + k * TS + 4 * g) = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < n; ++i) AT(arr_, k, i) = in[i];
        }
    };

    float* Xg = a.X + (size_t)b * (N + 1) * NX;
    float* Ug = a.U + (size_t)b * N * NU;
    const float* pg = a.params + (size_t)b * (N + 1) * NP;
    const float* yr = a.yref + (size_t)b * (a.yref_per_stage ? (size_t)N * NY : (size_t)NY);
    const float* yre = a.yref_e + (size_t)b * NX;
    const float* x0 = a.x0 + (size_t)b * NX;
    float* At = ws + wl.At;
    float* Bt = ws + wl.Bt;
    float* Kt = ws + wl.Kt;   // transposed images of K~ (N + 1 scratch slot)
    float* Ct = ws + wl.Ct;   // transposed images of Acl~ = A~ + B~K~ (N + 1)

    // Two-wave variant, horizon within one wave: the problem's linearisation runs here, lane = node, instead of in a kernel of
    // its own (the host leaves that launch out: qp_linearizes_itself).  With a second wave on the SIMD to fill its latency this is
    // +2 % per solve call at B = 8192; alone on the SIMD it is not (measured -0.6 % at B = 1024: the resident variant keeps the
    // separate kernel).  Scratch: the region of the sweep operands, which the prologue below fills.
    if constexpr (LEAN) {
        if (qp_linearizes_itself(LEAN, N)) {
            static_assert(LIN_LDS_FLOATS <= 4 * TS * 20, "scratch of the fused linearisation");
            linearize_node<M>(a, lane <= N, b, lane, smem + L.qv);
            __threadfence_block();
            wave_sync();
        }
    }
    for (int i = 4 * lane; i < L.conv; i += 256)                // padding entries stay finite
        *reinterpret_cast<f32x4*>(smem + i) = zero4();
    if constexpr (LEAN)
        for (int i = 4 * lane; i < SA_.total; i += 256) *reinterpret_cast<f32x4*>(arr + i) = zero4();
    phase_sync();
    // Stream the problem into the LDS through registers, PRE floats per lane and array at a time,
    // with the loads of ALL arrays in flight before the first is used (an element-wise copy loop
    // pays one memory round trip per 64 floats -- some thirty in a row for a horizon of 50).
    constexpr int PRE = 13;
    auto fetch = [&](const float* src, int total, int base, float (&v)[PRE]) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int e = base + 64 * u + lane;
            v[u] = src[e < total ? e : 0];
        }
    };
    // X and U of the previous solve, read through the warm-start shift (a.shift = 0: identity)
    auto fetch_x = [&](int base, float (&v)[PRE]) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int e = base + 64 * u + lane, ec = e < (N + 1) * NX ? e : 0;
            const int k = ec / NX;
            v[u] = Xg[ec + (shifted_node(k, a.shift, N) - k) * NX];
        }
    };
    auto fetch_u = [&](int base, float (&v)[PRE]) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int e = base + 64 * u + lane, ec = e < N * NU ? e : 0;
            const bool ok = a.shift == 0 || shifted_stage_valid(ec / NU, a.shift, N);
            const float t = Ug[ec + (ok ? a.shift * NU : 0)];
            v[u] = ok ? t : 0.0f;
        }
    };
    auto drain = [&](int total, int base, const float (&v)[PRE], auto&& sink) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int e = base + 64 * u + lane;
            if (e < total) sink(e, v[u]);
        }
    };
    auto put_x = [&](int e, float v) { const int k = e / NX; AT(Xs, k, e - k * NX) = v; };
    auto put_q = [&](int e, float v) { const int k = e / NX; qv[k * TS + slot_of(e - k * NX)] = v; };   // by slot
    auto put_u = [&](int e, float v) { const int k = e / NU; AT(Us, k, e - k * NU) = v; };
    auto put_r = [&](int e, float v) { const int k = e / NU; rv[k * TS + slot_of(e - k * NU)] = v; };
    auto put_c = [&](int e, float cj) {                        // c = G u - h at the linearisation point
        const int k = e / NG, j = e - k * NG;
        const float s = fmaxf(-cj, a.s_min);
        AT(sv, k, j) = s;
        AT(lv, k, j) = a.mu0 * fast_rcp(s);
    };
    const int n_x = (N + 1) * NX, n_u = N * NU, n_c = N * NG;
    const int n_max = n_x > n_c ? n_x : n_c;
    for (int base = 0; base < n_max; base += 64 * PRE) {
        float vx[PRE], vq[PRE], vu[PRE], vr[PRE], vc[PRE];
        fetch_x(base, vx);
        fetch(ws + wl.q, n_x, base, vq);
        fetch_u(base, vu);
        fetch(ws + wl.r, n_u, base, vr);
        fetch(ws + wl.c, n_c, base, vc);
        drain(n_x, base, vx, put_x);
        drain(n_x, base, vq, put_q);
        drain(n_u, base, vu, put_u);
        drain(n_u, base, vr, put_r);
        drain(n_c, base, vc, put_c);
    }
    float* dx0s = smem + L.dx0;
    if constexpr (LEAN) {
        if (lane < NX) dx0s[lane] = x0[lane] - Xg[lane];      // node 0 is a fixed point of the shift map
    }
    float cost_l = 0.0f, mu_l = 0.0f;
    int nact_l = 0;
    for (int k = lane; k <= N; k += 64) {
        cost_l += ws[wl.cost + k];
        if (k < N) {
            const unsigned am = reinterpret_cast<const unsigned*>(ws + wl.act)[k];
            actm[k] = am;
            // coupling mask of the stage, with the index of its static variant (+1; 0 = none) on top:
            // the stage loop then dispatches on one shift instead of recomputing the index every stage
            const unsigned um = reinterpret_cast<const unsigned*>(ws + wl.umk)[k];
            // bit 24: stage k-1 (the next one of the backward sweep, whose barrier product is built in this stage's
            // body) has active constraint rows in MFMA steps that this stage's pattern does not have
            const unsigned um_prev = reinterpret_cast<const unsigned*>(ws + wl.umk)[k > 0 ? k - 1 : 0];
            const unsigned wider = (k > 0 && (M::barrier_steps(um_prev) & ~M::barrier_steps(um)) != 0u) ? 1u : 0u;
            umask[k] = um | ((unsigned)(M::static_index(um) + 1) << 16) | (wider << 24);
            nact_l += __popc(am);
            mu_l += a.mu0 * (float)__popc(am);      // s * (mu0 / s) per active row
        }
    }
    // per-lane weights of "its" column (slot layout: column slot c holds logical index ic, HS the
    // homogeneous coordinate)
    const int ic = index_of(c);
    const bool c_is_x = ic >= 0 && ic < NX, c_is_u = ic >= 0 && ic < NU;
    const float wq_c = c_is_x ? a.W[c_is_x ? ic : 0] + a.reg : 0.0f;
    const float wr_c = c_is_u ? a.W[NX + (c_is_u ? ic : 0)] + a.reg : 0.0f;
    const float we_c = c_is_x ? a.We[c_is_x ? ic : 0] + a.reg_e : 0.0f;
    // per-lane constants of the stage sweeps: which element of q / r / sqrt(D) / v each of the
    // lane's four tile registers takes (a clamped LDS feature index and a 0/1 mask), so that the
    // cost tiles of a stage are built with unconditional loads and selects (no branches)
    SweepLane sl;
    sl.init(conv, lane, HS, a.rs_free);
    f32x4 Qc, Rc, Gc;            // constant parts: diag(Wx)+reg, diag(Wu)+reg, constraint matrix G
    bool qm[4], rm[4];           // masks: the register takes an element of q / r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * q4 + r, ir = index_of(row);      // row slot and its logical index
        const bool r_is_x = ir >= 0 && ir < NX, r_is_u = ir >= 0 && ir < NU;
        Qc[r] = (row == c && r_is_x) ? wq_c : 0.0f;
        Rc[r] = (row == c && r_is_u) ? wr_c : 0.0f;
        qm[r] = (c == HS && r_is_x) || (row == HS && c_is_x);
        rm[r] = (c == HS && r_is_u);
        Gc[r] = (row < NG && c_is_u) ? M::G(a.mp, row < NG ? row : 0, c_is_u ? ic : 0) : 0.0f;   // rows: constraints
    }
    const bool is_hx_col = (c == HS);
    phase_sync();

    int status = NMPC_STATUS_MAXITER;
    bool finished = false;
    STAMP_DECL;
    float cost = 0.0f, stepn = 0.0f, alpha = 1.0f;
    // phase S of the lean variant with one node per lane (see there): the node's rows of X, U in registers
    const bool rows_in_regs = LEAN && N < 64;
    const int kx_s = lane <= N ? lane : 0, ku_s = lane < N ? lane : 0;
    float xr[LEAN ? NX : 1], ur[LEAN ? NU : 1];
    {
        cost = wave_sum(cost_l);
        const int n_act = (int)(wave_sum((float)nact_l) + 0.5f);
        STAMP(0);
        // ------------------------------------------------------------ QP: interior point loop
        const bool use_ipm = (a.n_ipm > 0) && (n_act > 0);
        const int n_sweeps = use_ipm ? a.n_ipm : 1;
        bool qp_ok = true;
        float mu_sum = wave_sum(mu_l);          // sum of s.lam over the active rows
        const bool one_stage_per_lane = N <= 64;
        // step <- step + ap (new step - step), both trajectories
        auto blend = [&](float ap) {
            batched<10>(SA_.nX, lane, [&](int i) { const float d = dX[i]; return __builtin_fmaf(ap, dXp[i] - d, d); },
                        [&](int i, float v) { dX[i] = v; });
            batched<10>(SA_.nU, lane, [&](int i) { const float d = dU[i]; return __builtin_fmaf(ap, dUp[i] - d, d); },
                        [&](int i, float v) { dU[i] = v; });
        };
        for (int ii = 0; ii < n_sweeps; ++ii) {
            float tau = 0.0f;
            if (use_ipm) {
                // barrier coefficients of this iteration, lane = stage:
                //   D = lam/s, gsq = sqrt(D), gvt = (tau/s + lam + D c)/sqrt(D)
                // so that the stage sweep only builds Gs = gsq.G and Vt = gvt.e_nx (no divides there)
                tau = fmaxf(a.sigma * mu_sum / (float)n_act, a.tau_min);
                // (with N <= 64 the update phase of the previous iteration has already written them)
                if (ii == 0 || !one_stage_per_lane)
                for (int k = lane; k < N; k += 64) {
                    const unsigned am = actm[k];
                    float uk[NU], gk[NG], sk[NG], lk[NG];
                    ld_row(Us, k, uk);
                    ld_row(sv, k, sk);
                    ld_row(lv, k, lk);
                    M::gdot(a.mp, uk, gk);
#pragma unroll
                    for (int j = 0; j < NG; ++j) {
                        const bool on = (am >> j) & 1u;
                        const float s = sk[j], l = lk[j], cj = gk[j] - M::h(a.mp, j);
                        const float is = fast_rcp(s);
                        const float D = l * is;
                        const float rs = __builtin_amdgcn_rsqf(D);
                        gsq[k * TS + j] = on ? D * rs : 0.0f;
                        gvt[k * TS + j] = on ? __builtin_fmaf(D, cj, __builtin_fmaf(tau, is, l)) * rs : 0.0f;
                    }
                }
                phase_sync();
            }
            STAMP(1);
            // -------------------------------------------------------- phase R: backward sweep
            f32x4 P;
            {   // terminal: P~ = [diag(We)+reg_e, q_N; q_N', 0]
                const float qc = c_is_x ? qv[N * TS + c] : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * q4 + r, ir = index_of(row);
                    const bool r_is_x = ir >= 0 && ir < NX;
                    float v = 0.0f;
                    if (row == c && r_is_x) v = we_c;
                    if (c == HS && r_is_x) v = qv[N * TS + row];
                    if (row == HS && c_is_x) v = qc;
                    P[r] = v;
                }
            }
            // Backward sweep.  The wave is alone on its SIMD at B = 1024, so nothing else hides
            // latency: stage images are prefetched one stage ahead and the cost tiles of the next
            // stage are built inside the current one.
            auto sweep = [&](auto ipm_tag) {
                constexpr bool IPM = decltype(ipm_tag)::value;
                // The additive cost tiles of a stage (NextCost of backward_stage): Q~ = [Q q; q' 0],
                // S~ = [0 r], R, and with the barrier Gs = sqrt(D).G, vt = v/sqrt(D):
                // T = Gs'[Gs | vt] holds G'DG in its columns < nu and G'v in column nx (ONE product, the
                // Gauss-Newton contraction of the barrier); finish() splits it onto R and S~.
                struct NextCost {
                    f32x4 Gs, Vt, Tb, Qn, Sn, Rn;
                    const float *qrow, *rrow, *sqrow, *vtrow;   // 16 B of this lane's row quad of q, r, sqrt(D), vt
                    const float* qcol;                          // q at this lane's column slot
                    const bool *qm, *rm;
                    f32x4 Qc, Rc, Gc;
                    bool hx_col;
                    f32x4 q4v, r4v, sq4, vt4;
                    float qcv;
                    __device__ __forceinline__ void fetch() {
                        q4v = *reinterpret_cast<const f32x4*>(qrow);
                        r4v = *reinterpret_cast<const f32x4*>(rrow);
                        qcv = *qcol;
                        if constexpr (IPM) {
                            sq4 = *reinterpret_cast<const f32x4*>(sqrow);
                            vt4 = *reinterpret_cast<const f32x4*>(vtrow);
                        }
                    }
                    __device__ __forceinline__ void build() {
                        Rn = Rc;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            Qn[r] = Qc[r] + (qm[r] ? (hx_col ? q4v[r] : qcv) : 0.0f);
                            Sn[r] = rm[r] ? r4v[r] : 0.0f;
                        }
                        if constexpr (IPM) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                Gs[r] = Gc[r] * sq4[r];                  // rows >= ng: Gc = 0
                                Vt[r] = hx_col ? vt4[r] : Gs[r];         // [Gs | vt]
                            }
                            Tb = zero4();
                        }
                    }
                    __device__ __forceinline__ void mfma(int i) {
                        if constexpr (IPM && !BF16B) Tb = __builtin_amdgcn_mfma_f32_16x16x4f32(Gs[i], Vt[i], Tb, 0, 0, 0);
                        if constexpr (IPM && BF16B) { if (i == 0) Tb = xty_bf16(Gs, Vt); }
                    }
                    __device__ __forceinline__ void finish() {
                        if constexpr (IPM) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                Rn[r] += hx_col ? 0.0f : Tb[r];     // columns >= nu of Tb are zero
                                Sn[r] += hx_col ? Tb[r] : 0.0f;
                            }
                        }
                    }
                };
                auto next_cost = [&](int kk) {
                    NextCost nc;
                    nc.qrow = qv + kk * TS + 4 * q4; nc.rrow = rv + kk * TS + 4 * q4;
                    nc.sqrow = gsq + kk * TS + 4 * q4; nc.vtrow = gvt + kk * TS + 4 * q4;
                    nc.qcol = qv + kk * TS + c;
                    nc.qm = qm; nc.rm = rm;
                    nc.Qc = Qc; nc.Rc = Rc; nc.Gc = Gc; nc.hx_col = is_hx_col;
                    return nc;
                };
                ImageLane<M> il;
                il.init(lane);
                RowStoreLane ks, cs;
                ks.init(lane, G::KQ, G::K_FLOATS, N - 1, N);
                cs.init(lane, G::CQ, G::C_FLOATS, N - 1, N);
                f32x4 A0 = il.fix_A(il.load_A(At + (size_t)(N - 1) * G::A_FLOATS), lane);
                f32x4 B0 = il.fix_B(il.load_B(Bt + (size_t)(N - 1) * G::B_FLOATS), lane);
                f32x4 T0 = il.fix_Bt(il.load_Bt(Bt + (size_t)(N - 1) * G::B_FLOATS), lane);
                f32x4 Qt, St, Rt;
                {   // prologue: cost tiles of stage N-1
                    NextCost nc = next_cost(N - 1);
                    nc.fetch();
                    nc.build();
#pragma unroll
                    for (int i = 0; i < 4; ++i) nc.mfma(i);
                    nc.finish();
                    Qt = nc.Qn; St = nc.Sn; Rt = nc.Rn;
                }
                unsigned cm = umask[N - 1];
                int lane_zero = 0;
                asm volatile("" : "+v"(lane_zero));
                for (int k = N - 1; k >= 0; --k) {
#ifdef NMPC_STAMPS
                    sst.t0 = __builtin_readcyclecounter();
#endif
                    // next stage's images and coupling mask: issued first, consumed by the register
                    // rotation at the end of this stage -- by then only this stage's K~/Acl~ stores are
                    // younger (vmcnt retires in order), so neither the loads nor the stores stall the sweep
                    const int kn = (k > 0) ? k - 1 : 0;
#ifdef QP_T_ONEIMG      // timing build (tools/qp_traffic_timing.sh): every stage reads the images of stage 0 (cache-resident) -- results are wrong
                    const int ki = (a.B < 0) ? kn : 0;
#else
                    const int ki = kn;
#endif
                    const f32x4 A1 = il.load_A(At + (size_t)ki * G::A_FLOATS);
                    const f32x4 B1 = il.load_B(Bt + (size_t)ki * G::B_FLOATS);
                    const f32x4 T1 = il.load_Bt(Bt + (size_t)ki * G::B_FLOATS);
                    // (read through an offset the compiler cannot see is uniform: the value stays in a
                    // VGPR until the end of the stage instead of being scalarised, and waited for, here)
                    const unsigned cm_next = umask[kn + lane_zero];
                    f32x4 Kk, Acl;
                    NextCost sh = next_cost(kn);
                    // a static body holds the barrier-product steps of its own stance feet only: exact zeros elsewhere
                    // while the next stage's pattern is the same or narrower.  The rare stage before a touch-down
                    // (bit 24 of the mask word) runs the body of the widest pattern instead -- valid for any pattern
                    // (it eliminates every pivot for real), all four steps, and no extra code on the common path
                    auto run = [&](auto mask_tag) {
                        constexpr unsigned MK = decltype(mask_tag)::value;
                        constexpr unsigned STEPS = (IPM && !BF16B && MK != DYNAMIC_MASK) ? M::barrier_steps(MK) : 0xFu;
                        return backward_stage<NU, MK, true, ALLV, STEPS, LEAN ? 0 : 2>(P, A0, B0, T0, Qt, St, Rt, conv, sl, lane,
                                                                        cm & 0xFFFFu, Kk, Acl, sh SST_PASS);
                    };
                    bool ok = true;
                    // one static variant per mask: the model's short list as a chain (ALLV = false), or
                    // every listed variant through a decision tree; any other mask takes the run-time fallback
#define NMPC_VARIANT(I)                                                                                   \
    case I:                                                                                               \
        if constexpr (I < M::N_STATIC_MASKS)                                                              \
            ok = run(std::integral_constant<unsigned, M::static_mask(I < M::N_STATIC_MASKS ? I : 0)>{}); \
        else ok = run(std::integral_constant<unsigned, DYNAMIC_MASK>{});                                  \
        break;
#define NMPC_COMMON(J) (M::common_variant(J) >= 0 && vi == M::common_variant(J))
#define NMPC_RUN_COMMON(J) run(std::integral_constant<unsigned, M::static_mask(M::common_variant(J) >= 0 ? M::common_variant(J) : 0)>{})
                    constexpr int WIDEST = M::static_index((NU < 32) ? ((1u << NU) - 1u) : 0xFFFFFFFFu);
                    const int vi = ((cm >> 24) & 1u) ? WIDEST : (int)((cm >> 16) & 0x1Fu) - 1;
                    if constexpr (!ALLV) {
                        if (NMPC_COMMON(0)) ok = NMPC_RUN_COMMON(0);
                        else if (NMPC_COMMON(1)) ok = NMPC_RUN_COMMON(1);
                        else if (NMPC_COMMON(2)) ok = NMPC_RUN_COMMON(2);
                        else if (NMPC_COMMON(3)) ok = NMPC_RUN_COMMON(3);
                        else ok = run(std::integral_constant<unsigned, DYNAMIC_MASK>{});
                    } else {
                        switch (vi) {
                            NMPC_VARIANT(0) NMPC_VARIANT(1) NMPC_VARIANT(2) NMPC_VARIANT(3)
                            NMPC_VARIANT(4) NMPC_VARIANT(5) NMPC_VARIANT(6) NMPC_VARIANT(7)
                            NMPC_VARIANT(8) NMPC_VARIANT(9) NMPC_VARIANT(10) NMPC_VARIANT(11)
                            NMPC_VARIANT(12) NMPC_VARIANT(13) NMPC_VARIANT(14) NMPC_VARIANT(15)
                            default: ok = run(std::integral_constant<unsigned, DYNAMIC_MASK>{});
                        }
                    }
#undef NMPC_VARIANT
#undef NMPC_COMMON
#undef NMPC_RUN_COMMON
                    qp_ok = ok && qp_ok;
#ifdef QP_T_NOSTORE     // timing build: no gain stores -- results are wrong, the time tells what the stores cost
                    if (a.B < 0)
#endif
                    {
                    ks.store(Kt, Kk);      // gain tiles of this stage, read by the forward sweep
                    cs.store(Ct, Acl);
                    }
                    A0 = il.fix_A(A1, lane);
                    B0 = il.fix_B(B1, lane);
                    T0 = il.fix_Bt(T1, lane);
                    Qt = sh.Qn; St = sh.Sn; Rt = sh.Rn;
                    cm = __builtin_amdgcn_readfirstlane(cm_next);
                    SST_TILES(5);
                }
            };
            if (use_ipm) sweep(std::true_type{}); else sweep(std::false_type{});
            __threadfence_block();          // K~/Acl~ images: stored above, read by other lanes below
            wave_sync();
            STAMP(2);
            // -------------------------------------------------------- phase F: forward sweep
            // dx~+ = Acl~ dx~ , du = K~ dx~ as ONE row-per-lane mat-vec: lane i < nx holds row i of
            // Acl~ (the stored image is row-major, columns by slot), lane 16+j row j of K~; dx~ is
            // wave-uniform (SGPRs, refreshed by v_readlane).  13 FMAs + 12 readlanes per stage -- fp32
            // MFMA and VALU do not overlap on a SIMD, so the 16-column MFMA mat-vec cost 8x32 cycles.
            // Rows are prefetched FWD_PF stages ahead (a stage is shorter than an L2 miss).
            float* oX = use_ipm ? dXp : dX;
            float* oU = use_ipm ? dUp : dU;
            float vs[NX];
            if constexpr (LEAN) {      // (the same difference, kept in the LDS: X lives in the workspace)
#pragma unroll
                for (int i = 0; i < NX; ++i) vs[i] = dx0s[i];
                if (lane < NX) AT(oX, 0, lane) = dx0s[lane];
            } else {
#pragma unroll
                for (int i = 0; i < NX; ++i) vs[i] = x0[i] - AT(Xs, 0, i);
                if (lane < NX) AT(oX, 0, lane) = x0[lane] - AT(Xs, 0, lane);
            }
            constexpr int FWD_PF = 4, RQ4 = slot_of(NX - 1) / 4 + 1;
            const bool is_x = lane < NX, is_u = (lane >= 16 && lane < 16 + NU);
            // row of this lane in the image of stage 0, as a byte offset from the workspace base
            unsigned rowoff = 4u * (unsigned)((is_x ? wl.Ct : wl.Kt) + (is_x ? lane : is_u ? lane - 16 : 0) * TS);
            const unsigned rowstep = 4u * (is_x ? G::C_FLOATS : G::K_FLOATS);
            float* dst = is_x ? &AT(oX, 1, lane) : is_u ? &AT(oU, 0, lane - 16) : idle_sink;
            const int dstep = (is_x || is_u) ? (LEAN ? TS : 1) : 0;
            // the used floats of a row: quad 0 (three states and the homogeneous slot) as a 16 B load,
            // the other quads as 12 B loads (a load with dead elements lets the allocator reuse them at
            // once -- which means a wait on the load in flight)
            auto load_row = [&](unsigned off, f32x4 (&row)[RQ4]) {
                row[0] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ws) + (size_t)off);
#pragma unroll
                for (int g = 1; g < RQ4; ++g) {
                    const f32x3u v = *reinterpret_cast<const f32x3u*>(reinterpret_cast<const char*>(ws) + (size_t)off + 16 * g);
                    row[g] = f32x4{v[0], v[1], v[2], 0.0f};
                }
            };
            f32x4 ring[FWD_PF][RQ4];
#pragma unroll
            for (int j = 0; j < FWD_PF; ++j) load_row(rowoff + (unsigned)((j < N) ? j : N - 1) * rowstep, ring[j]);
            const unsigned rowlast = rowoff + (unsigned)(N - 1) * rowstep;
            rowoff += FWD_PF * rowstep;        // next row to prefetch
            // one stage: consume a ring slot, then refill it with the row FWD_PF stages ahead (the
            // refill comes after the last use, so old and new value share registers and the loop
            // carries no copies -- a copy of fresh load data would put a full wait into every stage)
            auto fwd_stage = [&](int k, f32x4 (&slot)[RQ4]) {
                // four accumulator chains: a dependent v_fmac issues only every ~11 cycles, four chains
                // keep the 4-cycle issue rate (tools/ubench/ubench_dpp.hip).  asm: left to itself the
                // compiler re-associates the sum and, worse, re-schedules the ring refills around it.
                float acc = slot[0][HS], acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;   // slot HS: times dx~[HS] = 1
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const float e = slot[slot_of(i) >> 2][slot_of(i) & 3];
                    float& ac = (i & 3) == 0 ? acc : (i & 3) == 1 ? acc1 : (i & 3) == 2 ? acc2 : acc3;
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(ac) : "s"(vs[i]), "v"(e));
                }
                acc = (acc + acc1) + (acc2 + acc3);
#ifdef QP_T_ONEIMG
                load_row((a.B < 0) ? rowoff : rowlast, slot);
#else
                load_row(rowoff < rowlast ? rowoff : rowlast, slot);
#endif
                rowoff += rowstep;
                dst[k * dstep] = acc;
#pragma unroll
                for (int i = 0; i < NX; ++i) vs[i] = bcast(acc, i);
                __builtin_amdgcn_sched_barrier(0);   // keep each refill inside its own stage
            };
            int k0 = 0;
            for (; k0 + FWD_PF <= N; k0 += FWD_PF) {     // whole groups: one straight-line body
#pragma unroll
                for (int j = 0; j < FWD_PF; ++j) fwd_stage(k0 + j, ring[j]);
            }
#pragma unroll
            for (int j = 0; j < FWD_PF - 1; ++j)          // the last N % FWD_PF stages
                if (k0 + j < N) fwd_stage(k0 + j, ring[j]);
            phase_sync();
            STAMP(3);
            // -------------------------------------------------------- phase I: IPM update
            if (use_ipm && one_stage_per_lane) {
                // Lane = stage, everything of the stage in registers: step direction
                //   ds = -(G du+ + c) - s ,  dlam = tau/s - lam - (lam/s) ds ,
                // fraction to the boundary  alpha = min(1, gamma / max(-d./.))  (one division per wave),
                // update, complementarity sum, and the barrier coefficients of the next iteration.
                const bool live = lane < N;
                const int k = live ? lane : 0;
                float du[NU], g[NG], uk[NU], cc[NG], s[NG], l[NG], ds[NG], dl[NG];
                ld_row(dUp, k, du);
                ld_row(Us, k, uk);
                ld_row(sv, k, s);
                ld_row(lv, k, l);
                // lean variant: the blend of the step with the new one rides in this phase (the stage's rows, requested
                // here, used once the step length is known) instead of two element-wise passes over the workspace
                const int kx = (lane <= N) ? lane : 0;
                float dxo[LEAN ? NX : 1], dxn[LEAN ? NX : 1], duo[LEAN ? NU : 1];
                if constexpr (LEAN) { ld_row(dX, kx, dxo); ld_row(dXp, kx, dxn); ld_row(dU, k, duo); }
                M::gdot(a.mp, du, g);
                M::gdot(a.mp, uk, cc);          // c = G u - h at the linearisation point
                const unsigned am = live ? actm[k] : 0u;
                float rp = 0.0f, rd = 0.0f;
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    cc[j] -= M::h(a.mp, j);
                    const float is = fast_rcp(s[j]);
                    ds[j] = -(g[j] + cc[j]) - s[j];
                    dl[j] = __builtin_fmaf(-(l[j] * is), ds[j], __builtin_fmaf(tau, is, -l[j]));
                    const bool on = (am >> j) & 1u;
                    rp = on ? fmaxf(rp, -ds[j] * is) : rp;
                    rd = on ? fmaxf(rd, -dl[j] * __builtin_amdgcn_rcpf(l[j])) : rd;
                }
                const float rpm = wave_max(rp), rdm = wave_max(rd);
                const float ap = rpm > a.gamma ? a.gamma / rpm : 1.0f;
                const float ad = rdm > a.gamma ? a.gamma / rdm : 1.0f;
                float m_l = 0.0f;
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const bool on = (am >> j) & 1u;
                    s[j] = on ? __builtin_fmaf(ap, ds[j], s[j]) : s[j];
                    l[j] = on ? __builtin_fmaf(ad, dl[j], l[j]) : l[j];
                    m_l = on ? __builtin_fmaf(s[j], l[j], m_l) : m_l;
                }
                if (live) { st_row(sv, k, s); st_row(lv, k, l); }
                mu_sum = wave_sum(m_l);
                if constexpr (LEAN) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) dxo[i] = __builtin_fmaf(ap, dxn[i] - dxo[i], dxo[i]);
#pragma unroll
                    for (int i = 0; i < NU; ++i) duo[i] = __builtin_fmaf(ap, du[i] - duo[i], duo[i]);
                    if (lane <= N) st_row(dX, kx, dxo);
                    if (live) st_row(dU, k, duo);
                    if (N == 64 && lane < NX) {            // node 64 has no lane of its own
                        const float d = AT(dX, 64, lane);
                        AT(dX, 64, lane) = __builtin_fmaf(ap, AT(dXp, 64, lane) - d, d);
                    }
                } else {
                    blend(ap);
                }
                if (ii + 1 < n_sweeps && live) {
                    const float tau_n = fmaxf(a.sigma * mu_sum / (float)n_act, a.tau_min);
                    float sq[NG], vt[NG];
#pragma unroll
                    for (int j = 0; j < NG; ++j) {
                        const bool on = (am >> j) & 1u;
                        const float is = fast_rcp(s[j]);
                        const float D = l[j] * is;
                        const float rs = __builtin_amdgcn_rsqf(D);
                        sq[j] = on ? D * rs : 0.0f;
                        vt[j] = on ? __builtin_fmaf(D, cc[j], __builtin_fmaf(tau_n, is, l[j])) * rs : 0.0f;
                    }
#pragma unroll
                    for (int j = 0; j < NG; ++j) { gsq[k * TS + j] = sq[j]; gvt[k * TS + j] = vt[j]; }
                }
                phase_sync();
            } else if (use_ipm) {
                float ap_l = 1.0f, ad_l = 1.0f;
                // long horizons (several stages per lane): the same update in two passes.
                // ds = -(G du+ + c) - s ;  dlam = tau/s - lam - (lam/s) ds   (recomputed in the
                // second pass rather than kept in runtime-indexed arrays, which would go to scratch)
                auto step_dir = [&](int k, int j, const float (&g)[NG], const float (&cc)[NG], float& s, float& l, float& dsj, float& dlj) {
                    s = AT(sv, k, j); l = AT(lv, k, j);
                    const float is = fast_rcp(s);
                    dsj = -(g[j] + cc[j]) - s;
                    dlj = __builtin_fmaf(-(l * is), dsj, __builtin_fmaf(tau, is, -l));
                };
                for (int k = lane; k < N; k += 64) {
                    float du[NU], g[NG], uk[NU], cc[NG];
#pragma unroll
                    for (int i = 0; i < NU; ++i) { du[i] = AT(dUp, k, i); uk[i] = AT(Us, k, i); }
                    M::gdot(a.mp, du, g);
                    M::gdot(a.mp, uk, cc);          // c = G u - h, recomputed (kept out of LDS)
#pragma unroll
                    for (int j = 0; j < NG; ++j) cc[j] -= M::h(a.mp, j);
                    const unsigned am = actm[k];
#pragma unroll
                    for (int j = 0; j < NG; ++j) {
                        float s, l, dsj, dlj;
                        step_dir(k, j, g, cc, s, l, dsj, dlj);
                        const bool on = (am >> j) & 1u;
                        if (on && dsj < 0.0f) ap_l = fminf(ap_l, -a.gamma * s / dsj);
                        if (on && dlj < 0.0f) ad_l = fminf(ad_l, -a.gamma * l / dlj);
                    }
                }
                const float ap = wave_min(ap_l), ad = wave_min(ad_l);
                float m_l = 0.0f;
                for (int k = lane; k < N; k += 64) {
                    float du[NU], g[NG], uk[NU], cc[NG];
#pragma unroll
                    for (int i = 0; i < NU; ++i) { du[i] = AT(dUp, k, i); uk[i] = AT(Us, k, i); }
                    M::gdot(a.mp, du, g);
                    M::gdot(a.mp, uk, cc);          // c = G u - h, recomputed (kept out of LDS)
#pragma unroll
                    for (int j = 0; j < NG; ++j) cc[j] -= M::h(a.mp, j);
                    const unsigned am = actm[k];
#pragma unroll
                    for (int j = 0; j < NG; ++j) {
                        float s, l, dsj, dlj;
                        step_dir(k, j, g, cc, s, l, dsj, dlj);
                        const bool on = (am >> j) & 1u;
                        s = on ? __builtin_fmaf(ap, dsj, s) : s;
                        l = on ? __builtin_fmaf(ad, dlj, l) : l;
                        AT(sv, k, j) = s;
                        AT(lv, k, j) = l;
                        if (on) m_l = __builtin_fmaf(s, l, m_l);
                    }
                }
                mu_sum = wave_sum(m_l);
                blend(ap);
                phase_sync();
            }
        }
        STAMP(4);
        // ------------------------------------------------------------ phase S: step
        float sn_l = 0.0f;
        bool bad_l = false;
        auto step_norm = [&](int, float v) {
            bad_l = bad_l || !(fabsf(v) <= 1e30f);
            sn_l = fmaxf(sn_l, fabsf(v));
        };
        // lean variant, one node per lane: the node's rows go through registers once -- norm, step and the store to the
        // caller's X, U -- instead of six element-wise passes over the workspace (each a memory round trip)
        float dxr[LEAN ? NX : 1], dur[LEAN ? NU : 1];
        if constexpr (LEAN) {
            if (rows_in_regs) {
                ld_row(dX, kx_s, dxr); ld_row(Xs, kx_s, xr); ld_row(dU, ku_s, dur); ld_row(Us, ku_s, ur);
#pragma unroll
                for (int i = 0; i < NX; ++i) step_norm(0, lane <= N ? dxr[i] : 0.0f);
#pragma unroll
                for (int i = 0; i < NU; ++i) step_norm(0, lane < N ? dur[i] : 0.0f);
            }
        }
        if (!rows_in_regs) {
            batched<10>(SA_.nX, lane, [&](int i) { return dX[i]; }, step_norm);
            batched<10>(SA_.nU, lane, [&](int i) { return dU[i]; }, step_norm);
        }
        stepn = wave_max(sn_l);
        const bool bad = __any(bad_l);
        if (bad) { status = NMPC_STATUS_NAN; finished = true; }
        alpha = 1.0f;
        if (a.line_search && !bad) {
            // l1 merit: cost + rho * (|x0 - X0| + sum |defects| + sum max(0, G u - h))
            auto merit = [&](float al) -> float {
                float m_l = 0.0f;
                for (int k = lane; k < N; k += 64) {
                    float x[NX], u[NU], xn[NX], p[NP > 0 ? NP : 1];
#pragma unroll
                    for (int i = 0; i < NX; ++i) x[i] = __builtin_fmaf(al, AT(dX, k, i), AT(Xs, k, i));
#pragma unroll
                    for (int i = 0; i < NU; ++i) u[i] = __builtin_fmaf(al, AT(dU, k, i), AT(Us, k, i));
#pragma unroll
                    for (int i = 0; i < NP; ++i) p[i] = pg[(size_t)k * NP + i];
                    M::step(a.mp, x, u, p, xn);
                    const float* yk = yr + (a.yref_per_stage ? (size_t)k * NY : 0);
                    float viol = 0.0f, cst = 0.0f;
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        const float e = x[i] - yk[i];
                        cst += 0.5f * a.W[i] * e * e;
                        viol += fabsf(xn[i] - __builtin_fmaf(al, AT(dX, k + 1, i), AT(Xs, k + 1, i)));
                    }
#pragma unroll
                    for (int i = 0; i < NU; ++i) {
                        const float e = u[i] - yk[NX + i];
                        cst += 0.5f * a.W[NX + i] * e * e;
                    }
                    if (a.n_ipm > 0) {
                        float g[NG];
                        M::gdot(a.mp, u, g);
                        const unsigned am = M::active_mask(a.mp, p);
#pragma unroll
                        for (int j = 0; j < NG; ++j)
                            if ((am >> j) & 1u) viol += fmaxf(g[j] - M::h(a.mp, j), 0.0f);
                    }
                    m_l += cst + a.rho * viol;
                }
                if (lane < NX) {
                    const float xe = __builtin_fmaf(al, AT(dX, N, lane), AT(Xs, N, lane));
                    const float e = xe - yre[lane];
                    m_l += 0.5f * a.We[lane] * e * e;
                    m_l += a.rho * fabsf(x0[lane] - __builtin_fmaf(al, AT(dX, 0, lane), AT(Xs, 0, lane)));
                }
                return wave_sum(m_l);
            };
            const float m0 = merit(0.0f);
            for (int t = 0; t < 6; ++t) {
                const float m1 = merit(alpha);
                if (m1 < m0 || t == 5) break;
                alpha *= 0.5f;
            }
        }
        if (!bad) {
            if (rows_in_regs) {
                if constexpr (LEAN) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) xr[i] = __builtin_fmaf(alpha, dxr[i], xr[i]);
#pragma unroll
                    for (int i = 0; i < NU; ++i) ur[i] = __builtin_fmaf(alpha, dur[i], ur[i]);
                }
            } else {
                batched<10>(SA_.nX, lane, [&](int i) { return __builtin_fmaf(alpha, dX[i], Xs[i]); }, [&](int i, float v) { Xs[i] = v; });
                batched<10>(SA_.nU, lane, [&](int i) { return __builtin_fmaf(alpha, dU[i], Us[i]); }, [&](int i, float v) { Us[i] = v; });
                phase_sync();
            }
            if (!qp_ok) { status = NMPC_STATUS_QP; finished = true; }
            else if (a.nlp_tol > 0.0f && stepn < a.nlp_tol) { status = NMPC_STATUS_OK; finished = true; }
        }
    }
    // a NaN step leaves the iterate of the previous iteration -- which, in a call that folded a warm-start shift into its
    // first iteration, is the SHIFTED previous solution: it is written back so that the caller's node bookkeeping
    // (last_node already advanced) and the trajectory stay aligned
    if (status != NMPC_STATUS_NAN || a.shift > 0) {
        if (rows_in_regs) {
            if constexpr (LEAN) {
                if (lane <= N) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) Xg[lane * NX + i] = xr[i];
                }
                if (lane < N) {
#pragma unroll
                    for (int i = 0; i < NU; ++i) Ug[lane * NU + i] = ur[i];
                }
            }
        } else {
            batched<10>((N + 1) * NX, lane, [&](int e) { const int k = e / NX; return AT(Xs, k, e - k * NX); },
                        [&](int e, float v) { Xg[e] = v; });
            batched<10>(N * NU, lane, [&](int e) { const int k = e / NU; return AT(Us, k, e - k * NU); },
                        [&](int e, float v) { Ug[e] = v; });
        }
    }
    STAMP(5);
#ifdef NMPC_STAMPS
    if (lane == 0 && a.dbg) {
        for (int i = 0; i < 8; ++i) a.dbg[16 * b + i] = (float)st_acc[i];
        for (int i = 0; i < 8; ++i) a.dbg[16 * b + 8 + i] = (float)sst.acc[i];
    }
#endif
    if (lane == 0) {
        flag[0] = finished ? 1 : 0;
        if (a.status) a.status[b] = status;
        if (a.stats) {
            a.stats[4 * b + 0] = cost;
            a.stats[4 * b + 1] = stepn;
            a.stats[4 * b + 2] = alpha;
            a.stats[4 * b + 3] = (float)(a.it + 1);
        }
    }
#undef AT
}

}  // namespace nmpc
