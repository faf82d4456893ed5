// nmpc_models.hpp -- the two declared plant models, evaluated lane-locally (lane = stage).
//
//  DoubleIntegrator  SURVEY.md 9.2 (BASELINE config 1)
//  Centroidal        SURVEY.md 9.3 (BASELINE config 2): single rigid body + 4 point feet,
//                    state slots in the order of the reference's 12-vector base cost
//                    (mpc_controller/utils/dynamics.py:121-124), forces in the order of f_sol
//                    (mpc_controller/utils/solver.py:417-421), Euler-rate map of
//                    mpc_controller/utils/transform.py:72-78, R = Rz Ry Rx (mpc.py:205).
// Mathematics: DESIGN.md section 3.  Each model provides
//   step()       x+ = phi(x,u,p)
//   linearize()  x+ and the columns of A = dphi/dx, B = dphi/du through emit callbacks
//   G(j,c), h(j), active_mask(), gdot()   the stage inequality  G u <= h  (linear in u)
#pragma once
#include "nmpc_tile.hpp"

// Floating-point contraction is OFF for the model functions (restored at the end of this header): they are inlined into the
// linearisation kernel AND into the QP kernel's two-wave variant, which linearises its problem itself -- and with contraction
// left to the backend the two instantiations fuse different products (the variants are compared bit for bit).  Unfused is also
// what the test oracle's C code is compiled as (-ffp-contract=off).
#pragma clang fp contract(off)

namespace nmpc {

struct ModelParams {
    float dt, mass, ixx, iyy, izz, gz, mu, umax;
    // whole-body model (nmpc_wb_model.hpp): Baumgarte gain of the stance constraint and the leg geometry
    float p_gain, hipx, hipy, lhip, l1, l2, res0, res1;
};

// ------------------------------------------------------------------------------------------------
struct DoubleIntegrator {
    static constexpr int ID = 0, NX = 4, NU = 2, NP = 0, NG = 4;

    __device__ static void step(const ModelParams& mp, const float (&x)[NX], const float (&u)[NU],
                                const float*, float (&xn)[NX]) {
        const float dt = mp.dt, h2 = 0.5f * dt * dt;
        xn[0] = x[0] + dt * x[2] + h2 * u[0];
        xn[1] = x[1] + dt * x[3] + h2 * u[1];
        xn[2] = x[2] + dt * u[0];
        xn[3] = x[3] + dt * u[1];
    }
    template <class EA, class EB>
    __device__ static void linearize(const ModelParams& mp, const float (&x)[NX], const float (&u)[NU],
                                     const float* p, float (&xn)[NX], EA&& emitA, EB&& emitB) {
        step(mp, x, u, p, xn);
        const float dt = mp.dt, h2 = 0.5f * dt * dt;
        { float c[NX] = {1.f, 0.f, 0.f, 0.f}; emitA(0, c); }
        { float c[NX] = {0.f, 1.f, 0.f, 0.f}; emitA(1, c); }
        { float c[NX] = {dt, 0.f, 1.f, 0.f}; emitA(2, c); }
        { float c[NX] = {0.f, dt, 0.f, 1.f}; emitA(3, c); }
        { float c[NX] = {h2, 0.f, dt, 0.f}; emitB(0, c); }
        { float c[NX] = {0.f, h2, 0.f, dt}; emitB(1, c); }
    }
    // rows: +u0, -u0, +u1, -u1 <= umax
    __device__ static float G(const ModelParams&, int j, int c) {
        return (c == (j >> 1)) ? ((j & 1) ? -1.f : 1.f) : 0.f;
    }
    __device__ static float h(const ModelParams& mp, int) { return mp.umax; }
    __device__ static unsigned active_mask(const ModelParams& mp, const float*) {
        return mp.umax > 0.f ? 0xFu : 0u;
    }
    __device__ static void gdot(const ModelParams&, const float (&v)[NU], float (&o)[NG]) {
#pragma clang fp contract(off)   // same rounding in every caller: the QP kernel's variants are compared bit for bit
        o[0] = v[0]; o[1] = -v[0]; o[2] = v[1]; o[3] = -v[1];
    }
    // inputs that couple with others in Huu: all of them, always -> one static variant
    static constexpr int N_STATIC_MASKS = 1;
    __host__ __device__ static constexpr unsigned static_mask(int) { return 0x3u; }
    __host__ __device__ static constexpr int static_index(unsigned mask) { return mask == 0x3u ? 0 : -1; }
    // variants of the short dispatch list (see Centroidal); -1: unused entry
    __host__ __device__ static constexpr int common_variant(int i) { return i == 0 ? 0 : -1; }
    // MFMA steps of the barrier product that can hold active rows under an input mask (all: the box rows)
    __host__ __device__ static constexpr unsigned barrier_steps(unsigned) { return 0xFu; }
    __device__ static unsigned input_mask(const ModelParams&, const float*) { return 0x3u; }
};

// ------------------------------------------------------------------------------------------------
struct Centroidal {
    static constexpr int ID = 1, NX = 12, NU = 12, NP = 16, NG = 16;

    struct Trig { float cz, sz, cy, sy, cx, sx; };

    // R = Rz(yaw) Ry(pitch) Rx(roll), row-major
    __device__ static void rot(const Trig& t, float (&R)[9]) {
        R[0] = t.cz * t.cy; R[1] = t.cz * t.sy * t.sx - t.sz * t.cx; R[2] = t.cz * t.sy * t.cx + t.sz * t.sx;
        R[3] = t.sz * t.cy; R[4] = t.sz * t.sy * t.sx + t.cz * t.cx; R[5] = t.sz * t.sy * t.cx - t.cz * t.sx;
        R[6] = -t.sy;       R[7] = t.cy * t.sx;                       R[8] = t.cy * t.cx;
    }
    // o = M' v
    __device__ static void tmul(const float (&M)[9], const float (&v)[3], float (&o)[3]) {
        o[0] = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
        o[1] = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
        o[2] = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
    }

    // shared front part: net force, world torque, next velocities.  w = (wx,wy,wz).
    struct Core {
        Trig t; float R[9]; float F[3]; float tau[3]; float w[3]; float vn[3]; float wn[3]; float T[9];
    };
    __device__ static void core(const ModelParams& mp, const float (&x)[NX], const float (&u)[NU],
                                const float* p, Core& k) {
        sincosf(x[3], &k.t.sz, &k.t.cz);
        sincosf(x[4], &k.t.sy, &k.t.cy);
        sincosf(x[5], &k.t.sx, &k.t.cx);
        rot(k.t, k.R);
        k.w[0] = x[11]; k.w[1] = x[10]; k.w[2] = x[9];
        k.F[0] = k.F[1] = k.F[2] = 0.f;
        k.tau[0] = k.tau[1] = k.tau[2] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float c = p[i];
            const float ax = p[4 + 3 * i] - x[0], ay = p[5 + 3 * i] - x[1], az = p[6 + 3 * i] - x[2];
            const float fx = u[3 * i], fy = u[3 * i + 1], fz = u[3 * i + 2];
            k.F[0] += c * fx; k.F[1] += c * fy; k.F[2] += c * fz;
            k.tau[0] += c * (ay * fz - az * fy);
            k.tau[1] += c * (az * fx - ax * fz);
            k.tau[2] += c * (ax * fy - ay * fx);
        }
        float tb[3];
        tmul(k.R, k.tau, tb);
        const float I[3] = {mp.ixx, mp.iyy, mp.izz};
        const float gy[3] = {(I[2] - I[1]) * k.w[1] * k.w[2], (I[0] - I[2]) * k.w[2] * k.w[0],
                             (I[1] - I[0]) * k.w[0] * k.w[1]};
        const float im = 1.0f / mp.mass;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            k.vn[j] = x[6 + j] + mp.dt * (k.F[j] * im + (j == 2 ? mp.gz : 0.f));
            k.wn[j] = k.w[j] + mp.dt * (tb[j] - gy[j]) / I[j];
        }
        const float icy = 1.0f / k.t.cy, ty = k.t.sy * icy;
        k.T[0] = 0.f; k.T[1] = k.t.sx * icy; k.T[2] = k.t.cx * icy;
        k.T[3] = 0.f; k.T[4] = k.t.cx;       k.T[5] = -k.t.sx;
        k.T[6] = 1.f; k.T[7] = k.t.sx * ty;  k.T[8] = k.t.cx * ty;
    }
    __device__ static void next_state(const ModelParams& mp, const float (&x)[NX], const Core& k,
                                      float (&xn)[NX]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float thd = k.T[3 * i] * k.wn[0] + k.T[3 * i + 1] * k.wn[1] + k.T[3 * i + 2] * k.wn[2];
            xn[i] = x[i] + mp.dt * k.vn[i];
            xn[3 + i] = x[3 + i] + mp.dt * thd;
            xn[6 + i] = k.vn[i];
        }
        xn[9] = k.wn[2]; xn[10] = k.wn[1]; xn[11] = k.wn[0];
    }
    __device__ static void step(const ModelParams& mp, const float (&x)[NX], const float (&u)[NU],
                                const float* p, float (&xn)[NX]) {
        Core k;
        core(mp, x, u, p, k);
        next_state(mp, x, k, xn);
    }

    // column of the 12x12 Jacobian from its (wx,wy,wz)-row part `dw` (d w+ / d var) and the extra
    // direct terms: rows r get `er`, rows th get eth + dt*T*dw, rows rdot get ev, rows w get dw reversed
    __device__ static void assemble(const ModelParams& mp, const Core& k, const float (&dw)[3],
                                    const float (&er)[3], const float (&eth)[3], const float (&ev)[3],
                                    float (&col)[NX]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            col[i] = er[i];
            col[3 + i] = eth[i] + mp.dt * (k.T[3 * i] * dw[0] + k.T[3 * i + 1] * dw[1] + k.T[3 * i + 2] * dw[2]);
            col[6 + i] = ev[i];
        }
        col[9] = dw[2]; col[10] = dw[1]; col[11] = dw[0];
    }

    template <class EA, class EB>
    __device__ static void linearize(const ModelParams& mp, const float (&x)[NX], const float (&u)[NU],
                                     const float* p, float (&xn)[NX], EA&& emitA, EB&& emitB) {
        Core k;
        core(mp, x, u, p, k);
        next_state(mp, x, k, xn);
        const float dt = mp.dt;
        const float iI[3] = {dt / mp.ixx, dt / mp.iyy, dt / mp.izz};  // dt * I^-1
        const float z3[3] = {0.f, 0.f, 0.f};
        const Trig& t = k.t;
        float col[NX];

        // ---- columns r_j : d tau / d r = [F]x ;  dw = dt I^-1 R' [F]x e_j
        {
            const float Fx[9] = {0.f, -k.F[2], k.F[1], k.F[2], 0.f, -k.F[0], -k.F[1], k.F[0], 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float v[3] = {Fx[j], Fx[3 + j], Fx[6 + j]};
                float dw[3];
                tmul(k.R, v, dw);
                dw[0] *= iI[0]; dw[1] *= iI[1]; dw[2] *= iI[2];
                float er[3] = {0.f, 0.f, 0.f};
                er[j] = 1.f;
                assemble(mp, k, dw, er, z3, z3, col);
                emitA(j, col);
            }
        }
        // ---- columns th_j (yaw, pitch, roll): dw = dt I^-1 (dR/dth_j)' tau ; plus d(T w+)/dth_j
        {
            float dR[9], dw[3];
            const float sw = t.sx * k.wn[1] + t.cx * k.wn[2], cw = t.cx * k.wn[1] - t.sx * k.wn[2];
            const float icy = 1.0f / t.cy, ty = t.sy * icy;
            // yaw: dRz Ry Rx
            dR[0] = -t.sz * t.cy; dR[1] = -t.sz * t.sy * t.sx - t.cz * t.cx; dR[2] = -t.sz * t.sy * t.cx + t.cz * t.sx;
            dR[3] = t.cz * t.cy;  dR[4] = t.cz * t.sy * t.sx - t.sz * t.cx;  dR[5] = t.cz * t.sy * t.cx + t.sz * t.sx;
            dR[6] = 0.f; dR[7] = 0.f; dR[8] = 0.f;
            tmul(dR, k.tau, dw);
            dw[0] *= iI[0]; dw[1] *= iI[1]; dw[2] *= iI[2];
            { const float eth[3] = {1.f, 0.f, 0.f}; assemble(mp, k, dw, z3, eth, z3, col); }
            emitA(3, col);
            // pitch: Rz dRy Rx
            dR[0] = -t.cz * t.sy; dR[1] = t.cz * t.cy * t.sx; dR[2] = t.cz * t.cy * t.cx;
            dR[3] = -t.sz * t.sy; dR[4] = t.sz * t.cy * t.sx; dR[5] = t.sz * t.cy * t.cx;
            dR[6] = -t.cy;        dR[7] = -t.sy * t.sx;       dR[8] = -t.sy * t.cx;
            tmul(dR, k.tau, dw);
            dw[0] *= iI[0]; dw[1] *= iI[1]; dw[2] *= iI[2];
            { const float eth[3] = {dt * ty * icy * sw, 1.f, dt * icy * icy * sw};
              assemble(mp, k, dw, z3, eth, z3, col); }
            emitA(4, col);
            // roll: Rz Ry dRx
            dR[0] = 0.f; dR[1] = t.cz * t.sy * t.cx + t.sz * t.sx;  dR[2] = -t.cz * t.sy * t.sx + t.sz * t.cx;
            dR[3] = 0.f; dR[4] = t.sz * t.sy * t.cx - t.cz * t.sx;  dR[5] = -t.sz * t.sy * t.sx - t.cz * t.cx;
            dR[6] = 0.f; dR[7] = t.cy * t.cx;                        dR[8] = -t.cy * t.sx;
            tmul(dR, k.tau, dw);
            dw[0] *= iI[0]; dw[1] *= iI[1]; dw[2] *= iI[2];
            { const float eth[3] = {dt * icy * cw, -dt * sw, 1.f + dt * ty * cw};
              assemble(mp, k, dw, z3, eth, z3, col); }
            emitA(5, col);
        }
        // ---- columns rdot_j
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float er[3] = {0.f, 0.f, 0.f}, ev[3] = {0.f, 0.f, 0.f};
            er[j] = dt; ev[j] = 1.f;
            assemble(mp, k, z3, er, z3, ev, col);
            emitA(6 + j, col);
        }
        // ---- columns of the body rates; state slot 9+j holds w[2-j]
        {
            const float I[3] = {mp.ixx, mp.iyy, mp.izz};
            // d(w x I w)/dw, row-major
            const float Gy[9] = {0.f, (I[2] - I[1]) * k.w[2], (I[2] - I[1]) * k.w[1],
                                 (I[0] - I[2]) * k.w[2], 0.f, (I[0] - I[2]) * k.w[0],
                                 (I[1] - I[0]) * k.w[1], (I[1] - I[0]) * k.w[0], 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int m = 2 - j;  // w index of this slot
                float dw[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) dw[i] = (i == m ? 1.f : 0.f) - iI[i] * Gy[3 * i + m];
                assemble(mp, k, dw, z3, z3, z3, col);
                emitA(9 + j, col);
            }
        }
        // ---- B columns: force component j of foot i
        const float im = 1.0f / mp.mass;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float c = p[i];
            const float a[3] = {p[4 + 3 * i] - x[0], p[5 + 3 * i] - x[1], p[6 + 3 * i] - x[2]};
            const float ax[9] = {0.f, -a[2], a[1], a[2], 0.f, -a[0], -a[1], a[0], 0.f};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float v[3] = {ax[j], ax[3 + j], ax[6 + j]};
                float dw[3];
                tmul(k.R, v, dw);
                dw[0] *= c * iI[0]; dw[1] *= c * iI[1]; dw[2] *= c * iI[2];
                float er[3] = {0.f, 0.f, 0.f}, ev[3] = {0.f, 0.f, 0.f};
                er[j] = dt * dt * c * im; ev[j] = dt * c * im;
                assemble(mp, k, dw, er, z3, ev, col);
                emitB(3 * i + j, col);
            }
        }
    }

    // friction pyramid faces jj of foot f: +fx, -fx, +fy, -fy  minus mu fz  <= 0, as row 4 jj + f: in the
    // 16-row tile of the barrier product lane quad q holds rows 4q..4q+3 and MFMA step i contracts register i
    // of every quad -- with this order step i is "foot i, all four faces", so the steps of swing feet (all
    // rows inactive: exact zeros) can be left out where the contact pattern is a compile-time constant
    __host__ __device__ static constexpr int row_of(int f, int jj) { return 4 * jj + f; }
    __device__ static float G(const ModelParams& mp, int j, int c) {
        const int f = j & 3, jj = j >> 2;
        const int d = c - 3 * f;
        return (d == (jj >> 1)) ? ((jj & 1) ? -1.f : 1.f) : (d == 2 ? -mp.mu : 0.f);
    }
    __device__ static float h(const ModelParams&, int) { return 0.f; }
    __device__ static unsigned active_mask(const ModelParams&, const float* p) {
        unsigned m = 0;
#pragma unroll
        for (int f = 0; f < 4; ++f) m |= (p[f] > 0.5f) ? (0x1111u << f) : 0u;
        return m;
    }
    // inputs that couple with others in Huu: the forces of stance feet.  A swing foot's force has a
    // zero column in B and no active constraint row, so its Huu row/column is exactly diagonal.
    __device__ static unsigned input_mask(const ModelParams&, const float* p) {
        unsigned m = 0;
#pragma unroll
        for (int f = 0; f < 4; ++f) m |= (p[f] > 0.5f) ? (0x7u << (3 * f)) : 0u;
        return m;
    }
    // Coupling masks that get a fully static stage body: every contact pattern of the four feet
    // (index = stance flags as a 4-bit number), so that each gait of the reference's gait table
    // (contact_planner.py:45-118: trot, pace, bound, crawl, jump ...) runs the straight-line
    // elimination; measured, the run-time-mask fallback is 25 % slower for the same pattern.
    static constexpr int N_STATIC_MASKS = 16;
    __host__ __device__ static constexpr unsigned static_mask(int i) {
        unsigned m = 0;
        for (int f = 0; f < 4; ++f) m |= ((i >> f) & 1) ? (0x7u << (3 * f)) : 0u;
        return m;
    }
    // The short dispatch list of the default kernel: a trot (the two diagonal pairs), four-foot stance,
    // flight.  Other patterns take the run-time fallback there; gaits that show them select the kernel
    // with all sixteen variants (nmpc_set_contact_patterns).
    __host__ __device__ static constexpr int common_variant(int i) { return i == 0 ? 9 : i == 1 ? 6 : i == 2 ? 15 : i == 3 ? 0 : -1; }
    // MFMA steps of the barrier product that can hold active rows under an input mask: step f = foot f
    // (row_of), active iff the foot is in stance
    __host__ __device__ static constexpr unsigned barrier_steps(unsigned mask) {
        unsigned m = 0;
        for (int f = 0; f < 4; ++f) m |= ((mask >> (3 * f)) & 1u) << f;
        return m;
    }
    // index of the static variant of a mask, -1 if there is none
    __host__ __device__ static constexpr int static_index(unsigned mask) {
        int i = 0;
        for (int f = 0; f < 4; ++f) i |= ((mask >> (3 * f)) & 1u) << f;
        return static_mask(i) == mask ? i : -1;
    }
    __device__ static void gdot(const ModelParams& mp, const float (&v)[NU], float (&o)[NG]) {
#pragma clang fp contract(off)   // same rounding in every caller: the QP kernel's variants are compared bit for bit
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            o[row_of(f, 0)] = __builtin_fmaf(-mp.mu, v[3 * f + 2], v[3 * f]);
            o[row_of(f, 1)] = __builtin_fmaf(-mp.mu, v[3 * f + 2], -v[3 * f]);
            o[row_of(f, 2)] = __builtin_fmaf(-mp.mu, v[3 * f + 2], v[3 * f + 1]);
            o[row_of(f, 3)] = __builtin_fmaf(-mp.mu, v[3 * f + 2], -v[3 * f + 1]);
        }
    }
};

}  // namespace nmpc

#pragma clang fp contract(fast)
