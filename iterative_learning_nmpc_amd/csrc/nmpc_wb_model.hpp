// nmpc_wb_model.hpp -- the declared whole-body kinodynamic quadruped (NMPC_MODEL_WHOLEBODY, BASELINE configs[2]).
//
// Layouts are the reference's (mpc_controller/utils/solver.py:88-92,185-187,405-418):
//   x = [q(18) = r(3), (yaw,pitch,roll), joints(12) | v(18) = qdot | h(6) centroidal momentum]
//   u = [a(18) = vdot | f[4][3] world forces, feet FL,FR,RL,RR]
//   p = [c(4) active, peak(4), plane_point(4x3)]                    (solver.py:212-252)
// The symbolic model lives in the absent contact_tamp; the mathematics declared in DESIGN.md 3.2 (restated
// by the test oracle as its model 2) is evaluated here lane-locally, one thread per (problem, node):
//   semi-implicit Euler  v+ = v + dt a, q+ = q + dt v+, h_lin+ = h_lin + dt (sum c_i f_i + m g),
//   h_ang+ = h_ang + dt sum c_i (p_i(q) - r) x f_i ; legs = hip abduction (x), thigh (y), knee (y), point foot.
// Residuals of the Gauss-Newton cost (order of W / yref, ny = 90; terminal ny_e = 66 without acc, f_reg):
//   base(12) joint(24) acc(12) swing(4) f_reg(12) contact(12) consist(6) pos(8)
// pos = world (x, y) of each foot against the planned location: the reference's pos_cost, weighted by W_foot_displacement in
// its contact-restricted mode and by 0 otherwise (solver.py:128-137,272-273)
// (dynamics.py:121-134, solver.py:108-141,170-177; contact = Baumgarte-stabilised stance constraint with
//  p_gain = W_foot_pos_constr_stab, consist = h - A_g(q) v, both as quadratic penalties).
#pragma once
#include "nmpc_models.hpp"

namespace nmpc {
namespace wb {

constexpr int NX = 42, NU = 30, NP = 20, NG = 16, NY = 90, NYE = 66;
constexpr int WQ = 0, WV = 18, WH = 36, WA = 0, WF = 18;
constexpr int RY_BASE = 0, RY_JOINT = 12, RY_ACC = 36, RY_SWING = 48, RY_FREG = 52, RY_CNT = 64, RY_CONS = 76, RY_POS = 82;
constexpr int RE_BASE = 0, RE_JOINT = 12, RE_SWING = 36, RE_CNT = 40, RE_CONS = 52, RE_POS = 58;
constexpr int NJR = 30;    // dense residual rows: contact 12 (rows 0..11), swing 4 (12..15), consist 6 (16..21), foot placement 8 (22..29)

struct M3 { float m[9]; };
__device__ __forceinline__ M3 mul(const M3& a, const M3& b) {
    M3 c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return c;
}
__device__ __forceinline__ M3 add(const M3& a, const M3& b) {
    M3 c;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] + b.m[i];
    return c;
}
__device__ __forceinline__ M3 scale(const M3& a, float s) {
    M3 c;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] * s;
    return c;
}
__device__ __forceinline__ void mv(const M3& a, const float (&v)[3], float (&o)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = a.m[3 * i] * v[0] + a.m[3 * i + 1] * v[1] + a.m[3 * i + 2] * v[2];
}
__device__ __forceinline__ void cross(const float (&a)[3], const float (&b)[3], float (&o)[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
// rotation about `AXIS` with its first and second derivative with respect to the angle
template <int AXIS>
__device__ __forceinline__ void rot_axis(float ang, M3& M, M3& M1, M3& M2) {
    float s, c;
    sincosf(ang, &s, &c);
#pragma unroll
    for (int i = 0; i < 9; ++i) M.m[i] = M1.m[i] = M2.m[i] = 0.0f;
    constexpr int i0 = (AXIS + 1) % 3, i1 = (AXIS + 2) % 3;
    M.m[4 * AXIS] = 1.0f;
    M.m[3 * i0 + i0] = c;   M.m[3 * i0 + i1] = -s;  M.m[3 * i1 + i0] = s;   M.m[3 * i1 + i1] = c;
    M1.m[3 * i0 + i0] = -s; M1.m[3 * i0 + i1] = -c; M1.m[3 * i1 + i0] = c;  M1.m[3 * i1 + i1] = -s;
    M2.m[3 * i0 + i0] = -c; M2.m[3 * i0 + i1] = s;  M2.m[3 * i1 + i0] = -s; M2.m[3 * i1 + i1] = -c;
}

// base orientation R = Rz(yaw) Ry(pitch) Rx(roll) (pin.rpy.rpyToMatrix, mpc.py:205), dR/dtheta_a, and their
// time derivatives along the Euler rates (RATES = false: rates taken as zero, Rd/Rad not filled)
struct BaseRot { M3 R, Ra[3], Rd, Rad[3]; };
template <bool RATES>
__device__ __forceinline__ void base_rotation(const float (&th)[3], const float (&thd)[3], BaseRot& o) {
    M3 M[3], M1[3], M2[3];
    rot_axis<2>(th[0], M[0], M1[0], M2[0]);
    rot_axis<1>(th[1], M[1], M1[1], M2[1]);
    rot_axis<0>(th[2], M[2], M1[2], M2[2]);
    const M3 YX = mul(M[1], M[2]);
    o.R = mul(M[0], YX);
    o.Ra[0] = mul(M1[0], YX);
    o.Ra[1] = mul(M[0], mul(M1[1], M[2]));
    o.Ra[2] = mul(M[0], mul(M[1], M1[2]));
    if constexpr (RATES) {
        M3 Md[3], M1d[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { Md[a] = scale(M1[a], thd[a]); M1d[a] = scale(M2[a], thd[a]); }
        auto prod_rule = [&](const M3& z, const M3& zd, const M3& y, const M3& yd, const M3& x, const M3& xd) {
            return add(add(mul(zd, mul(y, x)), mul(z, mul(yd, x))), mul(z, mul(y, xd)));
        };
        o.Rd = prod_rule(M[0], Md[0], M[1], Md[1], M[2], Md[2]);
        o.Rad[0] = prod_rule(M1[0], M1d[0], M[1], Md[1], M[2], Md[2]);
        o.Rad[1] = prod_rule(M[0], Md[0], M1[1], M1d[1], M[2], Md[2]);
        o.Rad[2] = prod_rule(M[0], Md[0], M[1], Md[1], M1[2], M1d[2]);
    }
}

// one leg in the base frame: foot position b, Jacobian Jb (row-major 3x3) and, with RATES, its time derivative
// and the foot velocity bd = Jb ql_dot
struct Leg { float b[3]; M3 Jb, Jbd; float bd[3]; };
template <bool RATES>
__device__ __forceinline__ void leg_kin(const ModelParams& mp, int f, const float (&ql)[3], const float (&wl)[3], Leg& o) {
    const float sgx = (f < 2) ? 1.0f : -1.0f, sgy = (f & 1) ? -1.0f : 1.0f;
    const float d = sgy * mp.lhip, l1 = mp.l1, l2 = mp.l2;
    float s1, c1, s2, c2, s23, c23;
    sincosf(ql[0], &s1, &c1);
    sincosf(ql[1], &s2, &c2);
    sincosf(ql[1] + ql[2], &s23, &c23);
    const float vx = -l1 * s2 - l2 * s23, vz = -l1 * c2 - l2 * c23, vx3 = -l2 * s23, vz3 = -l2 * c23;
    o.b[0] = sgx * mp.hipx + vx;
    o.b[1] = sgy * mp.hipy + d * c1 - vz * s1;
    o.b[2] = d * s1 + vz * c1;
    o.Jb.m[0] = 0.0f;               o.Jb.m[1] = vz;       o.Jb.m[2] = vz3;
    o.Jb.m[3] = -d * s1 - vz * c1;  o.Jb.m[4] = vx * s1;  o.Jb.m[5] = vx3 * s1;
    o.Jb.m[6] = d * c1 - vz * s1;   o.Jb.m[7] = -vx * c1; o.Jb.m[8] = -vx3 * c1;
    if constexpr (RATES) {
        const float w1 = wl[0], w2 = wl[1], w3 = wl[2], w23 = w2 + w3;
        const float s1d = c1 * w1, c1d = -s1 * w1;
        const float vxd = vz * w2 + vz3 * w3, vzd = -vx * w2 - vx3 * w3, vx3d = vz3 * w23, vz3d = -vx3 * w23;
        o.Jbd.m[0] = 0.0f;                               o.Jbd.m[1] = vzd;                  o.Jbd.m[2] = vz3d;
        o.Jbd.m[3] = -d * s1d - vzd * c1 - vz * c1d;     o.Jbd.m[4] = vxd * s1 + vx * s1d;  o.Jbd.m[5] = vx3d * s1 + vx3 * s1d;
        o.Jbd.m[6] = d * c1d - vzd * s1 - vz * s1d;      o.Jbd.m[7] = -vxd * c1 - vx * c1d; o.Jbd.m[8] = -vx3d * c1 - vx3 * c1d;
        mv(o.Jb, wl, o.bd);
    }
}

// column of x (or of v, +WV) that slot c of xi_f = [r, theta, ql_f] stands for
__host__ __device__ constexpr int xi_col(int f, int c) { return c < 6 ? c : 6 + 3 * f + (c - 6); }

// friction pyramid rows of foot f, faces +fx, -fx, +fy, -fy  minus mu fz <= 0: row 4 f + face (the oracle's order)
__device__ __forceinline__ void gdot(const ModelParams& mp, const float (&fv)[12], float (&o)[NG]) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const float mz = mp.mu * fv[3 * f + 2];
        o[4 * f + 0] = fv[3 * f] - mz;
        o[4 * f + 1] = -fv[3 * f] - mz;
        o[4 * f + 2] = fv[3 * f + 1] - mz;
        o[4 * f + 3] = -fv[3 * f + 1] - mz;
    }
}
__device__ __forceinline__ unsigned active_mask(const float* p) {
    unsigned m = 0;
#pragma unroll
    for (int f = 0; f < 4; ++f) m |= (p[f] > 0.5f) ? (0xFu << (4 * f)) : 0u;
    return m;
}

}  // namespace wb
}  // namespace nmpc
