/*
 * oracle/nmpc_oracle.c  --  CPU restatement of the batched NMPC solve path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import, link
 * or call this file; it is used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference's solve (mpc_controller/utils/solver.py:396-403
 * -> contact_tamp.AcadosSolverHelper.solve -> acados SQP / HPIPM / BLASFEO) lives
 * in an un-vendored submodule (.gitmodules:1-3, directory empty) and in acados
 * (git HEAD, unpinned; README.md:26-44).  None of it can be compiled or imported
 * in this image and the reference holds no test or golden vector for the solve
 * (SURVEY.md section 4, 8c).  This file therefore restates the *published*
 * algorithm family the reference calls (multiple-shooting Gauss-Newton SQP with a
 * Riccati-based primal-dual interior point QP, acados/HPIPM) with every free
 * choice DECLARED below; it is anchored to the reference only through
 *   - the call sites that fix dimensions, layouts, weights, iteration policy:
 *       mpc_controller/utils/solver.py:153-429, mpc_controller/mpc.py:317-369,464-473
 *       mpc_controller/config/quadruped/mpc_cost.py:26-72, mpc_opt.py:8-27
 *   - golden vectors of the host-side helpers that feed it (tests/golden/).
 *
 * Build: see oracle/Makefile -- compiled twice, -DORACLE_F32 (float) and default
 * (double):  liboracle_f32.so / liboracle_f64.so.
 *
 * ---------------------------------------------------------------------------
 * DECLARED ALGORITHM (shared, as mathematics, with DESIGN.md section 3)
 *
 * OCP    min  sum_{k<N} 1/2 |y_k - yref_k|^2_W + 1/2 |x_N - yref_e|^2_We
 *        s.t. x_0 = xhat,  x_{k+1} = phi(x_k,u_k,p_k),  G(p_k) u_k <= h(p_k)
 *        y = [x;u]  (ny = nx+nu), W diagonal            [solver.py:108-141: 1-D weights]
 *
 * SQP iteration (max_iter = 1 steady state, 15 on first solve: mpc.py:464-473)
 *   1 linearise   A_k, B_k analytic, d_k = phi(x_k,u_k) - x_{k+1},
 *                 q_k = Wx.(x_k - xref_k), r_k = Wu.(u_k - uref_k),
 *                 Gauss-Newton Hessian  Q = diag(Wx)+reg, R = diag(Wu)+reg,
 *                 terminal Q_N = diag(We)+reg_e      [mpc_cost.py:71-72 reg_eps]
 *   2 QP          fixed-iteration primal-dual interior point (n_ipm, default 6 =
 *                 max_qp_iter, mpc_opt.py:27) on slacks s, multipliers lam of
 *                 G du + (G u - h) + s = 0; every iteration solves the barrier-
 *                 modified LQ problem by one Riccati sweep:
 *                   D = lam/s, tau = max(sigma*mean(s.lam), tau_min),
 *                   Rt = R + G'DG,  rt = r + G'(tau/s + lam + D c)
 *                   (dx+,du+) = riccati(...)
 *                   ds = -(G du+ + c) - s ; dlam = tau/s - lam - D ds
 *                   a_p, a_d = fraction-to-boundary(gamma)
 *                   du += a_p (du+ - du); dx likewise; s += a_p ds; lam += a_d dlam
 *                 cold start each SQP iteration: s = max(-c, s_min), lam = mu0/s.
 *   3 step        alpha = 1 (acados default, no globalisation) or backtracking on
 *                 an l1 merit (opt.line_search).
 *
 * Riccati (backward, k = N-1..0), LDL' without pivoting on Huu:
 *   Pd = P d + p;  Huu = Rt + B'PB; Hux = B'PA; hu = rt + B'Pd
 *   Hxx = Q + A'PA; hx = q + A'Pd
 *   K = -Huu^-1 Hux; kff = -Huu^-1 hu;  P = Hxx + Hux'K;  p = hx + Hux'kff
 * forward: dx_0 = xhat - x_0; du = K dx + kff; dx+ = A dx + B du + d.
 *
 * MODELS
 *  model 0  double integrator: x=[p(2),v(2)], u=a(2), exact ZOH,
 *           optional box |u_i| <= umax  (ng = 4).
 *  model 1  centroidal single-rigid-body quadruped, nx=12, nu=12:
 *           x = [r(3), (yaw,pitch,roll), rdot(3), (wz,wy,wx) body rates]
 *               -- slot order of the reference's 12-vector base_cost / base_ref
 *                  (dynamics.py:121-124, mpc.py:217-231)
 *           u = f[4][3] world-frame forces, foot order FL,FR,RL,RR (solver.py:417-421)
 *           p = [c(4) contact flags (contact_planner.py:121-134), foot pos(4x3)]
 *           semi-implicit Euler, dt = T/N:
 *             rdot+ = rdot + dt (sum c_i f_i / m + g)
 *             w+    = w + dt I^-1 ( R' sum c_i (p_i - r) x f_i - w x I w )
 *             r+    = r + dt rdot+
 *             th+   = th + dt T(th) w+      T = transform.py:72-78 matrix
 *           R = Rz(yaw) Ry(pitch) Rx(roll)  (pin.rpy.rpyToMatrix, mpc.py:205)
 *           friction pyramid per stance foot, mu = 0.8 (solver.py:38):
 *             +-fx - mu fz <= 0, +-fy - mu fz <= 0   (ng = 16; rows of swing feet inactive)
 *  model 2  whole-body kinodynamic quadruped (BASELINE configs[2]), nx=42, nu=30.
 *           Layouts are the reference's (solver.py:88-92,185-187,405-418):
 *             x = [q(18) = r(3), (yaw,pitch,roll), joints(12) | v(18) = qdot | h(6) centroidal momentum]
 *             u = [a(18) = vdot | f[4][3] world forces, feet FL,FR,RL,RR]
 *           (the reference's velocity slots 3..5 are the Euler-angle rates, dynamics.py:69, so qdot = v).
 *           p = [c(4) active, peak(4), plane_point(4x3)]  (solver.py:212-252; plane normal = e_z, :216)
 *           The symbolic model itself lives in the absent contact_tamp; DECLARED here:
 *           semi-implicit Euler, dt = T/N:  v+ = v + dt a,  q+ = q + dt v+,
 *             h_lin+ = h_lin + dt (sum c_i f_i + m g),  h_ang+ = h_ang + dt sum c_i (p_i(q) - r) x f_i
 *           legs: hip abduction (x) at (+-hipx, +-hipy, 0), thigh (y) at (0, +-lhip, 0), knee (y) at
 *           (0,0,-l1), point foot at (0,0,-l2) -- the declared tree of workloads.quadruped_tree();
 *           single-rigid-body inertia for the momentum map: A_g(q) v = [m rdot; R I_b E(th) thdot],
 *           E = euler_derivative_to_local_angular (transform.py:80-86), COM at the base origin.
 *           Gauss-Newton least-squares cost, residuals and weights in this order (ny = 90):
 *             base(12)   [q[0:6], v[0:6]] - base_ref          W_base      dynamics.py:121-124
 *             joint(24)  [q[6:], v[6:]] - [joint_ref, 0]      W_joint     dynamics.py:126, solver.py:175-177
 *             acc(12)    a[6:]                                W_acc       dynamics.py:129
 *             swing(4)   peak_i z_foot_i(q) - step_height     W_swing     dynamics.py:131-134, solver.py:170
 *             f_reg(12)  f                                    W_cnt_f_reg solver.py:128-130
 *             contact(12) c_i (J_i(q) v + p_gain e_z (z_foot_i - plane_point_i,z))   [decl weight]
 *                        Baumgarte-stabilised stance constraint, p_gain = W_foot_pos_constr_stab = 50
 *                        (solver.py:219, mpc_cost.py:60), entering as a quadratic penalty: the
 *                        Riccati/IPM core of this build carries input inequalities only
 *             consist(6) h - A_g(q) v                                               [decl weight]
 *           terminal (ny_e = 66): base (W_e_base), joint (W_e_joint), swing, contact, consist, pos.
 *           friction pyramid on f as in model 1 (ng = 16).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORACLE_F32
typedef float real;
#define RSQRT(x) (1.0f / sqrtf(x))
#define SIN sinf
#define COS cosf
#define FABS fabsf
#else
typedef double real;
#define SIN sin
#define COS cos
#define FABS fabs
#endif

#define MAXN 48 /* max nx, nu, ng handled by the oracle's stack buffers */

/* model parameter vector mp[16]: dt, mass, Ixx, Iyy, Izz, gz, mu, umax, then (model 2) p_gain and the leg
 * geometry hipx, hipy, lhip, l1, l2; two reserved */
enum { MP_DT = 0, MP_MASS, MP_IXX, MP_IYY, MP_IZZ, MP_GZ, MP_MU, MP_UMAX,
       MP_PGAIN, MP_HIPX, MP_HIPY, MP_LHIP, MP_L1, MP_L2, MP_RES0, MP_RES1, MP_COUNT };
/* option vector opt[13] */
enum {
    OP_MAX_SQP = 0, OP_N_IPM, OP_NLP_TOL, OP_REG, OP_REG_E, OP_MU0, OP_SIGMA, OP_SMIN,
    OP_GAMMA, OP_LINE_SEARCH, OP_RHO, OP_YREF_PER_STAGE, OP_TAU_MIN,
    OP_IPM_WARM, OP_WS_SFLOOR, OP_WS_LFLOOR, OP_WS_SHIFT, OP_WS_HAVE, OP_COUNT
};

int oracle_real_size(void) { return (int)sizeof(real); }

int oracle_dims(int model_id, int *nx, int *nu, int *np, int *ng) {
    if (model_id == 0) { *nx = 4; *nu = 2; *np = 0; *ng = 4; return 0; }
    if (model_id == 1) { *nx = 12; *nu = 12; *np = 16; *ng = 16; return 0; }
    if (model_id == 2) { *nx = 42; *nu = 30; *np = 20; *ng = 16; return 0; }
    return -1;
}

/* number of cost residuals of a stage / of the terminal node (= length of W, yref / W_e, yref_e) */
int oracle_output_dims(int model_id, int *ny, int *nye) {
    int nx, nu, np, ng;
    if (oracle_dims(model_id, &nx, &nu, &np, &ng)) return -1;
    if (model_id == 2) { *ny = 90; *nye = 66; return 0; }
    *ny = nx + nu; *nye = nx;
    return 0;
}

/* ------------------------------------------------------------------ 3x3 helpers */
static void m3_mul(const real *a, const real *b, real *c) { /* c = a b */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            real s = 0;
            for (int k = 0; k < 3; k++) s += a[3 * i + k] * b[3 * k + j];
            c[3 * i + j] = s;
        }
}
static void m3t_vec(const real *a, const real *v, real *o) { /* o = a' v */
    for (int i = 0; i < 3; i++) o[i] = a[0 + i] * v[0] + a[3 + i] * v[1] + a[6 + i] * v[2];
}
static void cross3(const real *a, const real *b, real *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* ------------------------------------------------------------------ model 0 */
static void dyn_double_integrator(const real *mp, const real *x, const real *u, real *xn, real *A,
                                  real *B) {
    const real dt = mp[MP_DT];
    const int nx = 4, nu = 2;
    for (int i = 0; i < 2; i++) {
        xn[i] = x[i] + dt * x[2 + i] + (real)0.5 * dt * dt * u[i];
        xn[2 + i] = x[2 + i] + dt * u[i];
    }
    if (A) {
        memset(A, 0, sizeof(real) * nx * nx);
        memset(B, 0, sizeof(real) * nx * nu);
        for (int i = 0; i < 4; i++) A[i * nx + i] = 1;
        for (int i = 0; i < 2; i++) {
            A[i * nx + 2 + i] = dt;
            B[i * nu + i] = (real)0.5 * dt * dt;
            B[(2 + i) * nu + i] = dt;
        }
    }
}

/* ------------------------------------------------------------------ model 1 */
/* index helpers: state slots */
enum { XR = 0, XTH = 3, XV = 6, XW = 9 };

static void dyn_centroidal(const real *mp, const real *x, const real *u, const real *p, real *xn,
                           real *A, real *B) {
    const int nx = 12, nu = 12;
    const real dt = mp[MP_DT], m = mp[MP_MASS];
    const real I[3] = {mp[MP_IXX], mp[MP_IYY], mp[MP_IZZ]};
    const real gz = mp[MP_GZ];
    const real yaw = x[XTH + 0], pit = x[XTH + 1], rol = x[XTH + 2];
    /* body rates stored (wz,wy,wx); work with w = (wx,wy,wz) */
    const real w[3] = {x[XW + 2], x[XW + 1], x[XW + 0]};
    const real cz = COS(yaw), sz = SIN(yaw), cy = COS(pit), sy = SIN(pit), cx = COS(rol),
               sx = SIN(rol);
    const real Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    const real Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
    const real Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    const real dRz[9] = {-sz, -cz, 0, cz, -sz, 0, 0, 0, 0};
    const real dRy[9] = {-sy, 0, cy, 0, 0, 0, -cy, 0, -sy};
    const real dRx[9] = {0, 0, 0, 0, -sx, -cx, 0, cx, -sx};
    real RyRx[9], R[9], tmp[9], dR_yaw[9], dR_pit[9], dR_rol[9];
    m3_mul(Ry, Rx, RyRx);
    m3_mul(Rz, RyRx, R);
    m3_mul(dRz, RyRx, dR_yaw);
    m3_mul(dRy, Rx, tmp);
    m3_mul(Rz, tmp, dR_pit);
    m3_mul(Ry, dRx, tmp);
    m3_mul(Rz, tmp, dR_rol);

    /* net force and world torque about the COM */
    real F[3] = {0, 0, 0}, tau[3] = {0, 0, 0};
    for (int i = 0; i < 4; i++) {
        const real c = p[i];
        const real *f = u + 3 * i;
        real arm[3] = {p[4 + 3 * i + 0] - x[0], p[4 + 3 * i + 1] - x[1], p[4 + 3 * i + 2] - x[2]};
        real t[3];
        cross3(arm, f, t);
        for (int j = 0; j < 3; j++) {
            F[j] += c * f[j];
            tau[j] += c * t[j];
        }
    }
    real taub[3];
    m3t_vec(R, tau, taub);
    /* gyroscopic term w x (I w) */
    const real gyro[3] = {(I[2] - I[1]) * w[1] * w[2], (I[0] - I[2]) * w[2] * w[0],
                          (I[1] - I[0]) * w[0] * w[1]};
    real vn[3], wn[3];
    for (int j = 0; j < 3; j++) {
        vn[j] = x[XV + j] + dt * (F[j] / m + (j == 2 ? gz : 0));
        wn[j] = w[j] + dt * (taub[j] - gyro[j]) / I[j];
    }
    /* euler-rate map  thdot = T(th) w+   (transform.py:72-78; rows yaw,pitch,roll) */
    const real ty = sy / cy;
    const real T[9] = {0, sx / cy, cx / cy, 0, cx, -sx, 1, sx * ty, cx * ty};
    real thd[3];
    for (int i = 0; i < 3; i++) thd[i] = T[3 * i] * wn[0] + T[3 * i + 1] * wn[1] + T[3 * i + 2] * wn[2];
    for (int j = 0; j < 3; j++) {
        xn[XR + j] = x[XR + j] + dt * vn[j];
        xn[XTH + j] = x[XTH + j] + dt * thd[j];
        xn[XV + j] = vn[j];
    }
    xn[XW + 0] = wn[2];
    xn[XW + 1] = wn[1];
    xn[XW + 2] = wn[0];
    if (!A) return;

    /* ---- Jacobians.  First d wn / d(r, th, w, f_i) in (wx,wy,wz) row order. */
    real dwn_dr[9], dwn_dth[9], dwn_dw[9], dwn_df[4][9];
    /* d tau / d r = [F]x  (arm = p - r) */
    const real Fx[9] = {0, -F[2], F[1], F[2], 0, -F[0], -F[1], F[0], 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            real s = 0; /* (R' Fx)_ij */
            for (int k = 0; k < 3; k++) s += R[3 * k + i] * Fx[3 * k + j];
            dwn_dr[3 * i + j] = dt * s / I[i];
        }
    {
        real c0[3], c1[3], c2[3];
        m3t_vec(dR_yaw, tau, c0);
        m3t_vec(dR_pit, tau, c1);
        m3t_vec(dR_rol, tau, c2);
        for (int i = 0; i < 3; i++) {
            dwn_dth[3 * i + 0] = dt * c0[i] / I[i];
            dwn_dth[3 * i + 1] = dt * c1[i] / I[i];
            dwn_dth[3 * i + 2] = dt * c2[i] / I[i];
        }
    }
    {
        const real G[9] = {0, (I[2] - I[1]) * w[2], (I[2] - I[1]) * w[1],
                           (I[0] - I[2]) * w[2], 0, (I[0] - I[2]) * w[0],
                           (I[1] - I[0]) * w[1], (I[1] - I[0]) * w[0], 0};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                dwn_dw[3 * i + j] = (i == j ? 1 : 0) - dt * G[3 * i + j] / I[i];
    }
    for (int f = 0; f < 4; f++) {
        const real c = p[f];
        const real a[3] = {p[4 + 3 * f + 0] - x[0], p[4 + 3 * f + 1] - x[1], p[4 + 3 * f + 2] - x[2]};
        const real ax[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                real s = 0;
                for (int k = 0; k < 3; k++) s += R[3 * k + i] * ax[3 * k + j];
                dwn_df[f][3 * i + j] = dt * c * s / I[i];
            }
    }
    /* partial of T(th) w+ w.r.t. th with w+ held fixed (columns yaw,pitch,roll) */
    const real sw = sx * wn[1] + cx * wn[2], cw = cx * wn[1] - sx * wn[2];
    const real dTw[9] = {0, sy / (cy * cy) * sw, cw / cy,
                         0, 0, -sw,
                         0, sw / (cy * cy), ty * cw};

    memset(A, 0, sizeof(real) * nx * nx);
    memset(B, 0, sizeof(real) * nx * nu);
    /* map (wx,wy,wz) index j -> state slot */
    const int ws[3] = {XW + 2, XW + 1, XW + 0};
    for (int i = 0; i < 3; i++) {
        /* rdot+ rows */
        A[(XV + i) * nx + XV + i] = 1;
        /* r+ rows */
        A[(XR + i) * nx + XR + i] = 1;
        A[(XR + i) * nx + XV + i] = dt;
        for (int f = 0; f < 4; f++) {
            B[(XV + i) * nu + 3 * f + i] = dt * p[f] / m;
            B[(XR + i) * nu + 3 * f + i] = dt * dt * p[f] / m;
        }
    }
    for (int i = 0; i < 3; i++) {
        /* w+ rows (stored reversed) */
        const int row = ws[i];
        for (int j = 0; j < 3; j++) {
            A[row * nx + XR + j] = dwn_dr[3 * i + j];
            A[row * nx + XTH + j] = dwn_dth[3 * i + j];
            A[row * nx + ws[j]] = dwn_dw[3 * i + j];
            for (int f = 0; f < 4; f++) B[row * nu + 3 * f + j] = dwn_df[f][3 * i + j];
        }
    }
    for (int i = 0; i < 3; i++) {
        /* th+ rows: th + dt T w+ */
        const int row = XTH + i;
        for (int j = 0; j < 3; j++) {
            real sr = 0, sth = 0, swj = 0;
            for (int k = 0; k < 3; k++) {
                sr += T[3 * i + k] * dwn_dr[3 * k + j];
                sth += T[3 * i + k] * dwn_dth[3 * k + j];
                swj += T[3 * i + k] * dwn_dw[3 * k + j];
            }
            A[row * nx + XR + j] = dt * sr;
            A[row * nx + XTH + j] = (i == j ? 1 : 0) + dt * (dTw[3 * i + j] + sth);
            A[row * nx + ws[j]] = dt * swj;
            for (int f = 0; f < 4; f++) {
                real sf = 0;
                for (int k = 0; k < 3; k++) sf += T[3 * i + k] * dwn_df[f][3 * k + j];
                B[row * nu + 3 * f + j] = dt * sf;
            }
        }
    }
}

/* ------------------------------------------------------------------ model 2 (whole body) */
enum { WB_NX = 42, WB_NU = 30, WB_NP = 20, WB_NG = 16, WB_NY = 90, WB_NYE = 66 };
enum { WQ = 0, WV = 18, WH = 36, WA = 0, WF = 18 };                    /* state / input offsets */
enum { RY_BASE = 0, RY_JOINT = 12, RY_ACC = 36, RY_SWING = 48, RY_FREG = 52, RY_CNT = 64, RY_CONS = 76, RY_POS = 82 };
enum { RE_BASE = 0, RE_JOINT = 12, RE_SWING = 36, RE_CNT = 40, RE_CONS = 52, RE_POS = 58 };

static void m3_vec(const real *a, const real *v, real *o) { /* o = a v */
    for (int i = 0; i < 3; i++) o[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2];
}
static void m3_acc3(const real *a, const real *b, const real *c, real s, real *o) { /* o += s a b c */
    real t[9], u[9];
    m3_mul(a, b, t);
    m3_mul(t, c, u);
    for (int i = 0; i < 9; i++) o[i] += s * u[i];
}
static void rot_axis(int axis, real ang, real *M, real *M1, real *M2) { /* R, dR/dang, d2R/dang2 */
    const real c = COS(ang), s = SIN(ang);
    for (int i = 0; i < 9; i++) M[i] = M1[i] = M2[i] = 0;
    const int i0 = (axis + 1) % 3, i1 = (axis + 2) % 3;              /* the rotated plane */
    M[4 * axis] = 1;
    M[3 * i0 + i0] = c;  M[3 * i0 + i1] = -s; M[3 * i1 + i0] = s;  M[3 * i1 + i1] = c;
    M1[3 * i0 + i0] = -s; M1[3 * i0 + i1] = -c; M1[3 * i1 + i0] = c;  M1[3 * i1 + i1] = -s;
    M2[3 * i0 + i0] = -c; M2[3 * i0 + i1] = s;  M2[3 * i1 + i0] = -s; M2[3 * i1 + i1] = -c;
}

/* Kinematics of the four point feet at (q, v = qdot):  world position p_i, its Jacobian J_i (3x9) with respect to
 * xi_i = [r, theta, ql_i] and the time derivative Jd_i of that Jacobian along v.  Since
 *   d/dxi_c ( J(xi) xidot ) = sum_a d2p/dxi_a dxi_c xidot_a = d/dt ( dp/dxi_c ),
 * Jd_i is also the Jacobian of the foot velocity with respect to the configuration. */
typedef struct {
    real R[9], Ra[3][9], Rd[9], Rad[3][9];
    real b[4][3], Jb[4][9];
    real p[4][3], J[4][27], Jd[4][27];
} wb_kin_t;

static void wb_kinematics(const real *mp, const real *x, int with_rates, wb_kin_t *k) {
    real M[3][9], M1[3][9], M2[3][9], Md[3][9], M1d[3][9];
    static const int axis_of[3] = {2, 1, 0};                          /* yaw: z, pitch: y, roll: x */
    static const real zero9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < 3; a++) {
        rot_axis(axis_of[a], x[WQ + 3 + a], M[a], M1[a], M2[a]);
        const real rate = with_rates ? x[WV + 3 + a] : 0;
        for (int i = 0; i < 9; i++) { Md[a][i] = M1[a][i] * rate; M1d[a][i] = M2[a][i] * rate; }
    }
    memcpy(k->R, zero9, sizeof zero9);
    m3_acc3(M[0], M[1], M[2], 1, k->R);
    memcpy(k->Rd, zero9, sizeof zero9);
    m3_acc3(Md[0], M[1], M[2], 1, k->Rd);
    m3_acc3(M[0], Md[1], M[2], 1, k->Rd);
    m3_acc3(M[0], M[1], Md[2], 1, k->Rd);
    for (int a = 0; a < 3; a++) {
        const real *F[3] = {M[0], M[1], M[2]}, *Fd[3] = {Md[0], Md[1], Md[2]};
        F[a] = M1[a];
        Fd[a] = M1d[a];
        memcpy(k->Ra[a], zero9, sizeof zero9);
        m3_acc3(F[0], F[1], F[2], 1, k->Ra[a]);
        memcpy(k->Rad[a], zero9, sizeof zero9);
        m3_acc3(Fd[0], F[1], F[2], 1, k->Rad[a]);
        m3_acc3(F[0], Fd[1], F[2], 1, k->Rad[a]);
        m3_acc3(F[0], F[1], Fd[2], 1, k->Rad[a]);
    }
    static const real sgx[4] = {1, 1, -1, -1}, sgy[4] = {1, -1, 1, -1};
    const real l1 = mp[MP_L1], l2 = mp[MP_L2];
    for (int f = 0; f < 4; f++) {
        const real *ql = x + WQ + 6 + 3 * f, *wl = x + WV + 6 + 3 * f;
        const real d = sgy[f] * mp[MP_LHIP];
        const real s1 = SIN(ql[0]), c1 = COS(ql[0]), s2 = SIN(ql[1]), c2 = COS(ql[1]);
        const real s23 = SIN(ql[1] + ql[2]), c23 = COS(ql[1] + ql[2]);
        const real vx = -l1 * s2 - l2 * s23, vz = -l1 * c2 - l2 * c23, vx3 = -l2 * s23, vz3 = -l2 * c23;
        real *b = k->b[f], *Jb = k->Jb[f];
        b[0] = sgx[f] * mp[MP_HIPX] + vx;
        b[1] = sgy[f] * mp[MP_HIPY] + d * c1 - vz * s1;
        b[2] = d * s1 + vz * c1;
        Jb[0] = 0;                 Jb[1] = vz;       Jb[2] = vz3;
        Jb[3] = -d * s1 - vz * c1; Jb[4] = vx * s1;  Jb[5] = vx3 * s1;
        Jb[6] = d * c1 - vz * s1;  Jb[7] = -vx * c1; Jb[8] = -vx3 * c1;
        real Jbd[9], bd[3];
        {
            const real w1 = with_rates ? wl[0] : 0, w2 = with_rates ? wl[1] : 0, w3 = with_rates ? wl[2] : 0;
            const real w23 = w2 + w3;
            const real s1d = c1 * w1, c1d = -s1 * w1;
            const real vxd = vz * w2 + vz3 * w3, vzd = -vx * w2 - vx3 * w3;
            const real vx3d = vz3 * w23, vz3d = -vx3 * w23;
            Jbd[0] = 0;                                 Jbd[1] = vzd;                   Jbd[2] = vz3d;
            Jbd[3] = -d * s1d - vzd * c1 - vz * c1d;    Jbd[4] = vxd * s1 + vx * s1d;   Jbd[5] = vx3d * s1 + vx3 * s1d;
            Jbd[6] = d * c1d - vzd * s1 - vz * s1d;     Jbd[7] = -vxd * c1 - vx * c1d;  Jbd[8] = -vx3d * c1 - vx3 * c1d;
            const real wv[3] = {w1, w2, w3};
            m3_vec(Jb, wv, bd);
        }
        real Rb[3];
        m3_vec(k->R, b, Rb);
        for (int i = 0; i < 3; i++) k->p[f][i] = x[WQ + i] + Rb[i];
        real *J = k->J[f], *Jd = k->Jd[f];
        for (int i = 0; i < 27; i++) J[i] = Jd[i] = 0;
        for (int i = 0; i < 3; i++) J[9 * i + i] = 1;
        for (int a = 0; a < 3; a++) {
            real t0[3], t1[3], t2[3];
            m3_vec(k->Ra[a], b, t0);
            m3_vec(k->Rad[a], b, t1);
            m3_vec(k->Ra[a], bd, t2);
            for (int i = 0; i < 3; i++) { J[9 * i + 3 + a] = t0[i]; Jd[9 * i + 3 + a] = t1[i] + t2[i]; }
        }
        for (int c = 0; c < 3; c++) {
            const real jc[3] = {Jb[c], Jb[3 + c], Jb[6 + c]}, jdc[3] = {Jbd[c], Jbd[3 + c], Jbd[6 + c]};
            real t0[3], t1[3], t2[3];
            m3_vec(k->R, jc, t0);
            m3_vec(k->Rd, jc, t1);
            m3_vec(k->R, jdc, t2);
            for (int i = 0; i < 3; i++) { J[9 * i + 6 + c] = t0[i]; Jd[9 * i + 6 + c] = t1[i] + t2[i]; }
        }
    }
}
/* column of x (or of v, +WV) that slot c of xi_f = [r, theta, ql_f] stands for */
static int wb_xi(int f, int c) { return c < 6 ? c : 6 + 3 * f + (c - 6); }

static void dyn_wholebody(const real *mp, const real *x, const real *u, const real *p, real *xn, real *A, real *B) {
    const int nx = WB_NX, nu = WB_NU;
    const real dt = mp[MP_DT];
    wb_kin_t k;
    wb_kinematics(mp, x, 0, &k);
    for (int i = 0; i < 18; i++) {
        const real vn = x[WV + i] + dt * u[WA + i];
        xn[WV + i] = vn;
        xn[WQ + i] = x[WQ + i] + dt * vn;
    }
    real F[3] = {0, 0, 0}, tau[3] = {0, 0, 0}, arm[4][3];
    for (int f = 0; f < 4; f++) {
        const real c = p[f];
        const real *ff = u + WF + 3 * f;
        real t[3];
        m3_vec(k.R, k.b[f], arm[f]);
        cross3(arm[f], ff, t);
        for (int j = 0; j < 3; j++) { F[j] += c * ff[j]; tau[j] += c * t[j]; }
    }
    for (int j = 0; j < 3; j++) {
        xn[WH + j] = x[WH + j] + dt * (F[j] + (j == 2 ? mp[MP_MASS] * mp[MP_GZ] : 0));
        xn[WH + 3 + j] = x[WH + 3 + j] + dt * tau[j];
    }
    if (!A) return;
    memset(A, 0, sizeof(real) * nx * nx);
    memset(B, 0, sizeof(real) * nx * nu);
    for (int i = 0; i < nx; i++) A[i * nx + i] = 1;
    for (int i = 0; i < 18; i++) {
        A[(WQ + i) * nx + WV + i] = dt;
        B[(WQ + i) * nu + WA + i] = dt * dt;
        B[(WV + i) * nu + WA + i] = dt;
    }
    for (int f = 0; f < 4; f++) {
        const real c = p[f];
        const real *ff = u + WF + 3 * f;
        for (int j = 0; j < 3; j++) B[(WH + j) * nu + WF + 3 * f + j] = dt * c;
        /* d(arm x f)/df = [arm]x */
        const real ax[9] = {0, -arm[f][2], arm[f][1], arm[f][2], 0, -arm[f][0], -arm[f][1], arm[f][0], 0};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) B[(WH + 3 + i) * nu + WF + 3 * f + j] = dt * c * ax[3 * i + j];
        /* d(arm x f)/dxi_c = (d arm/dxi_c) x f ; arm = R b does not depend on r */
        for (int cc = 3; cc < 9; cc++) {
            const real da[3] = {k.J[f][cc], k.J[f][9 + cc], k.J[f][18 + cc]};
            real t[3];
            cross3(da, ff, t);
            for (int i = 0; i < 3; i++) A[(WH + 3 + i) * nx + WQ + wb_xi(f, cc)] += dt * c * t[i];
        }
    }
}

/* Residuals of the least-squares cost at (x, u, p), u == NULL: terminal node.  res[ny] (reference already
 * subtracted), and, if Jx != NULL, the Jacobian with respect to x, dense [ny][nx].  The Jacobian with respect to
 * u is a selection: rows RY_ACC + i -> u[6 + i], rows RY_FREG + i -> u[18 + i]. */
static void wb_residuals(const real *mp, const real *x, const real *u, const real *p, const real *yref, real *res,
                         real *Jx) {
    const int nx = WB_NX, term = (u == NULL);
    const int ny = term ? WB_NYE : WB_NY;
    const int r_sw = term ? RE_SWING : RY_SWING, r_ct = term ? RE_CNT : RY_CNT, r_cs = term ? RE_CONS : RY_CONS;
    const int r_ps = term ? RE_POS : RY_POS;
    wb_kin_t k;
    wb_kinematics(mp, x, 1, &k);
    if (Jx) memset(Jx, 0, sizeof(real) * ny * nx);
    for (int i = 0; i < 6; i++) {
        res[RY_BASE + i] = x[WQ + i];
        res[RY_BASE + 6 + i] = x[WV + i];
        if (Jx) { Jx[(RY_BASE + i) * nx + WQ + i] = 1; Jx[(RY_BASE + 6 + i) * nx + WV + i] = 1; }
    }
    for (int i = 0; i < 12; i++) {
        res[RY_JOINT + i] = x[WQ + 6 + i];
        res[RY_JOINT + 12 + i] = x[WV + 6 + i];
        if (Jx) { Jx[(RY_JOINT + i) * nx + WQ + 6 + i] = 1; Jx[(RY_JOINT + 12 + i) * nx + WV + 6 + i] = 1; }
    }
    if (!term) {
        for (int i = 0; i < 12; i++) { res[RY_ACC + i] = u[WA + 6 + i]; res[RY_FREG + i] = u[WF + i]; }
    }
    const real pg = mp[MP_PGAIN];
    for (int f = 0; f < 4; f++) {
        const real c = p[f], peak = p[4 + f], ppz = p[8 + 3 * f + 2];
        res[r_sw + f] = peak * k.p[f][2];
        real vel[3] = {0, 0, 0};
        for (int i = 0; i < 3; i++)
            for (int cc = 0; cc < 9; cc++) vel[i] += k.J[f][9 * i + cc] * x[WV + wb_xi(f, cc)];
        for (int i = 0; i < 3; i++) res[r_ct + 3 * f + i] = c * (vel[i] + (i == 2 ? pg * (k.p[f][2] - ppz) : 0));
        /* foot placement (pos_cost, solver.py:128-137,272-273): world x, y of the foot against the planned location */
        for (int i = 0; i < 2; i++) res[r_ps + 2 * f + i] = k.p[f][i];
        if (Jx)
            for (int cc = 0; cc < 9; cc++) {
                const int col = wb_xi(f, cc);
                for (int i = 0; i < 2; i++) Jx[(r_ps + 2 * f + i) * nx + WQ + col] = k.J[f][9 * i + cc];
                Jx[(r_sw + f) * nx + WQ + col] = peak * k.J[f][18 + cc];
                for (int i = 0; i < 3; i++) {
                    Jx[(r_ct + 3 * f + i) * nx + WQ + col] = c * (k.Jd[f][9 * i + cc] + (i == 2 ? pg * k.J[f][18 + cc] : 0));
                    Jx[(r_ct + 3 * f + i) * nx + WV + col] = c * k.J[f][9 * i + cc];
                }
            }
    }
    {   /* consistency  h - A_g(q) v,  A_g v = [m rdot ; R I_b E(theta) thetadot] */
        const real *th = x + WQ + 3, *thd = x + WV + 3;
        const real Ib[3] = {mp[MP_IXX], mp[MP_IYY], mp[MP_IZZ]};
        const real sy = SIN(th[1]), cy = COS(th[1]), sx = SIN(th[2]), cx = COS(th[2]);
        const real E[9] = {-sy, 0, 1, cy * sx, cx, 0, cx * cy, -sx, 0};
        const real Ea[3][9] = {{0, 0, 0, 0, 0, 0, 0, 0, 0},
                               {-cy, 0, 0, -sy * sx, 0, 0, -cx * sy, 0, 0},
                               {0, 0, 0, cy * cx, -sx, 0, -sx * cy, -cx, 0}};
        real wb[3], Iw[3], L[3];
        m3_vec(E, thd, wb);
        for (int i = 0; i < 3; i++) Iw[i] = Ib[i] * wb[i];
        m3_vec(k.R, Iw, L);
        for (int i = 0; i < 3; i++) {
            res[r_cs + i] = x[WH + i] - mp[MP_MASS] * x[WV + i];
            res[r_cs + 3 + i] = x[WH + 3 + i] - L[i];
        }
        if (Jx) {
            for (int i = 0; i < 3; i++) {
                Jx[(r_cs + i) * nx + WH + i] = 1;
                Jx[(r_cs + i) * nx + WV + i] = -mp[MP_MASS];
                Jx[(r_cs + 3 + i) * nx + WH + 3 + i] = 1;
            }
            for (int a = 0; a < 3; a++) {
                real t0[3], t1[3], ew[3], iew[3], ecol[3], iecol[3], t2[3];
                m3_vec(k.Ra[a], Iw, t0);
                m3_vec(Ea[a], thd, ew);
                for (int i = 0; i < 3; i++) iew[i] = Ib[i] * ew[i];
                m3_vec(k.R, iew, t1);
                for (int i = 0; i < 3; i++) { ecol[i] = E[3 * i + a]; iecol[i] = Ib[i] * ecol[i]; }
                m3_vec(k.R, iecol, t2);
                for (int i = 0; i < 3; i++) {
                    Jx[(r_cs + 3 + i) * nx + WQ + 3 + a] = -(t0[i] + t1[i]);
                    Jx[(r_cs + 3 + i) * nx + WV + 3 + a] = -t2[i];
                }
            }
        }
    }
    for (int i = 0; i < ny; i++) res[i] -= yref[i];
}

/* Gauss-Newton terms of one node: returns the cost 1/2 sum W res^2; if Q != NULL also
 *   Q = Jx' W Jx + reg I  [nx][nx],  q = Jx' W res,  and for a stage  Rd = diag(Ju' W Ju) + reg  [nu],  r = Ju' W res. */
static real wb_node_terms(const real *mp, const real *W, real reg, const real *x, const real *u, const real *p,
                          const real *yref, real *Q, real *q, real *Rd, real *r) {
    const int nx = WB_NX, nu = WB_NU, term = (u == NULL);
    const int ny = term ? WB_NYE : WB_NY;
    real res[WB_NY];
    real *Jx = Q ? (real *)malloc(sizeof(real) * ny * nx) : NULL;
    wb_residuals(mp, x, u, p, yref, res, Jx);
    real cost = 0;
    for (int i = 0; i < ny; i++) cost += (real)0.5 * W[i] * res[i] * res[i];
    if (Q) {
        for (int a = 0; a < nx; a++) {
            real g = 0;
            for (int i = 0; i < ny; i++) g += Jx[i * nx + a] * W[i] * res[i];
            q[a] = g;
            for (int b = 0; b < nx; b++) {
                real s = (a == b) ? reg : 0;
                for (int i = 0; i < ny; i++) s += Jx[i * nx + a] * W[i] * Jx[i * nx + b];
                Q[a * nx + b] = s;
            }
        }
        if (!term) {
            for (int i = 0; i < nu; i++) { Rd[i] = reg; r[i] = 0; }
            for (int i = 0; i < 12; i++) {
                Rd[WA + 6 + i] += W[RY_ACC + i];
                r[WA + 6 + i] = W[RY_ACC + i] * res[RY_ACC + i];
                Rd[WF + i] += W[RY_FREG + i];
                r[WF + i] = W[RY_FREG + i] * res[RY_FREG + i];
            }
        }
        free(Jx);
    }
    return cost;
}

/* test hooks: residuals and their state Jacobian (terminal: u == NULL) */
void oracle_wb_residuals(const real *mp, const real *x, const real *u, const real *p, const real *yref, real *res,
                         real *Jx) {
    wb_residuals(mp, x, u, p, yref, res, Jx);
}
void oracle_wb_feet(const real *mp, const real *x, real *pos /*[4][3]*/, real *vel /*[4][3]*/) {
    wb_kin_t k;
    wb_kinematics(mp, x, 1, &k);
    for (int f = 0; f < 4; f++)
        for (int i = 0; i < 3; i++) {
            pos[3 * f + i] = k.p[f][i];
            real s = 0;
            for (int cc = 0; cc < 9; cc++) s += k.J[f][9 * i + cc] * x[WV + wb_xi(f, cc)];
            vel[3 * f + i] = s;
        }
}

void oracle_dynamics(int model_id, const real *mp, const real *x, const real *u, const real *p,
                     real *xn, real *A, real *B) {
    if (model_id == 0)
        dyn_double_integrator(mp, x, u, xn, A, B);
    else if (model_id == 1)
        dyn_centroidal(mp, x, u, p, xn, A, B);
    else
        dyn_wholebody(mp, x, u, p, xn, A, B);
}

/* inequality rows  G u <= h ; act[j] = 1 if the row is enforced */
void oracle_constraints(int model_id, const real *mp, const real *p, real *G, real *h, int *act) {
    if (model_id == 0) {
        const int nu = 2;
        const real um = mp[MP_UMAX];
        memset(G, 0, sizeof(real) * 4 * nu);
        for (int i = 0; i < 2; i++) {
            G[(2 * i) * nu + i] = 1;
            G[(2 * i + 1) * nu + i] = -1;
            h[2 * i] = um;
            h[2 * i + 1] = um;
            act[2 * i] = act[2 * i + 1] = um > 0;
        }
        return;
    }
    const int nu = (model_id == 2) ? WB_NU : 12, fo = (model_id == 2) ? WF : 0; /* forces start at u[fo] */
    const real mu = mp[MP_MU];
    memset(G, 0, sizeof(real) * 16 * nu);
    for (int f = 0; f < 4; f++) {
        const int a = p[f] > (real)0.5;
        for (int j = 0; j < 4; j++) {
            const int row = 4 * f + j;
            G[row * nu + fo + 3 * f + (j >> 1)] = (j & 1) ? (real)-1 : (real)1; /* +fx,-fx,+fy,-fy */
            G[row * nu + fo + 3 * f + 2] = -mu;
            h[row] = 0;
            act[row] = a;
        }
    }
}

/* ------------------------------------------------------------------ dense Riccati
 * All matrices row-major.  Q[(N+1)][nx][nx], R[N][nu][nu], q[(N+1)][nx], r[N][nu],
 * A[N][nx][nx], B[N][nx][nu], d[N][nx].  Outputs dX[(N+1)][nx], dU[N][nu] and, if
 * non-NULL, K[N][nu][nx], kff[N][nu], P[(N+1)][nx][nx].
 * returns 0, or 4 if a pivot of Huu is not positive (status "qp failure").
 */
int oracle_riccati(int nx, int nu, int N, const real *Q, const real *R, const real *q,
                   const real *r, const real *A, const real *B, const real *d, const real *dx0,
                   real *dX, real *dU, real *Kout, real *kout, real *Pout) {
    real *K = (real *)malloc(sizeof(real) * N * nu * nx);
    real *kf = (real *)malloc(sizeof(real) * N * nu);
    real P[MAXN * MAXN], p[MAXN], Pd[MAXN], PA[MAXN * MAXN], PB[MAXN * MAXN];
    real Huu[MAXN * MAXN], Hux[MAXN * MAXN], hu[MAXN], Hxx[MAXN * MAXN], hx[MAXN];
    real Y[MAXN * MAXN], yv[MAXN], dd[MAXN];
    int status = 0;
    memcpy(P, Q + (size_t)N * nx * nx, sizeof(real) * nx * nx);
    memcpy(p, q + (size_t)N * nx, sizeof(real) * nx);
    if (Pout) memcpy(Pout + (size_t)N * nx * nx, P, sizeof(real) * nx * nx);
    for (int k = N - 1; k >= 0; k--) {
        const real *Ak = A + (size_t)k * nx * nx, *Bk = B + (size_t)k * nx * nu;
        const real *dk = d + (size_t)k * nx;
        for (int i = 0; i < nx; i++) {
            real s = p[i];
            for (int j = 0; j < nx; j++) s += P[i * nx + j] * dk[j];
            Pd[i] = s;
            for (int j = 0; j < nx; j++) {
                real t = 0;
                for (int l = 0; l < nx; l++) t += P[i * nx + l] * Ak[l * nx + j];
                PA[i * nx + j] = t;
            }
            for (int j = 0; j < nu; j++) {
                real t = 0;
                for (int l = 0; l < nx; l++) t += P[i * nx + l] * Bk[l * nu + j];
                PB[i * nu + j] = t;
            }
        }
        for (int i = 0; i < nu; i++) {
            for (int j = 0; j < nu; j++) {
                real t = R[(size_t)k * nu * nu + i * nu + j];
                for (int l = 0; l < nx; l++) t += Bk[l * nu + i] * PB[l * nu + j];
                Huu[i * nu + j] = t;
            }
            for (int j = 0; j < nx; j++) {
                real t = 0;
                for (int l = 0; l < nx; l++) t += Bk[l * nu + i] * PA[l * nx + j];
                Hux[i * nx + j] = t;
            }
            real t = r[(size_t)k * nu + i];
            for (int l = 0; l < nx; l++) t += Bk[l * nu + i] * Pd[l];
            hu[i] = t;
        }
        for (int i = 0; i < nx; i++) {
            for (int j = 0; j < nx; j++) {
                real t = Q[(size_t)k * nx * nx + i * nx + j];
                for (int l = 0; l < nx; l++) t += Ak[l * nx + i] * PA[l * nx + j];
                Hxx[i * nx + j] = t;
            }
            real t = q[(size_t)k * nx + i];
            for (int l = 0; l < nx; l++) t += Ak[l * nx + i] * Pd[l];
            hx[i] = t;
        }
        /* LDL' elimination of Huu applied to [Hux | hu]; Y = D^-1/2 L^-1 Hux */
        for (int i = 0; i < nu; i++) {
            memcpy(Y + i * nx, Hux + i * nx, sizeof(real) * nx);
            yv[i] = hu[i];
        }
        for (int j = 0; j < nu; j++) {
            const real piv = Huu[j * nu + j];
            if (!(piv > 0)) status = 4;
            dd[j] = piv;
            const real inv = (real)1 / piv;
            for (int i = j + 1; i < nu; i++) {
                const real l = Huu[i * nu + j] * inv;
                for (int c = j + 1; c < nu; c++) Huu[i * nu + c] -= l * Huu[j * nu + c];
                for (int c = 0; c < nx; c++) Y[i * nx + c] -= l * Y[j * nx + c];
                yv[i] -= l * yv[j];
                Huu[i * nu + j] = l; /* unit-lower multiplier */
            }
        }
        /* K = -L^-T D^-1 (L^-1 Hux) by back substitution */
        real *Kk = K + (size_t)k * nu * nx, *kk = kf + (size_t)k * nu;
        for (int j = nu - 1; j >= 0; j--) {
            const real inv = (real)1 / dd[j];
            for (int c = 0; c < nx; c++) {
                real s = -Y[j * nx + c] * inv;
                for (int i = j + 1; i < nu; i++) s -= Huu[i * nu + j] * Kk[i * nx + c];
                Kk[j * nx + c] = s;
            }
            real s = -yv[j] * inv;
            for (int i = j + 1; i < nu; i++) s -= Huu[i * nu + j] * kk[i];
            kk[j] = s;
        }
        /* P = Hxx - (L^-1 Hux)' D^-1 (L^-1 Hux) ; p = hx - (..)' D^-1 (L^-1 hu) */
        for (int i = 0; i < nx; i++) {
            for (int j = 0; j < nx; j++) {
                real t = Hxx[i * nx + j];
                for (int l = 0; l < nu; l++) t -= Y[l * nx + i] * Y[l * nx + j] / dd[l];
                P[i * nx + j] = t;
            }
            real t = hx[i];
            for (int l = 0; l < nu; l++) t -= Y[l * nx + i] * yv[l] / dd[l];
            p[i] = t;
        }
        /* keep P symmetric */
        for (int i = 0; i < nx; i++)
            for (int j = i + 1; j < nx; j++) {
                const real t = (real)0.5 * (P[i * nx + j] + P[j * nx + i]);
                P[i * nx + j] = P[j * nx + i] = t;
            }
        if (Pout) memcpy(Pout + (size_t)k * nx * nx, P, sizeof(real) * nx * nx);
    }
    /* forward */
    memcpy(dX, dx0, sizeof(real) * nx);
    for (int k = 0; k < N; k++) {
        const real *Ak = A + (size_t)k * nx * nx, *Bk = B + (size_t)k * nx * nu;
        const real *Kk = K + (size_t)k * nu * nx;
        real *dx = dX + (size_t)k * nx, *du = dU + (size_t)k * nu, *dxn = dX + (size_t)(k + 1) * nx;
        for (int i = 0; i < nu; i++) {
            real s = kf[(size_t)k * nu + i];
            for (int j = 0; j < nx; j++) s += Kk[i * nx + j] * dx[j];
            du[i] = s;
        }
        for (int i = 0; i < nx; i++) {
            real s = d[(size_t)k * nx + i];
            for (int j = 0; j < nx; j++) s += Ak[i * nx + j] * dx[j];
            for (int j = 0; j < nu; j++) s += Bk[i * nu + j] * du[j];
            dxn[i] = s;
        }
    }
    if (Kout) memcpy(Kout, K, sizeof(real) * N * nu * nx);
    if (kout) memcpy(kout, kf, sizeof(real) * N * nu);
    free(K);
    free(kf);
    return status;
}

/* ------------------------------------------------------------------ cost / merit */
static real stage_cost(int nx, int nu, const real *W, const real *x, const real *u, const real *yref) {
    real c = 0;
    for (int i = 0; i < nx; i++) c += (real)0.5 * W[i] * (x[i] - yref[i]) * (x[i] - yref[i]);
    for (int i = 0; i < nu; i++)
        c += (real)0.5 * W[nx + i] * (u[i] - yref[nx + i]) * (u[i] - yref[nx + i]);
    return c;
}

static real merit(int model_id, int nx, int nu, int np, int ng, int N, const real *mp,
                  const real *W, const real *We, const real *x0, const real *yref, int per_stage,
                  const real *yref_e, const real *params, const real *X, const real *U, real rho,
                  real *cost_out) {
    real cost = 0, viol = 0;
    real xn[MAXN], G[MAXN * MAXN], h[MAXN];
    int act[MAXN];
    const int ny = (model_id == 2) ? WB_NY : nx + nu;
    for (int i = 0; i < nx; i++) viol += FABS(x0[i] - X[i]);
    for (int k = 0; k < N; k++) {
        const real *x = X + (size_t)k * nx, *u = U + (size_t)k * nu;
        const real *p = params + (size_t)k * np;
        if (model_id == 2)
            cost += wb_node_terms(mp, W, 0, x, u, p, yref + (per_stage ? (size_t)k * ny : 0), NULL, NULL, NULL, NULL);
        else
        cost += stage_cost(nx, nu, W, x, u, yref + (per_stage ? (size_t)k * ny : 0));
        oracle_dynamics(model_id, mp, x, u, p, xn, NULL, NULL);
        for (int i = 0; i < nx; i++) viol += FABS(xn[i] - X[(size_t)(k + 1) * nx + i]);
        if (ng > 0) {
            oracle_constraints(model_id, mp, p, G, h, act);
            for (int j = 0; j < ng; j++) {
                if (!act[j]) continue;
                real c = -h[j];
                for (int i = 0; i < nu; i++) c += G[j * nu + i] * u[i];
                if (c > 0) viol += c;
            }
        }
    }
    if (model_id == 2)
        cost += wb_node_terms(mp, We, 0, X + (size_t)N * nx, NULL, params + (size_t)N * np, yref_e, NULL, NULL, NULL, NULL);
    else
    for (int i = 0; i < nx; i++) {
        const real e = X[(size_t)N * nx + i] - yref_e[i];
        cost += (real)0.5 * We[i] * e * e;
    }
    if (cost_out) *cost_out = cost;
    return cost + rho * viol;
}

/* ------------------------------------------------------------------ NMPC solve
 * Single problem.  X[(N+1)][nx], U[N][nu] in: warm start, out: solution.
 * yref: [N][ny] if opt[OP_YREF_PER_STAGE] else [ny];  params [(N+1)][np].
 * stats[4] = {cost at last linearisation, max|step|, alpha, sqp iterations}.
 * returns status 0 ok(converged), 1 NaN, 2 max iter, 4 qp failure.
 */
int oracle_solve_ws(int model_id, int N, const real *mp, const real *opt, const real *W,
                    const real *We, const real *x0, const real *yref, const real *yref_e,
                    const real *params, real *X, real *U, real *stats, real *Sio, real *Lio);
int oracle_solve(int model_id, int N, const real *mp, const real *opt, const real *W,
                 const real *We, const real *x0, const real *yref, const real *yref_e,
                 const real *params, real *X, real *U, real *stats) {
    return oracle_solve_ws(model_id, N, mp, opt, W, We, x0, yref, yref_e, params, X, U, stats, NULL, NULL);
}
/* The same solve with the interior point's slacks and multipliers as in/out state, Sio / Lio [N][ng] (NULL: none).
 * opt[OP_IPM_WARM] != 0: warm-started interior point (the reference: set_warm_start_inner_qp, warm_start_multipliers,
 * solver.py:76-77,339) -- the multipliers of the previous QP solution are the starting point of the next one, across
 * the SQP iterations of a call and, with opt[OP_WS_HAVE], across calls through the warm-start shift opt[OP_WS_SHIFT]
 * (stage k starts from the previous call's stage k + shift; the exposed tail cold-starts).  As HPIPM's warm start does,
 * the slacks are recomputed from the primal warm start, s = max(-c, s_floor), and the multipliers are kept,
 * lam = max(lam_prev, lam_floor).  [decl] */
int oracle_solve_ws(int model_id, int N, const real *mp, const real *opt, const real *W,
                    const real *We, const real *x0, const real *yref, const real *yref_e,
                    const real *params, real *X, real *U, real *stats, real *Sio, real *Lio) {
    int nx, nu, np, ng;
    if (oracle_dims(model_id, &nx, &nu, &np, &ng)) return -1;
    const int ny = (model_id == 2) ? WB_NY : nx + nu;
    const int max_sqp = (int)opt[OP_MAX_SQP], n_ipm = (int)opt[OP_N_IPM];
    const int per_stage = opt[OP_YREF_PER_STAGE] != 0;
    const real reg = opt[OP_REG], reg_e = opt[OP_REG_E];
    size_t sQ = (size_t)(N + 1) * nx * nx, sR = (size_t)N * nu * nu;
    real *Q = (real *)calloc(sQ, sizeof(real)), *R = (real *)calloc(sR, sizeof(real));
    real *Rt = (real *)calloc(sR, sizeof(real));
    real *q = (real *)calloc((size_t)(N + 1) * nx, sizeof(real));
    real *r = (real *)calloc((size_t)N * nu, sizeof(real)), *rt = (real *)calloc((size_t)N * nu, sizeof(real));
    real *A = (real *)calloc((size_t)N * nx * nx, sizeof(real));
    real *Bm = (real *)calloc((size_t)N * nx * nu, sizeof(real));
    real *d = (real *)calloc((size_t)N * nx, sizeof(real));
    real *dX = (real *)calloc((size_t)(N + 1) * nx, sizeof(real)), *dU = (real *)calloc((size_t)N * nu, sizeof(real));
    real *dXp = (real *)calloc((size_t)(N + 1) * nx, sizeof(real)), *dUp = (real *)calloc((size_t)N * nu, sizeof(real));
    real *G = (real *)calloc((size_t)N * MAXN * MAXN, sizeof(real));
    real *c = (real *)calloc((size_t)N * MAXN, sizeof(real));
    real *s = (real *)calloc((size_t)N * MAXN, sizeof(real)), *lam = (real *)calloc((size_t)N * MAXN, sizeof(real));
    int *act = (int *)calloc((size_t)N * MAXN, sizeof(int));
    real *Xt = (real *)calloc((size_t)(N + 1) * nx, sizeof(real)), *Ut = (real *)calloc((size_t)N * nu, sizeof(real));
    real dx0[MAXN], xn[MAXN], h[MAXN];
    int status = 2, it = 0;
    real cost = 0, stepn = 0, alpha = 1;

    for (it = 0; it < max_sqp; it++) {
        /* 1. linearise */
        cost = 0;
        for (int k = 0; k < N; k++) {
            const real *x = X + (size_t)k * nx, *u = U + (size_t)k * nu;
            const real *p = params + (size_t)k * np;
            const real *yr = yref + (per_stage ? (size_t)k * ny : 0);
            oracle_dynamics(model_id, mp, x, u, p, xn, A + (size_t)k * nx * nx, Bm + (size_t)k * nx * nu);
            for (int i = 0; i < nx; i++) d[(size_t)k * nx + i] = xn[i] - X[(size_t)(k + 1) * nx + i];
            if (model_id == 2) {   /* dense Gauss-Newton blocks J'WJ */
                real Rd[WB_NU];
                cost += wb_node_terms(mp, W, reg, x, u, p, yr, Q + (size_t)k * nx * nx, q + (size_t)k * nx, Rd,
                                      r + (size_t)k * nu);
                for (int i = 0; i < nu; i++) R[(size_t)k * nu * nu + i * nu + i] = Rd[i];
            } else {
            for (int i = 0; i < nx; i++) {
                q[(size_t)k * nx + i] = W[i] * (x[i] - yr[i]);
                Q[(size_t)k * nx * nx + i * nx + i] = W[i] + reg;
            }
            for (int i = 0; i < nu; i++) {
                r[(size_t)k * nu + i] = W[nx + i] * (u[i] - yr[nx + i]);
                R[(size_t)k * nu * nu + i * nu + i] = W[nx + i] + reg;
            }
            cost += stage_cost(nx, nu, W, x, u, yr);
            }
            if (ng > 0 && n_ipm > 0) {
                oracle_constraints(model_id, mp, p, G + (size_t)k * MAXN * MAXN, h, act + (size_t)k * MAXN);
                for (int j = 0; j < ng; j++) {
                    real cv = -h[j];
                    for (int i = 0; i < nu; i++) cv += G[(size_t)k * MAXN * MAXN + j * nu + i] * u[i];
                    c[(size_t)k * MAXN + j] = cv;
                }
            }
        }
        if (model_id == 2)
            cost += wb_node_terms(mp, We, reg_e, X + (size_t)N * nx, NULL, params + (size_t)N * np, yref_e,
                                  Q + (size_t)N * nx * nx, q + (size_t)N * nx, NULL, NULL);
        for (int i = 0; i < nx; i++) {
            if (model_id != 2) {
            const real e = X[(size_t)N * nx + i] - yref_e[i];
            q[(size_t)N * nx + i] = We[i] * e;
            Q[(size_t)N * nx * nx + i * nx + i] = We[i] + reg_e;
            cost += (real)0.5 * We[i] * e * e;
            }
            dx0[i] = x0[i] - X[i];
        }
        /* 2. QP */
        int qp_status = 0, n_act = 0;
        if (ng > 0 && n_ipm > 0)
            for (int k = 0; k < N; k++)
                for (int j = 0; j < ng; j++) n_act += act[(size_t)k * MAXN + j];
        if (n_act == 0) {
            qp_status = oracle_riccati(nx, nu, N, Q, R, q, r, A, Bm, d, dx0, dX, dU, NULL, NULL, NULL);
        } else {
            const real mu0 = opt[OP_MU0], sigma = opt[OP_SIGMA], smin = opt[OP_SMIN], gamma = opt[OP_GAMMA];
            const int warm = opt[OP_IPM_WARM] != 0, ws_shift = (int)opt[OP_WS_SHIFT];
            const real sfloor = opt[OP_WS_SFLOOR], lfloor = opt[OP_WS_LFLOOR];
            for (int k = 0; k < N; k++)
                for (int j = 0; j < ng; j++) {
                    const size_t id = (size_t)k * MAXN + j;
                    int have = 0;
                    real lprev = 0;
                    if (warm && it > 0) { have = 1; lprev = lam[id]; }
                    else if (warm && Lio && opt[OP_WS_HAVE] != 0 && k + ws_shift < N) { have = 1; lprev = Lio[(size_t)(k + ws_shift) * ng + j]; }
                    if (have && act[id]) {
                        s[id] = -c[id] > sfloor ? -c[id] : sfloor;
                        lam[id] = lprev > lfloor ? lprev : lfloor;
                    } else {
                        s[id] = -c[id] > smin ? -c[id] : smin;
                        lam[id] = mu0 / s[id];
                    }
                }
            memset(dX, 0, sizeof(real) * (N + 1) * nx);
            memset(dU, 0, sizeof(real) * N * nu);
            for (int ii = 0; ii < n_ipm; ii++) {
                real mu = 0;
                for (int k = 0; k < N; k++)
                    for (int j = 0; j < ng; j++) {
                        const size_t id = (size_t)k * MAXN + j;
                        if (act[id]) mu += s[id] * lam[id];
                    }
                real tau = sigma * mu / (real)n_act;
                if (tau < opt[OP_TAU_MIN]) tau = opt[OP_TAU_MIN];
                for (int k = 0; k < N; k++) {
                    const real *Gk = G + (size_t)k * MAXN * MAXN;
                    memcpy(Rt + (size_t)k * nu * nu, R + (size_t)k * nu * nu, sizeof(real) * nu * nu);
                    memcpy(rt + (size_t)k * nu, r + (size_t)k * nu, sizeof(real) * nu);
                    for (int j = 0; j < ng; j++) {
                        const size_t id = (size_t)k * MAXN + j;
                        if (!act[id]) continue;
                        const real D = lam[id] / s[id];
                        const real v = tau / s[id] + lam[id] + D * c[id];
                        for (int a = 0; a < nu; a++) {
                            rt[(size_t)k * nu + a] += Gk[j * nu + a] * v;
                            for (int b = 0; b < nu; b++)
                                Rt[(size_t)k * nu * nu + a * nu + b] += Gk[j * nu + a] * D * Gk[j * nu + b];
                        }
                    }
                }
                int st = oracle_riccati(nx, nu, N, Q, Rt, q, rt, A, Bm, d, dx0, dXp, dUp, NULL, NULL, NULL);
                if (st) qp_status = st;
                real ap = 1, ad = 1;
                for (int k = 0; k < N; k++) {
                    const real *Gk = G + (size_t)k * MAXN * MAXN;
                    for (int j = 0; j < ng; j++) {
                        const size_t id = (size_t)k * MAXN + j;
                        if (!act[id]) continue;
                        real gd = 0;
                        for (int a = 0; a < nu; a++) gd += Gk[j * nu + a] * dUp[(size_t)k * nu + a];
                        const real ds = -(gd + c[id]) - s[id];
                        const real dl = tau / s[id] - lam[id] - lam[id] / s[id] * ds;
                        if (ds < 0) { const real a_ = -gamma * s[id] / ds; if (a_ < ap) ap = a_; }
                        if (dl < 0) { const real a_ = -gamma * lam[id] / dl; if (a_ < ad) ad = a_; }
                    }
                }
                for (int k = 0; k < N; k++) {
                    const real *Gk = G + (size_t)k * MAXN * MAXN;
                    for (int j = 0; j < ng; j++) {
                        const size_t id = (size_t)k * MAXN + j;
                        if (!act[id]) continue;
                        real gd = 0;
                        for (int a = 0; a < nu; a++) gd += Gk[j * nu + a] * dUp[(size_t)k * nu + a];
                        const real ds = -(gd + c[id]) - s[id];
                        const real dl = tau / s[id] - lam[id] - lam[id] / s[id] * ds;
                        s[id] += ap * ds;
                        lam[id] += ad * dl;
                    }
                }
                for (size_t i = 0; i < (size_t)(N + 1) * nx; i++) dX[i] += ap * (dXp[i] - dX[i]);
                for (size_t i = 0; i < (size_t)N * nu; i++) dU[i] += ap * (dUp[i] - dU[i]);
            }
        }
        /* 3. step */
        stepn = 0;
        int bad = 0;
        for (size_t i = 0; i < (size_t)(N + 1) * nx; i++) {
            if (!(dX[i] == dX[i]) || FABS(dX[i]) > (real)1e30) bad = 1;
            if (FABS(dX[i]) > stepn) stepn = FABS(dX[i]);
        }
        for (size_t i = 0; i < (size_t)N * nu; i++) {
            if (!(dU[i] == dU[i]) || FABS(dU[i]) > (real)1e30) bad = 1;
            if (FABS(dU[i]) > stepn) stepn = FABS(dU[i]);
        }
        if (bad) { status = 1; it++; break; }
        alpha = 1;
        if (opt[OP_LINE_SEARCH] != 0) {
            const real rho = opt[OP_RHO];
            const real m0 = merit(model_id, nx, nu, np, ng, N, mp, W, We, x0, yref, per_stage, yref_e,
                                  params, X, U, rho, NULL);
            for (int t = 0; t < 6; t++) {
                for (size_t i = 0; i < (size_t)(N + 1) * nx; i++) Xt[i] = X[i] + alpha * dX[i];
                for (size_t i = 0; i < (size_t)N * nu; i++) Ut[i] = U[i] + alpha * dU[i];
                const real m1 = merit(model_id, nx, nu, np, ng, N, mp, W, We, x0, yref, per_stage,
                                      yref_e, params, Xt, Ut, rho, NULL);
                if (m1 < m0 || t == 5) break;
                alpha *= (real)0.5;
            }
        }
        for (size_t i = 0; i < (size_t)(N + 1) * nx; i++) X[i] += alpha * dX[i];
        for (size_t i = 0; i < (size_t)N * nu; i++) U[i] += alpha * dU[i];
        if (qp_status) { status = qp_status; it++; break; }
        if (opt[OP_NLP_TOL] > 0 && stepn < opt[OP_NLP_TOL]) { status = 0; it++; break; }
    }
    if (stats) { stats[0] = cost; stats[1] = stepn; stats[2] = alpha; stats[3] = (real)it; }
    if (Sio && Lio)
        for (int k = 0; k < N; k++)
            for (int j = 0; j < ng; j++) {
                Sio[(size_t)k * ng + j] = s[(size_t)k * MAXN + j];
                Lio[(size_t)k * ng + j] = act[(size_t)k * MAXN + j] ? lam[(size_t)k * MAXN + j] : 0;
            }
    free(Q); free(R); free(Rt); free(q); free(r); free(rt); free(A); free(Bm); free(d);
    free(dX); free(dU); free(dXp); free(dUp); free(G); free(c); free(s); free(lam); free(act);
    free(Xt); free(Ut);
    return status;
}

/* batch driver, OpenMP over problems (nthreads <= 0: runtime default).
 * Layouts batch-major: x0[B][nx], yref[B][N][ny] or [B][ny], yref_e[B][nx],
 * params[B][N+1][np], X[B][N+1][nx], U[B][N][nu], status[B], stats[B][4]. */
int oracle_solve_batch_ws(int model_id, int N, int B, const real *mp, const real *opt, const real *W,
                          const real *We, const real *x0, const real *yref, const real *yref_e,
                          const real *params, real *X, real *U, int *status, real *stats, int nthreads,
                          real *S, real *L);
int oracle_solve_batch(int model_id, int N, int B, const real *mp, const real *opt, const real *W,
                       const real *We, const real *x0, const real *yref, const real *yref_e,
                       const real *params, real *X, real *U, int *status, real *stats, int nthreads) {
    return oracle_solve_batch_ws(model_id, N, B, mp, opt, W, We, x0, yref, yref_e, params, X, U, status, stats, nthreads, NULL, NULL);
}
/* S, L: [B][N][ng] in/out interior-point state (NULL: none) */
int oracle_solve_batch_ws(int model_id, int N, int B, const real *mp, const real *opt, const real *W,
                          const real *We, const real *x0, const real *yref, const real *yref_e,
                          const real *params, real *X, real *U, int *status, real *stats, int nthreads,
                          real *S, real *L) {
    int nx, nu, np, ng;
    if (oracle_dims(model_id, &nx, &nu, &np, &ng)) return -1;
    int ny, nye;
    oracle_output_dims(model_id, &ny, &nye);
    const size_t syr = opt[OP_YREF_PER_STAGE] != 0 ? (size_t)N * ny : (size_t)ny;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
    for (int b = 0; b < B; b++) {
        status[b] = oracle_solve_ws(model_id, N, mp, opt, W, We, x0 + (size_t)b * nx, yref + b * syr,
                                    yref_e + (size_t)b * nye, params + (size_t)b * (N + 1) * np,
                                    X + (size_t)b * (N + 1) * nx, U + (size_t)b * N * nu,
                                    stats ? stats + (size_t)b * 4 : NULL,
                                    S ? S + (size_t)b * N * ng : NULL, L ? L + (size_t)b * N * ng : NULL);
    }
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* shift a warm start left by `shift` stages (solver.py:304-322): states 1..N-shift
 * take the old states shift+1..N, inputs 0..N-shift-1 take the old shift..N-1,
 * the newly exposed tail of the contact forces is zero-filled (solver.py:320); the state tail and the
 * tail of the first nu_keep inputs (the whole-body model's accelerations: solver.py:316 writes
 * [:, :n_warm_start] only) keep their previous values (repeat_last=False, solver.py:328). */
void oracle_shift_warm_start(int nx, int nu, int nu_keep, int N, int B, int shift, real *X, real *U) {
    if (shift <= 0) return;
    if (shift > N) shift = N;
    const int nw = N - shift;
    for (int b = 0; b < B; b++) {
        real *x = X + (size_t)b * (N + 1) * nx, *u = U + (size_t)b * N * nu;
        memmove(x + nx, x + (size_t)(shift + 1) * nx, sizeof(real) * nw * nx);
        memmove(u, u + (size_t)shift * nu, sizeof(real) * nw * nu);
        for (int k = nw; k < N; k++)
            for (int i = nu_keep; i < nu; i++) u[(size_t)k * nu + i] = 0;
    }
}

/* tracking error (Behavior_Cloning/utils/data_collection_force_perturbation.py:138-156):
 * err[b][t] = || S[b][t][1:] - Snom[t][1:] ||_2  (column 0 = phase is skipped). */
void oracle_tracking_error(int B, int T, int ns, const real *S, const real *Snom, real *err) {
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T; t++) {
            real acc = 0;
            for (int i = 1; i < ns; i++) {
                const real e = S[((size_t)b * T + t) * ns + i] - Snom[(size_t)t * ns + i];
                acc += e * e;
            }
            err[(size_t)b * T + t] = sqrt(acc);
        }
}
