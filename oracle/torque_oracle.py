"""CPU restatement of the reference's torque layer -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (iterative_learning_nmpc_amd/) never does.

What it restates, in numpy float64:
  * `QuadrupedDynamics.id_torques` (mpc_controller/utils/dynamics.py:136-163):
        tau = rnea(model, q, v, a)[-nu:]  -  sum_feet  f_ee . J_ee[:3, -nu:]
    with J_ee the frame Jacobian in LOCAL_WORLD_ALIGNED axes, i.e. f_ee is a WORLD-frame force acting
    at the foot point.  Subtracting J^T f is the same as applying f as an external force inside the
    recursion, which is what `id_torques` below does,
  * `LocomotionMPC._compute_pd_torques` (mpc_controller/mpc.py:592-599).
PARITY UNPINNED: the reference evaluates this with pinocchio 3.3.1 (environment.yml:255), which is not
installed, on a URDF that is not in the image; its tests hold no vectors for it.  `rnea` restates the
published recursive Newton-Euler algorithm (Featherstone, Rigid Body Dynamics Algorithms, table 5.1) for
a tree of 1-DoF joints -- the reference's state [px, py, pz, yaw, pitch, roll, joints] with q_dot = v is such
a tree: three prismatic and three revolute virtual joints in front of the legs.  tests/test_torque_oracle.py
checks it against a derivation that shares nothing with the recursion: the Lagrangian equations of motion
built from forward kinematics and geometric Jacobians (`lagrangian_torques`).

Model (shared with include/nmpc_torque.h): joint i has a parent (-1 = world), a type (0 revolute, 1
prismatic), a unit axis in its own frame, a fixed placement (R_fix, p_fix) of the joint frame in the parent
frame (x_parent = R_fix x_joint + p_fix), and carries a body (mass, centre of mass, inertia about the centre
of mass in body axes as xx, xy, xz, yy, yz, zz).  Parents come before children."""
import numpy as np


def _skew(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])


def _axis_rotation(axis, angle):
    K = _skew(axis)
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def _inertia_matrix(i6):
    xx, xy, xz, yy, yz, zz = i6
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])


class TreeModel:
    def __init__(self, parent, jtype, axis, placement_R, placement_p, mass, com, inertia, foot_joint, foot_offset,
                 n_actuated, gravity=(0.0, 0.0, -9.81)):
        self.parent = np.asarray(parent, int); self.jtype = np.asarray(jtype, int)
        self.axis = np.asarray(axis, float); self.R_fix = np.asarray(placement_R, float); self.p_fix = np.asarray(placement_p, float)
        self.mass = np.asarray(mass, float); self.com = np.asarray(com, float); self.inertia = np.asarray(inertia, float)
        self.foot_joint = np.asarray(foot_joint, int); self.foot_offset = np.asarray(foot_offset, float)
        self.n, self.nu, self.gravity = len(self.parent), int(n_actuated), np.asarray(gravity, float)
        assert all(p < i for i, p in enumerate(self.parent)), "parents come before children"

    @classmethod
    def from_arrays(cls, t):
        """From the argument dictionary of the product's BatchedTorqueLayer (the tests hand both the same arrays)."""
        return cls(t["parent"], t["joint_type"], t["axis"], t["placement_R"], t["placement_p"], t["mass"], t["com"], t["inertia"],
                   t["foot_joint"], t["foot_offset"], t["n_actuated"], t.get("gravity", (0.0, 0.0, -9.81)))

    def joint_transform(self, i, qi):
        """(R, p): x_parent = R x_child + p."""
        if self.jtype[i] == 0:
            return self.R_fix[i] @ _axis_rotation(self.axis[i], qi), self.p_fix[i].copy()
        return self.R_fix[i].copy(), self.p_fix[i] + self.R_fix[i] @ (self.axis[i] * qi)

    def forward_kinematics(self, q):
        """World pose (R_w, p_w) of every joint frame."""
        Rw, pw = [None] * self.n, [None] * self.n
        for i in range(self.n):
            R, p = self.joint_transform(i, q[i])
            if self.parent[i] < 0:
                Rw[i], pw[i] = R, p
            else:
                Rw[i], pw[i] = Rw[self.parent[i]] @ R, pw[self.parent[i]] + Rw[self.parent[i]] @ p
        return Rw, pw


def id_torques(m: TreeModel, q, v, a, f_world):
    """dynamics.py:136-163 for one sample: all n generalised forces (the reference keeps the last nu)."""
    n = m.n
    w = np.zeros((n, 3)); vo = np.zeros((n, 3)); dw = np.zeros((n, 3)); dvo = np.zeros((n, 3))
    Rw = [None] * n; Rl = [None] * n; pl = [None] * n
    fn = np.zeros((n, 3)); fl = np.zeros((n, 3))                 # moment about the body origin, force
    for i in range(n):
        R, p = m.joint_transform(i, q[i])
        Rl[i], pl[i] = R, p
        par = m.parent[i]
        if par < 0:
            w_p = vo_p = dw_p = np.zeros(3); dvo_p = -m.gravity; Rw[i] = R        # gravity as a base acceleration
        else:
            w_p, vo_p, dw_p, dvo_p = w[par], vo[par], dw[par], dvo[par]; Rw[i] = Rw[par] @ R
        w_i = R.T @ w_p; vo_i = R.T @ (vo_p + np.cross(w_p, p))
        dw_i = R.T @ dw_p; dvo_i = R.T @ (dvo_p + np.cross(dw_p, p))
        s_w, s_v = (m.axis[i], np.zeros(3)) if m.jtype[i] == 0 else (np.zeros(3), m.axis[i])
        vj_w, vj_v = s_w * v[i], s_v * v[i]
        # a_i = X a_parent + S qdd + v_i x (S qd)
        dw_i = dw_i + s_w * a[i] + np.cross(w_i, vj_w)
        dvo_i = dvo_i + s_v * a[i] + np.cross(w_i, vj_v) + np.cross(vo_i, vj_w)
        w_i = w_i + vj_w; vo_i = vo_i + vj_v
        w[i], vo[i], dw[i], dvo[i] = w_i, vo_i, dw_i, dvo_i
        # f = I a + v x* (I v), spatial inertia about the body origin
        mass, c, Ic = m.mass[i], m.com[i], _inertia_matrix(m.inertia[i])
        h_l = mass * (vo_i + np.cross(w_i, c)); h_n = Ic @ w_i + np.cross(c, h_l)
        g_l = mass * (dvo_i + np.cross(dw_i, c)); g_n = Ic @ dw_i + np.cross(c, g_l)
        fn[i] = g_n + np.cross(w_i, h_n) + np.cross(vo_i, h_l)
        fl[i] = g_l + np.cross(w_i, h_l)
    for k, j in enumerate(m.foot_joint):                          # world force at the foot point of body j
        l = Rw[j].T @ np.asarray(f_world[k], float)
        fl[j] -= l; fn[j] -= np.cross(m.foot_offset[k], l)
    tau = np.zeros(n)
    for i in range(n - 1, -1, -1):
        tau[i] = m.axis[i] @ (fn[i] if m.jtype[i] == 0 else fl[i])
        par = m.parent[i]
        if par >= 0:
            l_p = Rl[i] @ fl[i]
            fl[par] += l_p; fn[par] += Rl[i] @ fn[i] + np.cross(pl[i], l_p)
    return tau


def id_torques_batch(m, q, v, a, f_world):
    """[B, nu]: the reference's return value, tau[-nu:]."""
    return np.stack([id_torques(m, q[b], v[b], a[b], f_world[b])[-m.nu:] for b in range(len(q))])


def pd_torques(tau_ff, q, v, q_plan, v_plan, kp, kd, nu):
    """mpc.py:592-599."""
    return tau_ff + kp * (q_plan[..., -nu:] - q[..., -nu:]) + kd * (v_plan[..., -nu:] - v[..., -nu:])


# ---------------------------------------------------------------------------------------------------
# Independent derivation for the tests: Lagrange's equations from forward kinematics.
def _mass_matrix_and_potential(m: TreeModel, q):
    Rw, pw = m.forward_kinematics(q)
    n = m.n
    z = [Rw[i] @ m.axis[i] for i in range(n)]                     # world joint axes
    M, V = np.zeros((n, n)), 0.0
    for b in range(n):
        if m.mass[b] == 0.0 and not np.any(m.inertia[b]):
            continue
        pc = pw[b] + Rw[b] @ m.com[b]
        Jv, Jw = np.zeros((3, n)), np.zeros((3, n))
        k = b
        while k >= 0:                                             # joints on the path to the root move body b
            if m.jtype[k] == 0:
                Jw[:, k] = z[k]; Jv[:, k] = np.cross(z[k], pc - pw[k])
            else:
                Jv[:, k] = z[k]
            k = m.parent[k]
        Iw = Rw[b] @ _inertia_matrix(m.inertia[b]) @ Rw[b].T
        M += m.mass[b] * Jv.T @ Jv + Jw.T @ Iw @ Jw
        V -= m.mass[b] * m.gravity @ pc
    return M, V


def lagrangian_torques(m: TreeModel, q, v, a, f_world, h=1e-6):
    """tau = M qdd + (dM/dt) qd - d/dq (1/2 qd' M qd) + dV/dq - sum J_foot' f; derivatives in q by central
    differences of quantities that come from forward kinematics alone."""
    n = m.n
    M, _ = _mass_matrix_and_potential(m, q)
    dM = np.zeros((n, n, n)); dV = np.zeros(n)
    for k in range(n):
        e = np.zeros(n); e[k] = h
        Mp, Vp = _mass_matrix_and_potential(m, q + e); Mm, Vm = _mass_matrix_and_potential(m, q - e)
        dM[k] = (Mp - Mm) / (2 * h); dV[k] = (Vp - Vm) / (2 * h)
    Mdot = np.einsum("kij,k->ij", dM, v)
    tau = M @ a + Mdot @ v - 0.5 * np.einsum("kij,i,j->k", dM, v, v) + dV
    for kf, j in enumerate(m.foot_joint):                         # foot point Jacobian by differences of its position
        def foot(qq):
            Rw, pw = m.forward_kinematics(qq)
            return pw[j] + Rw[j] @ m.foot_offset[kf]
        J = np.zeros((3, n))
        for k in range(n):
            e = np.zeros(n); e[k] = h
            J[:, k] = (foot(q + e) - foot(q - e)) / (2 * h)
        tau -= J.T @ np.asarray(f_world[kf], float)
    return tau
