"""Seeded synthetic inputs for the BASELINE.json configurations.

No files, no network: everything is drawn from numpy `default_rng(seed)`.
  config 1  double integrator  nx=4  nu=2  N=20  B=1     (plumbing)
  config 2  centroidal quadruped nx=12 nu=12 N=50 B=1024 (headline)
  config 3  whole-body quadruped nx=42 nu=30 N=30 B=8192 (BASELINE configs[2])
Weights are the reference's Go2 trot cost (mpc_controller/config/quadruped/mpc_cost.py:26-57),
contact flags come from the trot gait table (contact_planner.py:121-134) at a random phase,
base references from the velocity-tracking reference generator (mpc.py:210-272).
Mass / inertia / hip offsets are declared Go2-like constants (not in the reference).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict

import numpy as np

from .config import get_quadruped_config
from .contact_planner import ContactPlanner
from .references import base_ref_vel_tracking

FEET = ("FL_foot", "FR_foot", "RL_foot", "RR_foot")
HIP_OFFSETS = np.array([[0.19, 0.14, 0.0], [0.19, -0.14, 0.0], [-0.19, 0.14, 0.0], [-0.19, -0.14, 0.0]])

MODEL_DOUBLE_INTEGRATOR = 0
MODEL_CENTROIDAL = 1
MODEL_WHOLEBODY = 2
MODEL_DIMS = {MODEL_DOUBLE_INTEGRATOR: dict(nx=4, nu=2, np=0, ng=4, ny=6, ny_e=4),
              MODEL_CENTROIDAL: dict(nx=12, nu=12, np=16, ng=16, ny=24, ny_e=12),
              MODEL_WHOLEBODY: dict(nx=42, nu=30, np=20, ng=16, ny=90, ny_e=66)}

# model parameter vector (shared layout with include/nmpc.h: NMPC_MP_*)
MP_NAMES = ("dt", "mass", "Ixx", "Iyy", "Izz", "gz", "mu", "umax",
            "p_gain", "hipx", "hipy", "lhip", "l1", "l2", "res0", "res1")
# declared penalty weights of the whole-body model's stance and momentum-consistency residuals
W_CONTACT = 50.0
W_CONSISTENCY = 100.0


def model_params(**kw) -> np.ndarray:
    d = dict(dt=0.02, mass=15.0, Ixx=0.11, Iyy=0.27, Izz=0.33, gz=-9.81, mu=0.8, umax=0.0,
             p_gain=50.0, hipx=0.19, hipy=0.047, lhip=0.095, l1=0.213, l2=0.213, res0=0.0, res1=0.0)
    d.update(kw)
    return np.array([d[k] for k in MP_NAMES], dtype=np.float64)


@dataclass
class Workload:
    """One batch of independent NMPC problems (all arrays float64, batch-major)."""
    model_id: int
    N: int
    mp: np.ndarray            # [8]
    W: np.ndarray             # [nx+nu]
    W_e: np.ndarray           # [nx]
    x0: np.ndarray            # [B, nx]
    yref: np.ndarray          # [B, N, nx+nu] per stage, or [B, nx+nu] stage-constant (solver.py:169)
    yref_e: np.ndarray        # [B, nx]
    params: np.ndarray        # [B, N+1, np]
    X: np.ndarray             # [B, N+1, nx] warm start
    U: np.ndarray             # [B, N, nu]   warm start
    meta: Dict = field(default_factory=dict)

    @property
    def B(self) -> int:
        return self.x0.shape[0]


def double_integrator(B: int = 1, N: int = 20, seed: int = 0, umax: float = 0.0) -> Workload:
    """Config 1: x=[p(2),v(2)], u=a(2), dt=0.05, W=diag(10,10,1,1,.1,.1), W_e = 10 Q."""
    rng = np.random.default_rng(seed)
    x0 = rng.uniform(-1, 1, (B, 4))
    W = np.array([10.0, 10.0, 1.0, 1.0, 0.1, 0.1])
    X = np.repeat(x0[:, None, :], N + 1, axis=1)
    return Workload(MODEL_DOUBLE_INTEGRATOR, N, model_params(dt=0.05, umax=umax), W, 10.0 * W[:4], x0,
                    np.zeros((B, N, 6)), np.zeros((B, 4)), np.zeros((B, N + 1, 0)), X,
                    np.zeros((B, N, 2)), dict(config="1: double integrator"))


def centroidal_trot(B: int = 1024, N: int = 50, seed: int = 0, warm: str = "stand") -> Workload:
    """Config 2: centroidal quadruped, trot, T=1.0 s.

    warm = "stand": X held at x0, every stance foot carries an equal share of the weight
    (a feasible interior start); warm = "zero": U = 0 (first-solve case, solver.py:320 tail).
    """
    rng = np.random.default_rng(seed)
    gait, opt, cost = get_quadruped_config("trot", "go2")
    T = opt.time_horizon
    dt = T / N
    mp = model_params(dt=dt)
    planner = ContactPlanner(FEET, dt, gait)

    x0 = np.zeros((B, 12))
    x0[:, 0:2] = rng.normal(0, 0.05, (B, 2))
    x0[:, 2] = 0.30 + rng.normal(0, 0.02, B)
    x0[:, 3:6] = rng.normal(0, 0.1, (B, 3))
    x0[:, 6:9] = rng.normal(0, 0.2, (B, 3))
    x0[:, 9:12] = rng.normal(0, 0.3, (B, 3))
    v_des = np.stack([rng.uniform(0, 0.3, B), rng.uniform(-0.1, 0.1, B), np.zeros(B)], axis=1)
    i_node = rng.integers(0, planner.nodes_per_cycle, B)

    contacts = planner.get_contacts_batch(i_node, N + 1).astype(np.float64)   # [B,4,N+1]
    feet = x0[:, None, :3] * [1, 1, 0] + HIP_OFFSETS[None] + rng.normal(0, 0.02, (B, 4, 3))
    feet[:, :, 2] = 0.0
    params = np.zeros((B, N + 1, 16))
    params[:, :, :4] = np.moveaxis(contacts, 1, 2)
    params[:, :, 4:] = feet.reshape(B, 1, 12)

    # Per-stage output reference: the base part is stage-constant as in the reference
    # (solver.py:169); the force part is the gravity-compensating share of each stance foot
    # [decl]: without the reference's whole-body kinematic cost nothing else holds the base up.
    n_stance = np.maximum(contacts[:, :, :N].sum(1), 1.0)                      # [B,N]
    fz_share = np.moveaxis(contacts[:, :, :N], 1, 2) * ((-mp[5] * mp[1]) / n_stance)[:, :, None]
    yref = np.zeros((B, N, 24))
    yref[:, :, 14::3] = fz_share
    yref_e = np.zeros((B, 12))
    base = np.zeros((B, 12))
    for b in range(B):
        q = np.zeros(18)
        q[:6] = x0[b, :6]
        ref_state = np.zeros(12)
        ref_state[:2] = x0[b, :2]
        ref_state[3] = x0[b, 3]
        base[b], yref_e[b] = base_ref_vel_tracking(q, v_des[b], np.zeros(3), ref_state, T,
                                                   gait.nom_height)
    yref[:, :, :12] = base[:, None, :]
    W = np.concatenate([cost.W_base, cost.W_cnt_f_reg.ravel()])
    X = np.repeat(x0[:, None, :], N + 1, axis=1)
    U = np.zeros((B, N, 12))
    if warm == "stand":
        U[:, :, 2::3] = fz_share
    return Workload(MODEL_CENTROIDAL, N, mp, W, cost.W_e_base.copy(), x0, yref, yref_e, params, X, U,
                    dict(config="2: centroidal trot", v_des=v_des, i_node=i_node,
                         reg=cost.reg_eps, reg_e=cost.reg_eps_e))


def wholebody_weights(cost, w_contact: float = W_CONTACT, w_consistency: float = W_CONSISTENCY, w_foot_displacement: float = 0.0):
    """W[90], W_e[66] of the whole-body model in the residual order of include/nmpc.h: base, joint, acc, swing,
    f_reg, contact, consistency, foot placement / base, joint, swing, contact, consistency, foot placement
    (solver.py:108-141).  w_foot_displacement: the pos_cost weight on x, y of every foot -- 0 outside the reference's
    contact-restricted mode, `cost.W_foot_displacement[0]` inside it (solver.py:131-137)."""
    W = np.concatenate([cost.W_base, cost.W_joint, cost.W_acc, cost.W_swing, np.asarray(cost.W_cnt_f_reg).ravel(),
                        np.full(12, w_contact), np.full(6, w_consistency), np.full(8, w_foot_displacement)])
    W_e = np.concatenate([cost.W_e_base, cost.W_e_joint, cost.W_swing, np.full(12, w_contact),
                          np.full(6, w_consistency), np.full(8, w_foot_displacement)])
    return W, W_e


def anchor_plane_points(contacts: np.ndarray, feet_w: np.ndarray, height_offset: float = 0.0) -> np.ndarray:
    """Per-node plane points of the four feet, [N+1, 4, 3]: (0, 0, height_offset) everywhere
    (`init_contacts_parameters`, solver.py:212-225), then a foot that is in contact at node 0 keeps its current
    position up to its next swing node (`setup_initial_feet_pos`, solver.py:194-210, including its `argmin`: a foot
    in contact over the whole window is not anchored).  contacts: [4, N+1] flags, feet_w: [4, 3]."""
    n1 = contacts.shape[1]
    pp = np.zeros((n1, 4, 3))
    pp[:, :, 2] = height_offset
    for f in range(4):
        if contacts[f, 0]:
            next_swing = int(np.argmin(contacts[f]))
            pp[:next_swing, f, :] = feet_w[f]
    return pp


def wholebody_trot(B: int = 8192, N: int = 30, seed: int = 0, sigma_joint: float = 0.2, foot_placement: float = 0.0) -> Workload:
    """BASELINE configs[2]: whole-body 18-DoF quadruped, nx = 42, nu = 30, trot, T = 1.0 s.

    x0: base as config 2, joints q_home + N(0, sigma_joint) (DAgger/cfgs: sigma_joint_pos 0.2), joint rates
    N(0, 0.2), momentum slots consistent with (q, v) as the reference's `pin_data.hg` (solver.py:187).  Warm start:
    the state held, a = 0, every stance foot carrying an equal share of the weight.
    foot_placement > 0: the reference's contact-restricted cost (pos_cost, solver.py:128-137) with that weight on x, y of
    every foot, against a synthetic plan -- the nominal foothold under each hip carried along with the commanded velocity."""
    from . import wholebody as wb
    rng = np.random.default_rng(seed)
    gait, opt, cost = get_quadruped_config("trot", "go2")
    T = opt.time_horizon
    dt = T / N
    mp = model_params(dt=dt)
    planner = ContactPlanner(FEET, dt, gait)
    d = MODEL_DIMS[MODEL_WHOLEBODY]

    q0 = np.zeros((B, 18)); v0 = np.zeros((B, 18))
    q0[:, 0:2] = rng.normal(0, 0.05, (B, 2))
    q0[:, 2] = 0.30 + rng.normal(0, 0.02, B)
    q0[:, 3:6] = rng.normal(0, 0.1, (B, 3))
    q0[:, 6:] = wb.Q_HOME + rng.normal(0, sigma_joint, (B, 12))
    v0[:, 0:3] = rng.normal(0, 0.2, (B, 3))
    v0[:, 3:6] = rng.normal(0, 0.3, (B, 3))
    v0[:, 6:] = rng.normal(0, 0.2, (B, 12))
    v_des = np.stack([rng.uniform(0, 0.3, B), rng.uniform(-0.1, 0.1, B), np.zeros(B)], axis=1)
    i_node = rng.integers(0, planner.nodes_per_cycle, B)
    contacts = planner.get_contacts_batch(i_node, N + 1).astype(np.float64)      # [B,4,N+1]

    inertia = mp[2:5]
    from .mpc import base_ref_vel_tracking_batch                 # the vectorised restatement of mpc.py:210-272
    x0 = np.concatenate([q0, v0, wb.centroidal_momentum_batch(q0, v0, mp[1], inertia)], axis=1)
    params = np.zeros((B, N + 1, d["np"]))
    params[:, :, 0:4] = np.moveaxis(contacts, 1, 2)
    params[:, :, 4:8] = 1.0 - params[:, :, 0:4]                                   # peak = 1 - contact (contact_planner.py:136-149)
    feet_w = wb.feet_position_w_batch(q0)
    # anchor_plane_points for the batch: a foot in contact at node 0 keeps its position up to its first swing node
    first_swing = np.argmin(contacts, axis=2)                                      # [B, 4]; 0 if it never swings (nothing anchored)
    anchored = (np.arange(N + 1)[None, None, :] < first_swing[:, :, None]) & (contacts[:, :, :1] > 0.5)     # [B, 4, N+1]
    pp = np.where(anchored[..., None], feet_w[:, :, None, :], 0.0)                 # [B, 4, N+1, 3]
    params[:, :, 8:] = np.moveaxis(pp, 1, 2).reshape(B, N + 1, 12)
    ref_state = np.zeros((B, 12))
    ref_state[:, :2] = q0[:, :2]
    ref_state[:, 3] = q0[:, 3]
    base, base_e = base_ref_vel_tracking_batch(q0, v_des, np.zeros((B, 3)), ref_state, T, gait.nom_height)
    yref = np.zeros((B, N, d["ny"]))
    yref_e = np.zeros((B, d["ny_e"]))
    yref[:, :, 0:12] = base[:, None, :]
    yref_e[:, 0:12] = base_e
    yref[:, :, 12:24] = wb.Q_HOME                  # joint reference = nominal pose, zero rates (solver.py:175-177)
    yref[:, :, 48:52] = gait.step_height           # swing-height reference (solver.py:170)
    yref_e[:, 12:24] = wb.Q_HOME
    yref_e[:, 36:40] = gait.step_height
    W, W_e = wholebody_weights(cost, w_foot_displacement=foot_placement)
    if foot_placement > 0.0:
        cy, sy = np.cos(q0[:, 3]), np.sin(q0[:, 3])
        hip_w = q0[:, None, :2] + np.stack([cy[:, None] * HIP_OFFSETS[None, :, 0] - sy[:, None] * HIP_OFFSETS[None, :, 1],
                                            sy[:, None] * HIP_OFFSETS[None, :, 0] + cy[:, None] * HIP_OFFSETS[None, :, 1]], axis=-1)   # [B, 4, 2]
        t_k = (np.arange(N + 1) + 0.0) * dt
        plan = hip_w[:, None] + v_des[:, None, None, :2] * t_k[None, :, None, None]                                       # [B, N+1, 4, 2]
        yref[:, :, 82:90] = plan[:, 1:].reshape(B, N, 8)          # stage k refers to the location at node k + 1 (solver.py:272)
        yref_e[:, 58:66] = plan[:, -1].reshape(B, 8)
    X = np.repeat(x0[:, None, :], N + 1, axis=1)
    U = np.zeros((B, N, d["nu"]))
    n_stance = np.maximum(contacts[:, :, :N].sum(1), 1.0)
    U[:, :, 20::3] = np.moveaxis(contacts[:, :, :N], 1, 2) * ((-mp[5] * mp[1]) / n_stance)[:, :, None]
    # force reference = gravity share of each stance foot [decl], as in config 2: with a zero reference the
    # regularisation alone makes sagging cheaper than standing
    yref[:, :, 52:64] = U[:, :, 18:30]
    return Workload(MODEL_WHOLEBODY, N, mp, W, W_e, x0, yref, yref_e, params, X, U,
                    dict(config="3: whole-body trot", v_des=v_des, i_node=i_node,
                         reg=cost.reg_eps, reg_e=cost.reg_eps_e))


def quadruped_tree(seed: int = 0, perturb: float = 0.0) -> Dict[str, np.ndarray]:
    """A declared quadruped-shaped tree of 1-DoF joints for the torque layer (include/nmpc_torque.h) -- NOT the
    reference's URDF, which is not in the image: three prismatic + three revolute virtual joints (yaw, pitch,
    roll: the reference's state order, dynamics.py:146-148), a 6.9 kg trunk on the last of them, four legs
    [FL, FR, RL, RR] of hip abduction (x), hip flexion (y), knee (y) with point feet.  perturb > 0 tilts axes and
    placements randomly so that tests do not only see axis-aligned geometry.  Keys = BatchedTorqueLayer arguments."""
    rng = np.random.default_rng(seed)
    cols = {k: [] for k in ("parent", "joint_type", "axis", "placement_R", "placement_p", "mass", "com", "inertia")}

    def rotation(axis, angle):
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)

    def add(par, typ, ax, p, m_, c, i6):
        ax = np.asarray(ax, float) + perturb * rng.standard_normal(3)
        tilt = rng.standard_normal(3)
        Rf = rotation(tilt / np.linalg.norm(tilt), perturb * rng.standard_normal()) if perturb else np.eye(3)
        for k, val in zip(cols, (par, typ, ax / np.linalg.norm(ax), Rf, np.asarray(p, float), m_, np.asarray(c, float), np.asarray(i6, float))):
            cols[k].append(val)
        return len(cols["parent"]) - 1
    j = -1
    for typ, ax in ((1, (1, 0, 0)), (1, (0, 1, 0)), (1, (0, 0, 1)), (0, (0, 0, 1)), (0, (0, 1, 0))):
        j = add(j, typ, ax, (0, 0, 0), 0.0, (0, 0, 0), (0, 0, 0, 0, 0, 0))
    trunk = add(j, 0, (1, 0, 0), (0, 0, 0), 6.9, (0.02, 0.0, -0.005), (0.025, 1e-4, 2e-4, 0.098, 1e-5, 0.107))
    foot_joint, foot_offset = [], []
    for sx, sy in ((1, 1), (1, -1), (-1, 1), (-1, -1)):
        hip = add(trunk, 0, (1, 0, 0), (0.19 * sx, 0.047 * sy, 0.0), 0.68, (-0.005 * sx, 0.001 * sy, 0.0), (4.9e-4, 0, 0, 6.4e-4, 0, 5.7e-4))
        thigh = add(hip, 0, (0, 1, 0), (0.0, 0.095 * sy, 0.0), 1.15, (-0.004, -0.016 * sy, -0.033), (5.8e-3, 0, 3e-4, 5.6e-3, 0, 1.0e-3))
        calf = add(thigh, 0, (0, 1, 0), (0.0, 0.0, -0.213), 0.19, (0.006, 0.0, -0.13), (2.4e-3, 0, 0, 2.4e-3, 0, 4e-5))
        foot_joint.append(calf); foot_offset.append((0.0, 0.0, -0.213))
    out = {k: np.asarray(v) for k, v in cols.items()}
    out.update(foot_joint=np.asarray(foot_joint), foot_offset=np.asarray(foot_offset, float), n_actuated=12,
               gravity=np.array([0.0, 0.0, -9.81]))
    return out
