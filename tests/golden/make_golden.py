"""Generate golden vectors from the reference's own host-side helpers.

Run in the build container only (needs /root/reference; never on the GPU box):
    python tests/golden/make_golden.py
Outputs small data fixtures next to this file (inputs + expected outputs, no reference code).

What is pinned: the *inputs* of the NMPC solve and its post-processing -- configuration constants,
gait/contact tables, base references, Hermite up-sampling, Euler-rate maps, Raibert foot targets and
the tracking-error / OOD selection.  The solve itself cannot be pinned: it runs inside acados/HPIPM
through the empty `contact_tamp` submodule -> "parity unpinned".

The reference modules import third-party packages that are absent from this image (pinocchio,
typeguard, contact_tamp, mj_pin, hydra, omegaconf, h5py, mujoco, wandb).  None of them takes part in the
arithmetic pinned here except `pin.rpy.rpyToMatrix`, which is provided below from its published
definition R = Rz(yaw) Ry(pitch) Rx(roll); everything else is an inert placeholder module.
"""
from __future__ import annotations

import enum
import importlib.abc
import importlib.machinery
import json
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------ placeholder modules
class _Anything:
    """Inert stand-in: any attribute, call, subclass or decorator use works and does nothing."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()

    def __mro_entries__(self, bases):
        return (object,)


class _StubLoader(importlib.abc.Loader):
    def create_module(self, spec):
        mod = types.ModuleType(spec.name)
        mod.__path__ = []

        def _attr(name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Anything()

        mod.__getattr__ = _attr
        return mod

    def exec_module(self, module):
        pass


class _StubFinder(importlib.abc.MetaPathFinder):
    ROOTS = ("pinocchio", "typeguard", "contact_tamp", "mj_pin", "hydra", "omegaconf", "h5py",
             "mujoco", "wandb", "tkinter")

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, _StubLoader(), is_package=True)
        return None


def _install_stubs():
    sys.meta_path.insert(0, _StubFinder())
    import pinocchio  # noqa: F401  (placeholder)
    import typeguard

    def rpy_to_matrix(*rpy):
        r, p, y = (rpy[0] if len(rpy) == 1 else rpy)
        cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
        Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
        Ry = np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]])
        Rx = np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]])
        return Rz @ Ry @ Rx

    rpy = types.ModuleType("pinocchio.rpy")
    rpy.rpyToMatrix = rpy_to_matrix
    sys.modules["pinocchio"].rpy = rpy
    sys.modules["pinocchio.rpy"] = rpy
    typeguard.typechecked = lambda cls: cls

    class HPIPM_MODE(enum.Enum):
        speed_abs = 0
        speed = 1
        balance = 2
        robust = 3

    import contact_tamp.traj_opt_acados.interface.acados_helper as ah
    ah.HPIPM_MODE = HPIPM_MODE
    ah.AcadosSolverHelper = type("AcadosSolverHelper", (), {})
    import contact_tamp.traj_opt_acados.models.floating_base_dynamics as fb
    fb.FloatingBaseDynamics = type("FloatingBaseDynamics", (), {})
    import mj_pin.abstract as mja
    mja.PinController = type("PinController", (), {})
    mja.Controller = type("Controller", (), {})
    mja.DataRecorder = type("DataRecorder", (), {})


def _jsonable(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, enum.Enum):
        return v.name
    if isinstance(v, tuple):
        return list(v)
    return v


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    rng = np.random.default_rng(20240)

    # ---- 1. configuration constants -------------------------------------------------------
    from mpc_controller.config.quadruped.utils import get_quadruped_config
    cfg_out = {}
    for gait in ("trot", "slow_trot"):
        g, o, c = get_quadruped_config(gait, "go2")
        cfg_out[gait] = dict(
            gait={k: _jsonable(v) for k, v in vars(g).items()},
            opt={**{k: _jsonable(v) for k, v in vars(o).items()},
                 "dt_nodes": o.get_dt_nodes(), "dt_bounds": list(o.get_dt_bounds())},
            cost={k: _jsonable(v) for k, v in vars(c).items()})
    from mpc_controller.config.quadruped.mpc_gait import GaitConfigFactory
    cfg_out["gaits"] = {name: {k: _jsonable(v) for k, v in vars(GaitConfigFactory.get(name)).items()}
                        for name in ("trot", "slow_trot", "jump", "crawl", "pace", "bound")}
    with open(os.path.join(OUT, "config.json"), "w") as f:
        json.dump(cfg_out, f, indent=1, sort_keys=True)

    # ---- 2. gait tables and windows ------------------------------------------------------------
    from mpc_controller.utils.contact_planner import ContactPlanner, RaiberContactPlanner
    feet = ["FL_foot", "FR_foot", "RL_foot", "RR_foot"]
    cp = {}
    cases = [("trot", 0.04), ("trot", 0.02), ("slow_trot", 0.04), ("crawl", 0.04), ("crawl", 0.02),
             ("pace", 0.05), ("bound", 0.02)]
    queries = [(0, 26), (3, 26), (11, 26), (12, 51), (37, 51), (5, 7), (123456, 31)]
    for ci, (gait, dt) in enumerate(cases):
        pl = ContactPlanner(feet, dt, GaitConfigFactory.get(gait))
        cp[f"c{ci}_npc"] = np.int64(pl.nodes_per_cycle)
        cp[f"c{ci}_gait_sequence"] = pl.gait_sequence
        cp[f"c{ci}_switch_cnt"] = pl.switch_cnt
        cp[f"c{ci}_peak_swing"] = pl.peak_swing
        for qi, (i_node, n) in enumerate(queries):
            cp[f"c{ci}_q{qi}_contacts"] = pl.get_contacts(i_node, n).copy()
            cp[f"c{ci}_q{qi}_peaks"] = pl.get_peaks(i_node, n).copy()
            mk, bk = pl.get_make_break_contacts(i_node, n)
            cp[f"c{ci}_q{qi}_make"] = mk.copy()
            cp[f"c{ci}_q{qi}_break"] = bk.copy()
    cp["cases"] = np.array([f"{g}:{dt}" for g, dt in cases])
    cp["queries"] = np.array(queries, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "contact_planner.npz"), **cp)

    # ---- 3. Raibert foot targets -----------------------------------------------------------------
    rb = {}
    hips = np.array([[0.1934, 0.142, 0.0], [0.1934, -0.142, 0.0], [-0.1934, 0.142, 0.0], [-0.1934, -0.142, 0.0]])
    n_rb = 6
    rb["hips"] = hips
    rb_in = np.zeros((n_rb, 17))
    for i in range(n_rb):
        pl = RaiberContactPlanner(feet, 0.04, GaitConfigFactory.get("trot"), hips.copy(),
                                  y_offset=0.02, x_offset=0.04, foot_size=0.0085, cache_cnt=False)
        pos = rng.normal(0, 0.1, 3) + [0, 0, 0.3]
        v_w = rng.normal(0, 0.2, 3)
        rpy = rng.normal(0, 0.2, 3)
        com = pos + rng.normal(0, 0.01, 3)
        v_des = np.array([rng.uniform(0.05, 0.4), rng.uniform(-0.1, 0.1), 0.0])
        w_yaw = rng.uniform(-0.3, 0.3)
        i_node = int(rng.integers(0, 40))
        pl.set_state(pos, v_w, rpy, com, v_des, w_yaw)
        rb_in[i] = np.concatenate([pos, v_w, rpy, com, v_des, [w_yaw, i_node]])
        rb[f"loc{i}"] = pl.get_locations(i_node, 26)
    rb["inputs"] = rb_in
    np.savez_compressed(os.path.join(OUT, "raibert.npz"), **rb)

    # ---- 4. base references, reference integration, Hermite up-sampling -----------------------------
    from mpc_controller.mpc import LocomotionMPC
    n_case = 24
    ref_in = np.zeros((n_case, 18 + 3 + 3 + 12))
    ref_out = np.zeros((n_case, 2, 12))
    inc_out = np.zeros((n_case, 12))
    for i in range(n_case):
        mpc = object.__new__(LocomotionMPC)
        mpc.executor = _Anything()
        mpc.solver = types.SimpleNamespace(config_opt=types.SimpleNamespace(time_horizon=1.0))
        mpc.config_gait = types.SimpleNamespace(nom_height=0.30)
        mpc.height_offset = 0.0 if i % 3 else 0.02
        mpc.velocity_goal = None
        mpc.sim_dt = 1.0e-3
        q = np.zeros(18)
        q[:3] = rng.normal(0, 0.5, 3)
        q[3:6] = rng.normal(0, 0.4, 3)
        if i == 0:
            q[:] = 0.0
            q[2] = 0.3
        v_des = np.array([rng.uniform(-0.1, 0.4), rng.uniform(-0.15, 0.15), 0.0])
        if i == 0:
            v_des = np.array([0.3, 0.0, 0.0])
        if i == 1:
            v_des = np.array([0.15, 0.0, 0.0])
        w_des = np.array([0.0, 0.0, rng.uniform(-0.4, 0.4) if i % 2 else 0.0])
        state = np.zeros(12)
        if i > 1:
            state[:2] = q[:2] + rng.normal(0, 0.05, 2)
            state[3] = q[3] + rng.normal(0, 0.05)
        mpc.v_des, mpc.w_des, mpc.base_ref_vel_tracking = v_des, w_des, state.copy()
        ref_in[i] = np.concatenate([q, v_des, w_des, state])
        b, be = mpc.compute_base_ref_vel_tracking(q)
        ref_out[i, 0], ref_out[i, 1] = b, be
        mpc.increment_base_ref_position()
        inc_out[i] = mpc.base_ref_vel_tracking
    hm = {}
    mpc = object.__new__(LocomotionMPC)
    mpc.executor, mpc.velocity_goal = _Anything(), None
    for name, (n_nodes, n_interp, d) in {"a": (25, 1000, 18), "b": (50, 1000, 12), "c": (10, 37, 3)}.items():
        mpc.n_interp_plan = n_interp
        dt_sol = np.full(n_nodes, 1.0 / n_nodes) if name != "c" else rng.uniform(0.02, 0.07, n_nodes)
        t = np.concatenate(([0.0], np.cumsum(dt_sol)))
        pos, vel = rng.normal(0, 1, (n_nodes + 1, d)), rng.normal(0, 1, (n_nodes + 1, d))
        acc = rng.normal(0, 1, (n_nodes, d))
        ip, iv = mpc.interpolate_trajectory_with_derivatives(t, pos, vel, acc)
        hm.update({f"{name}_t": t, f"{name}_pos": pos, f"{name}_vel": vel, f"{name}_acc": acc,
                   f"{name}_ipos": ip, f"{name}_ivel": iv, f"{name}_n": np.int64(n_interp)})
        mpc.sim_dt = 1.0e-3
    mpc.config_opt = types.SimpleNamespace(time_horizon=1.0, n_nodes=25)
    hm["id_repeat_1000_25"] = np.int32(np.linspace(0, 1, 1000) * (25 - 1))
    np.savez_compressed(os.path.join(OUT, "references.npz"), ref_in=ref_in, ref_out=ref_out,
                        inc_out=inc_out, **hm)

    # ---- 5. Euler-rate maps -----------------------------------------------------------------------
    from mpc_controller.utils.transform import (euler_derivative_to_local_angular,
                                                 local_angular_to_euler_derivative)
    ypr = rng.normal(0, 0.5, (16, 3))
    w = rng.normal(0, 1.0, (16, 3))
    np.savez_compressed(
        os.path.join(OUT, "transform.npz"), ypr=ypr, w=w,
        to_euler=np.array([local_angular_to_euler_derivative(a, b) for a, b in zip(ypr, w)]),
        to_local=np.array([euler_derivative_to_local_angular(a, b) for a, b in zip(ypr, w)]))

    # ---- 6. tracking error -> OOD selection ---------------------------------------------------------
    from Behavior_Cloning.utils.data_collection_force_perturbation import DataCollection
    T, ns = 60, 44
    t_nom = np.round(np.arange(T) * 1e-3 + 0.5, 4)
    s_nom = rng.normal(0, 1, (T, ns))
    picked = []

    class _Sink:
        def append(self, states, vc_goals, cc_goals, actions):
            picked.append(np.asarray(states[0]))

    dc = object.__new__(DataCollection)
    dc.ood_database = _Sink()
    n_pert = 5
    s_pert = s_nom[None] + rng.normal(0, 0.62, (n_pert, T, ns))
    s_pert[:, :, 0] += 50.0  # phase column must be ignored
    t_pert = np.tile(t_nom, (n_pert, 1))
    t_pert[3, 40:] += 7.0    # samples with no nominal counterpart are skipped
    sel = np.zeros((n_pert, T), dtype=bool)
    with tempfile.TemporaryDirectory() as d:
        np.savez(os.path.join(d, "traj_nominal_0.npz"), time=t_nom, state=s_nom)
        for b in range(n_pert):
            picked.clear()
            goals = np.zeros((T, 3))
            dc.save_ood_val_set_l2_distance(d, s_pert[b], goals, goals, np.zeros((T, 12)), t_pert[b],
                                            f"traj_pert_{b}.npz")
            for s in picked:
                sel[b, int(np.argmax(np.all(s_pert[b] == s, axis=1)))] = True
    np.savez_compressed(os.path.join(OUT, "tracking_error.npz"), t_nom=t_nom, s_nom=s_nom, s_pert=s_pert,
                        t_pert=t_pert, ood_selected=sel, threshold=np.float64(4.0))
    print("golden vectors written to", OUT, "| OOD selected per rollout:", sel.sum(1))


if __name__ == "__main__":
    main()
