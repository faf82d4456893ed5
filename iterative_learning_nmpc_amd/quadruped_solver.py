"""`QuadrupedAcadosSolver` on the MI355X C-ABI: the reference's solver facade, same names and shapes.

Drop-in for `mpc_controller/utils/solver.py:15-429` as `LocomotionMPC` uses it:

    ctor(path_urdf, feet_frame_names, config_opt, config_cost, height_offset, print_info, compute_timings)  :18-26
    .dyn.{update_pin, get_feet_position_w, q, v, h, a, feet, name, base_cost, joint_cost, acc_cost, swing_cost}
    .dt_nodes .config_opt .config_cost .restrict_cnt .timings .last_node
    .set_contact_restriction(bool)  .reset()  .update_cost(cfg)  .set_cost_weights()
    .init(i_node, q, v, base_ref, base_ref_e, joint_ref, step_height, cnt_sequence, cnt_locations, swing_peak)  :355-394
    .solve() -> (q_sol[N+1,18], v_sol[N+1,18], a_sol[N,18], f_sol[N,4,3], dt_sol[N])                             :396-429
    .set_max_iter / .set_nlp_tol / .set_qp_tol                                                                  :75-79
    dict views  .states[name][dim, N+1]  .inputs[name][dim, N]  .params[name][dim, N+1]
                .cost_ref[name][dim, N]  .cost_ref_terminal[name][dim]                                          :88-91

Below the class sits `libnmpc_hip.so` (include/nmpc.h, model NMPC_MODEL_WHOLEBODY) where the reference has
`AcadosSolverHelper` -> acados/HPIPM.  The numpy views are the interface, exactly as in the reference; `update_solver`
packs them into the batch-major device tensors of the C-ABI and `solve` unpacks the result.  A leading batch axis
(`batch > 1`) turns every view into `[B, dim, nodes]` and every `init` argument into its batched form.

Declared differences (`UNSUPPORTED` lists what a caller may have switched on and does not get):
  * `path_urdf` is accepted and not read: pinocchio / the URDF are not in the image, the model is the declared
    quadruped of `wholebody.py` (geometry of `workloads.quadruped_tree()`);
  * the names of the views that come from the absent `contact_tamp` are declared here (`NAMES`);
  * contact-restricted mode (`set_contact_restriction(True)`, the Raibert planner): the foot-placement cost `pos_cost`
    (x, y of each foot against the planned location, weight `W_foot_displacement`, solver.py:128-137,272-273) IS part of
    the solved model; the hard patch constraint behind `restrict` / `range_radius` (|p_xy - plane_point_xy| <= cnt_radius
    at touch-down nodes, inside contact_tamp) is NOT -- the views are kept, a warning is issued once;
  * `config_opt.torque_limit` (default True, config_abstract.py:68): joint-torque bounds are state-dependent general
    inequalities tau(q, a, f); the interior point of this build carries input inequalities only -- not enforced, warned;
  * the force regularisation refers to ZERO force as the reference's (solver.py:128-130); `force_reference="gravity_share"`
    regularises to the weight share of the stance feet instead (what the synthetic workloads of BASELINE configs[1] / [2]
    use: with this build's soft stance penalty a zero reference lets the base sag);
  * `warm_start_multipliers` has no counterpart: the interior point cold-starts every SQP iteration; `qp_tol` is
    stored, the interior point runs `max_qp_iter` iterations.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import wholebody as wb
from .profiling import time_fn
from .workloads import MODEL_DIMS, MODEL_WHOLEBODY, W_CONSISTENCY, W_CONTACT, model_params


class _Named:
    def __init__(self, name: str):
        self.name = name


class _PointContact:
    """names of one foot's parameters / cost terms (contact_tamp.PointContact, absent: names declared)"""

    def __init__(self, frame_name: str, model_name: str):
        self.frame_name = frame_name
        for attr in ("active", "plane_point", "plane_normal", "p_gain", "restrict", "range_radius", "peak"):
            setattr(self, attr, _Named(f"{attr}_{frame_name}_{model_name}"))
        self.f_reg = _Named(f"f_reg_{frame_name}_{model_name}")
        self.pos_cost = _Named(f"pos_cost_{frame_name}_{model_name}")
        self.force_name = f"f_{frame_name}_{model_name}"       # solver.py:319


class QuadrupedDynamics:
    """host side of `mpc_controller/utils/dynamics.py:10-106` for the declared model: names of the symbolic
    quantities and the kinematics the controller asks for between solves"""

    def __init__(self, feet_frame_names: List[str], mass: float, inertia, name: str = "go2"):
        self.name = name
        self.q, self.v, self.h, self.a = _Named("q"), _Named("v"), _Named("h"), _Named("a")
        self.base_cost, self.joint_cost = _Named("base_cost"), _Named("joint_cost")       # dynamics.py:35-38
        self.acc_cost, self.swing_cost = _Named("acc_cost"), _Named("sw_cost")
        self.feet = [_PointContact(n, name) for n in feet_frame_names]
        self.mass, self.inertia = float(mass), np.asarray(inertia, float)
        self.nu = wb.N_JOINTS
        self._q = np.zeros(18)
        self._v = np.zeros(18)

    def update_pin(self, q: np.ndarray, v: np.ndarray):                                     # dynamics.py:48-51
        self._q, self._v = np.array(q, float), np.array(v, float)

    def get_feet_position_w(self) -> np.ndarray:                                            # dynamics.py:100-106
        if self._q.ndim == 1:
            return wb.feet_position_w(self._q)
        return np.stack([wb.feet_position_w(q) for q in self._q])

    @property
    def hg(self) -> np.ndarray:                                                             # pin_data.hg, solver.py:187
        if self._q.ndim == 1:
            return wb.centroidal_momentum(self._q, self._v, self.mass, self.inertia)
        return np.stack([wb.centroidal_momentum(q, v, self.mass, self.inertia) for q, v in zip(self._q, self._v)])


class QuadrupedAcadosSolver:
    NAME = "quadruped_solver"

    def __init__(self, path_urdf: str, feet_frame_names: List[str], config_opt, config_cost,
                 height_offset: float = 0., print_info: bool = False, compute_timings: bool = True,
                 batch: int = 1, device="cuda:0", force_reference: str = "zero",
                 w_contact: float = W_CONTACT, w_consistency: float = W_CONSISTENCY, strict: bool = False):
        self.feet_frame_names = list(feet_frame_names)
        assert len(self.feet_frame_names) == 4, "the declared model is a quadruped"
        self.config_opt, self.config_cost = config_opt, config_cost
        self.height_offset, self.print_info = height_offset, print_info
        self.restrict_cnt = False
        self.batch, self.device = int(batch), device
        assert force_reference in ("zero", "gravity_share")
        self.force_reference = force_reference
        self.w_contact, self.w_consistency = float(w_contact), float(w_consistency)
        # what the configuration asks for and this build does not enforce (module docstring); strict: raise instead of warn
        self.strict, self.unsupported = bool(strict), []
        if getattr(self.config_opt, "torque_limit", False):
            self._unsupported("torque_limit", "config_opt.torque_limit is set (the reference's default): joint-torque limits are not "
                              "enforced by this solver (input inequalities only); set torque_limit=False to acknowledge")
        self.dt_nodes = self.config_opt.get_dt_nodes()
        self.enable_time_opt = self.config_opt.enable_time_opt
        assert not self.enable_time_opt, "time optimisation (dt as an input) is not part of the declared model"
        self.mp = model_params(dt=self.dt_nodes, mu=0.8,                                    # mu: solver.py:38
                               p_gain=float(np.asarray(self.config_cost.W_foot_pos_constr_stab)[0]))
        assert np.all(np.asarray(self.config_cost.W_foot_pos_constr_stab) == self.mp[8]), "one Baumgarte gain for all feet"
        self.dyn = QuadrupedDynamics(self.feet_frame_names, self.mp[1], self.mp[2:5])
        self.default_normal = np.array([0., 0., 1.])
        self.compute_timings = compute_timings
        self.timings = defaultdict(list)
        self._dev = None                      # BatchedNmpcSolver, created at the first solve (needs the GPU)
        self._dims = MODEL_DIMS[MODEL_WHOLEBODY]
        self.reset()

    def _unsupported(self, key: str, msg: str):
        if key in self.unsupported:
            return
        self.unsupported.append(key)
        if self.strict:
            raise NotImplementedError(msg)
        import warnings
        warnings.warn(f"QuadrupedAcadosSolver (MI355X): {msg}", UserWarning, stacklevel=3)

    # ------------------------------------------------------------------------------------------ views
    def _shape(self, *s):
        return (self.batch, *s) if self.batch > 1 else tuple(s)

    def _alloc_views(self):
        N = self.config_opt.n_nodes
        z = lambda *s: np.zeros(self._shape(*s))
        d = self.dyn
        self.states = {d.q.name: z(18, N + 1), d.v.name: z(18, N + 1), d.h.name: z(6, N + 1)}
        self.inputs = {d.a.name: z(18, N)}
        self.params, self.cost_ref, self.cost_ref_terminal = {}, {}, {}
        for f in d.feet:
            self.inputs[f.force_name] = z(3, N)
            for attr, dim in (("active", 1), ("plane_point", 3), ("plane_normal", 3), ("p_gain", 1), ("restrict", 1),
                              ("range_radius", 1), ("peak", 1)):
                self.params[getattr(f, attr).name] = z(dim, N + 1)
            self.cost_ref[f.f_reg.name] = z(3, N)
            self.cost_ref[f.pos_cost.name] = z(3, N)
            self.cost_ref_terminal[f.pos_cost.name] = z(3)
        for name, dim in ((d.base_cost.name, 12), (d.joint_cost.name, 24), (d.swing_cost.name, 4)):
            self.cost_ref[name] = z(dim, N)
            self.cost_ref_terminal[name] = z(dim)
        self.cost_ref[d.acc_cost.name] = z(12, N)

    def get_data_template(self) -> Dict[str, Dict[str, np.ndarray]]:
        """name -> array templates of the weights and of the initial state (solver.py:81,113-125,185-187)"""
        d = self.dyn
        W = {d.base_cost.name: np.zeros(12), d.joint_cost.name: np.zeros(24), d.acc_cost.name: np.zeros(12),
             d.swing_cost.name: np.zeros(4)}
        W_e = {d.base_cost.name: np.zeros(12), d.joint_cost.name: np.zeros(24), d.swing_cost.name: np.zeros(4)}
        for f in d.feet:
            W[f.f_reg.name] = np.zeros(3)
            W[f.pos_cost.name], W_e[f.pos_cost.name] = np.zeros(3), np.zeros(3)
        return {"W": W, "W_e": W_e, "x": {d.q.name: np.zeros(18), d.v.name: np.zeros(18), d.h.name: np.zeros(6)}}

    # ------------------------------------------------------------------------------------------ configuration
    def reset(self):                                                                        # solver.py:66-98
        self.last_node = 0
        self.timings = defaultdict(list)
        self._opts = dict(max_iter=self.config_opt.max_iter, max_qp_iter=self.config_opt.max_qp_iter,
                          nlp_tol=self.config_opt.nlp_tol, qp_tol=self.config_opt.qp_tol)
        self.data = self.get_data_template()
        self._alloc_views()
        self.set_cost_weights()
        N = self.config_opt.n_nodes
        self.q_sol_euler = np.zeros(self._shape(N + 1, 18))
        self.v_sol_euler = np.zeros(self._shape(N + 1, 18))
        self.a_sol = np.zeros(self._shape(N, 18))
        self.h_sol = np.zeros(self._shape(N + 1, 6))
        self.f_sol = np.zeros(self._shape(N, 4, 3))
        self.dt_node_sol = np.zeros(self._shape(N))
        if self._dev is not None:
            self._push_opts()

    def set_max_iter(self, n: int):
        self._opts["max_iter"] = int(n); self._push_opts()

    def set_nlp_tol(self, tol: float):
        self._opts["nlp_tol"] = float(tol); self._push_opts()

    def set_qp_tol(self, tol: float):
        self._opts["qp_tol"] = float(tol); self._push_opts()

    def set_warm_start_inner_qp(self, on: bool): pass       # the interior point cold-starts (module docstring)
    def set_warm_start_nlp(self, on: bool): pass

    def _push_opts(self):
        if self._dev is not None:
            o = self._opts
            self._dev.set_max_iter(o["max_iter"]); self._dev.set_max_qp_iter(o["max_qp_iter"])
            self._dev.set_nlp_tol(o["nlp_tol"]); self._dev.set_qp_tol(o["qp_tol"])

    def set_contact_restriction(self, restrict: bool = True):                               # solver.py:100-103
        self.restrict_cnt = restrict
        if restrict and self.config_cost.cnt_radius < 1.0e9:
            self._unsupported("patch_restriction", f"contact restriction: the foot-placement cost (W_foot_displacement = "
                              f"{float(self.config_cost.W_foot_displacement[0]):g}) is solved, the hard patch constraint of radius "
                              f"{self.config_cost.cnt_radius} m behind `restrict` / `range_radius` is not")
        self.set_cost_weights()

    def update_cost(self, config_cost):                                                     # solver.py:105-110
        self.config_cost = config_cost
        self.set_cost_weights()

    def set_cost_weights(self):                                                             # solver.py:112-141
        d, c = self.dyn, self.config_cost
        W, W_e = self.data["W"], self.data["W_e"]
        W_e[d.base_cost.name] = np.array(c.W_e_base); W[d.base_cost.name] = np.array(c.W_base)
        W[d.acc_cost.name] = np.array(c.W_acc); W[d.swing_cost.name] = np.array(c.W_swing)
        W[d.joint_cost.name] = np.array(c.W_joint); W_e[d.joint_cost.name] = np.array(c.W_e_joint)
        W_e[d.swing_cost.name] = np.array(c.W_swing)
        for i, f in enumerate(d.feet):
            W[f.f_reg.name] = np.array(c.W_cnt_f_reg[i])
            disp = c.W_foot_displacement[0] if self.restrict_cnt else 0.
            W[f.pos_cost.name] = np.array([disp, disp, 0.]); W_e[f.pos_cost.name] = np.array([disp, disp, 0.])
        self.update_cost_weights()

    def update_cost_weights(self):
        """the weight vectors of the C-ABI in its residual order (include/nmpc.h)"""
        d = self.dyn
        W, W_e = self.data["W"], self.data["W_e"]
        # foot placement: the model carries x, y of each foot (8 rows); the reference never weights z (solver.py:133-134)
        for f in d.feet:
            assert W[f.pos_cost.name][2] == 0 and W_e[f.pos_cost.name][2] == 0, "pos_cost: only x, y are part of the model"
        self._W = np.concatenate([W[d.base_cost.name], W[d.joint_cost.name], W[d.acc_cost.name], W[d.swing_cost.name],
                                  np.concatenate([W[f.f_reg.name] for f in d.feet]),
                                  np.full(12, self.w_contact), np.full(6, self.w_consistency),
                                  np.concatenate([W[f.pos_cost.name][:2] for f in d.feet])])
        self._W_e = np.concatenate([W_e[d.base_cost.name], W_e[d.joint_cost.name], W_e[d.swing_cost.name],
                                    np.full(12, self.w_contact), np.full(6, self.w_consistency),
                                    np.concatenate([W_e[f.pos_cost.name][:2] for f in d.feet])])
        if self._dev is not None:
            self._dev.set_cost_weights(self._W, self._W_e, self.config_cost.reg_eps, self.config_cost.reg_eps_e)

    # ------------------------------------------------------------------------------------------ init helpers
    @staticmethod
    def _col(a):
        """[..., dim] -> [..., dim, 1]: a value repeated over the node axis of a view"""
        return np.asarray(a, float)[..., None]

    def setup_reference(self, base_ref, base_ref_e, joint_ref, step_height):               # solver.py:153-177
        d = self.dyn
        if base_ref_e is None:
            base_ref_e = np.array(base_ref).copy()
        self.cost_ref[d.base_cost.name][:] = self._col(base_ref)
        self.cost_ref[d.swing_cost.name][:] = step_height
        self.cost_ref_terminal[d.base_cost.name][:] = base_ref_e
        self.cost_ref_terminal[d.swing_cost.name][:] = step_height
        joint_ref = np.asarray(joint_ref, float)
        joint_ref_vel = np.concatenate((joint_ref, np.zeros_like(joint_ref)), axis=-1)
        self.cost_ref[d.joint_cost.name][:] = self._col(joint_ref_vel)
        self.cost_ref_terminal[d.joint_cost.name][:] = joint_ref_vel

    def setup_initial_state(self, q_euler, v_global):                                       # solver.py:179-192
        d = self.dyn
        self.data["x"][d.q.name] = np.array(q_euler, float)
        self.data["x"][d.v.name] = np.array(v_global, float)
        self.data["x"][d.h.name] = d.hg
        self.set_initial_state(self.data["x"])

    def set_initial_state(self, x: Dict[str, np.ndarray]):
        d = self.dyn
        self._x0 = np.concatenate([x[d.q.name], x[d.v.name], x[d.h.name]], axis=-1)
        for name in (d.q.name, d.v.name, d.h.name):
            self.states[name][..., 0] = x[name]

    def init_contacts_parameters(self):                                                     # solver.py:212-225
        for i, f in enumerate(self.dyn.feet):
            self.params[f.active.name][:] = 1.
            self.params[f.plane_normal.name][:] = self.default_normal[:, None]
            self.params[f.plane_point.name][:] = 0.
            self.params[f.plane_point.name][..., -1, :] = self.height_offset
            self.params[f.p_gain.name][:] = self.config_cost.W_foot_pos_constr_stab[i]
            self.params[f.restrict.name][:] = 0.
            self.params[f.range_radius.name][:] = self.config_cost.cnt_radius if self.restrict_cnt else 1.0e10

    def setup_cnt_status(self, cnt_sequence, peak_plan=None):                               # solver.py:227-252
        cnt_sequence = np.asarray(cnt_sequence)
        assert cnt_sequence.shape[-1] == self.config_opt.n_nodes + 1, \
            f"Invalid contact plan shape. Wrong number of optimization nodes. ({cnt_sequence.shape[-1]} vs {self.config_opt.n_nodes + 1})"
        assert cnt_sequence.shape[-2] == len(self.feet_frame_names), "Invalid contact plan shape. Wrong number of end effectors."
        for i, f in enumerate(self.dyn.feet):
            self.params[f.active.name][..., 0, :] = cnt_sequence[..., i, :]
            if self.config_opt.opt_peak and peak_plan is not None:
                self.params[f.peak.name][..., 0, :] = np.asarray(peak_plan)[..., i, :]
            if self.restrict_cnt:
                seq = cnt_sequence[..., i, :].astype(np.int64)
                restrict = np.diff(seq, prepend=seq[..., :1], axis=-1)
                restrict[restrict == -1] = 0
                self.params[f.restrict.name][..., 0, :] = restrict

    @time_fn("setup_contact_plan")
    def setup_contact_loc(self, contact_loc_plan):                                          # solver.py:254-276
        plan = np.asarray(contact_loc_plan, float)
        assert plan.shape[-2] == self.config_opt.n_nodes + 1, "Invalid contact plan shape. Wrong number of optimization nodes."
        assert plan.shape[-3] == len(self.feet_frame_names), "Invalid contact plan shape. Wrong number of end effectors."
        assert plan.shape[-1] == 3, "Invalid contact plan shape. 3D points required."
        for i, f in enumerate(self.dyn.feet):
            loc = np.swapaxes(plan[..., i, :, :], -1, -2)                                   # [.., 3, N+1]
            self.params[f.plane_point.name][:] = loc
            self.cost_ref[f.pos_cost.name][:] = loc[..., 1:]
            self.cost_ref_terminal[f.pos_cost.name][:] = loc[..., -1]

    def setup_initial_feet_pos(self, i_node: int = 0):                                      # solver.py:194-210
        feet_pos = self.dyn.get_feet_position_w()                                           # [.., 4, 3]
        for i, f in enumerate(self.dyn.feet):
            act = self.params[f.active.name]
            if i_node == 0:
                act[..., 0, 0] = 1
            pp = self.params[f.plane_point.name]
            if self.batch == 1:
                if act[0, 0]:
                    next_swing = int(np.argmin(act[0, :]))
                    pp[:, :next_swing] = feet_pos[i][:, None]
            else:
                for b in range(self.batch):
                    if act[b, 0, 0]:
                        next_swing = int(np.argmin(act[b, 0, :]))
                        pp[b, :, :next_swing] = feet_pos[b, i][:, None]

    @time_fn("warm_start_solver")
    def warm_start_solver(self, i_node: int, repeat_last: bool = False):                    # solver.py:290-342
        d = self.dyn
        start_node = i_node - self.last_node
        N = self.config_opt.n_nodes
        n_ws = N - start_node
        sw = lambda a: np.swapaxes(a, -1, -2)
        self.states[d.q.name][..., 1:n_ws + 1] = sw(self.q_sol_euler[..., start_node + 1:, :])
        self.states[d.v.name][..., 1:n_ws + 1] = sw(self.v_sol_euler[..., start_node + 1:, :])
        self.states[d.h.name][..., 1:n_ws + 1] = sw(self.h_sol[..., start_node + 1:, :])
        self.inputs[d.a.name][..., :n_ws] = sw(self.a_sol[..., start_node:, :])
        for i, f in enumerate(d.feet):
            self.inputs[f.force_name][..., :n_ws] = sw(self.f_sol[..., start_node:, i, :])
            self.inputs[f.force_name][..., n_ws:] = 0.
            if repeat_last:
                self.inputs[f.force_name][..., n_ws:] = self.f_sol[..., -1, i, :][..., None]
        if repeat_last and n_ws < N:
            self.states[d.q.name][..., n_ws:] = self.q_sol_euler[..., -1, :][..., None]
            self.states[d.v.name][..., n_ws:] = self.v_sol_euler[..., -1, :][..., None]
            self.states[d.h.name][..., n_ws:] = self.h_sol[..., -1, :][..., None]
            self.inputs[d.a.name][..., n_ws:] = self.a_sol[..., -1, :][..., None]
        self.last_node = i_node

    # ------------------------------------------------------------------------------------------ pack / unpack
    def _b(self, a):
        """view -> batch-major array with an explicit batch axis"""
        return a if self.batch > 1 else a[None]

    def pack_problem(self) -> Dict[str, np.ndarray]:
        """the dict views as the batch-major arrays of the C-ABI: x0[B,42], yref[B,N,90], yref_e[B,66],
        params[B,N+1,20], X[B,N+1,42], U[B,N,30] (what `update_solver` hands to the solver, solver.py:345-353)"""
        d, N, B = self.dyn, self.config_opt.n_nodes, self.batch
        t = lambda a: np.swapaxes(self._b(a), -1, -2)                                       # [B, nodes, dim]
        X = np.concatenate([t(self.states[d.q.name]), t(self.states[d.v.name]), t(self.states[d.h.name])], axis=-1)
        U = np.concatenate([t(self.inputs[d.a.name])] + [t(self.inputs[f.force_name]) for f in d.feet], axis=-1)
        act = np.concatenate([t(self.params[f.active.name]) for f in d.feet], axis=-1)      # [B, N+1, 4]
        peak = np.concatenate([t(self.params[f.peak.name]) for f in d.feet], axis=-1)
        pp = np.concatenate([t(self.params[f.plane_point.name]) for f in d.feet], axis=-1)
        params = np.concatenate([act, peak, pp], axis=-1)
        f_ref = np.concatenate([t(self.cost_ref[f.f_reg.name]) for f in d.feet], axis=-1)   # [B, N, 12]
        if self.force_reference == "gravity_share":
            n_st = np.maximum(act[:, :N].sum(-1, keepdims=True), 1.0)
            f_ref = f_ref.copy()
            f_ref[:, :, 2::3] += act[:, :N] * (-self.mp[5] * self.mp[1]) / n_st
        pos_ref = np.concatenate([t(self.cost_ref[f.pos_cost.name])[..., :2] for f in d.feet], axis=-1)        # [B, N, 8]
        pos_ref_e = np.concatenate([self._b(self.cost_ref_terminal[f.pos_cost.name])[..., :2] for f in d.feet], axis=-1)
        yref = np.concatenate([t(self.cost_ref[d.base_cost.name]), t(self.cost_ref[d.joint_cost.name]),
                               t(self.cost_ref[d.acc_cost.name]), t(self.cost_ref[d.swing_cost.name]), f_ref,
                               np.zeros((B, N, 18)), pos_ref], axis=-1)
        yref_e = np.concatenate([self._b(self.cost_ref_terminal[d.base_cost.name]), self._b(self.cost_ref_terminal[d.joint_cost.name]),
                                 self._b(self.cost_ref_terminal[d.swing_cost.name]), np.zeros((B, 18)), pos_ref_e], axis=-1)
        return dict(x0=self._b(self._x0), yref=yref, yref_e=yref_e, params=params, X=X, U=U)

    @time_fn("update_solver")
    def update_solver(self) -> None:                                                        # solver.py:344-353
        self._problem = self.pack_problem()

    @time_fn("init_solver")
    def init(self, i_node: int, q, v, base_ref, base_ref_e, joint_ref, step_height: float, cnt_sequence,
             cnt_locations=None, swing_peak=None):                                          # solver.py:355-394
        self.setup_reference(base_ref, base_ref_e, joint_ref, step_height)
        self.setup_initial_state(q, v)
        self.init_contacts_parameters()
        self.setup_cnt_status(cnt_sequence, swing_peak)
        if self.restrict_cnt:
            assert cnt_locations is not None, "Contact plan not provided"
            self.setup_contact_loc(cnt_locations)
        self.setup_initial_feet_pos(i_node)
        if i_node > 0 and self.config_opt.warm_start_sol:
            self.warm_start_solver(i_node, repeat_last=False)
        self.update_solver()

    def _device_solver(self):
        if self._dev is None:
            from .solver import BatchedNmpcSolver                                           # raises without a HIP device
            self._dev = BatchedNmpcSolver(MODEL_WHOLEBODY, self.config_opt.n_nodes, self.batch, self.device)
            self._dev.set_model_params(self.mp)
            self._dev.set_cost_weights(self._W, self._W_e, self.config_cost.reg_eps, self.config_cost.reg_eps_e)
            self._push_opts()
        return self._dev

    def _buffers(self):
        """device tensors of the problem and pinned host mirrors, allocated once: a solve is six asynchronous uploads, the
        kernels, four asynchronous downloads and ONE stream synchronisation (the caller reads the numpy views right away)"""
        import torch
        if getattr(self, "_dbuf", None) is None:
            s = self._device_solver()
            p = self._problem
            self._hbuf = {k: torch.empty(p[k].shape, dtype=torch.float32).pin_memory() for k in ("x0", "yref", "yref_e", "params", "X", "U")}
            self._dbuf = {k: torch.empty(p[k].shape, dtype=torch.float32, device=s.device) for k in self._hbuf}
            self._dbuf["status"] = torch.empty(self.batch, dtype=torch.int32, device=s.device)
            self._dbuf["stats"] = torch.empty(self.batch, 4, dtype=torch.float32, device=s.device)
            self._hbuf["status"] = torch.empty(self.batch, dtype=torch.int32).pin_memory()
            self._hbuf["stats"] = torch.empty(self.batch, 4, dtype=torch.float32).pin_memory()
        return self._hbuf, self._dbuf

    @time_fn("solve")
    def solve(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]:   # solver.py:396-429
        import torch
        s = self._device_solver()
        p = self._problem
        h, d = self._buffers()
        for k in ("x0", "yref", "yref_e", "params", "X", "U"):
            h[k].numpy()[...] = p[k]
            d[k].copy_(h[k], non_blocking=True)
        s.solve(d["x0"], d["yref"], d["yref_e"], d["params"], d["X"], d["U"], d["status"], d["stats"])
        for k in ("X", "U", "status", "stats"):
            h[k].copy_(d[k], non_blocking=True)
        torch.cuda.current_stream(s.device).synchronize()
        self.status, self.stats = h["status"].numpy().copy(), h["stats"].numpy().copy()
        self.parse_sol(h["X"].numpy().astype(np.float64), h["U"].numpy().astype(np.float64))
        return self.q_sol_euler, self.v_sol_euler, self.a_sol, self.f_sol, self.dt_node_sol

    def parse_sol(self, X: np.ndarray, U: np.ndarray):
        """solution -> dict views -> the five solution arrays (solver.py:403-429)"""
        d = self.dyn
        un = (lambda a: a) if self.batch > 1 else (lambda a: a[0])
        sw = lambda a: np.swapaxes(a, -1, -2)
        self.states[d.q.name][:] = un(sw(X[:, :, 0:18])); self.states[d.v.name][:] = un(sw(X[:, :, 18:36]))
        self.states[d.h.name][:] = un(sw(X[:, :, 36:42])); self.inputs[d.a.name][:] = un(sw(U[:, :, 0:18]))
        for i, f in enumerate(d.feet):
            self.inputs[f.force_name][:] = un(sw(U[:, :, 18 + 3 * i:21 + 3 * i]))
        self.q_sol_euler[:] = sw(self.states[d.q.name]); self.v_sol_euler[:] = sw(self.states[d.v.name])
        self.h_sol[:] = sw(self.states[d.h.name]); self.a_sol[:] = sw(self.inputs[d.a.name])
        self.f_sol[:] = np.stack([sw(self.inputs[f.force_name]) for f in d.feet], axis=-2)
        self.dt_node_sol[:] = self.dt_nodes

    def print_timings(self):
        from .profiling import print_timings
        print_timings(self.timings)
