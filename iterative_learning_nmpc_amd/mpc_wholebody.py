"""`LocomotionMPC` over the whole-body solver facade: the reference's controller calls, unchanged.

Restates the part of mpc_controller/mpc.py that drives the solver, with the reference's
names, on top of `quadruped_solver.QuadrupedAcadosSolver` (the C-ABI facade):
    __init__ bookkeeping             mpc.py:25-119   (gait / opt / cost configs, planner, replanning_steps)
    reset                            mpc.py:121-169
    set_command                      mpc.py:197-202
    increment_base_ref_position      mpc.py:204-208
    compute_base_ref_vel_tracking    mpc.py:210-272  (references.base_ref_vel_tracking, golden-pinned)
    optimize                         mpc.py:317-369  (contacts, peaks, [Raibert locations], base refs, init, solve)
    set_convergence_on_first_iter    mpc.py:464-473
    interpolate_trajectory_with_derivatives / id_repeat   mpc.py:142,371-414  (references.hermite_upsample)
    open_loop                        mpc.py:416-462  (simulator-free receding horizon: plant = interpolated plan)
Batch = 1 keeps the reference's array shapes; `batch > 1` adds a leading axis to q, v and every returned array
(all rollouts share the gait clock).  MuJoCo, the keyboard goal, plotting and the asynchronous executor are outside
the path (C8, C10, C11).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Optional, Tuple

import numpy as np

from . import wholebody as wb
from .config import get_quadruped_config
from .contact_planner import ContactPlanner, RaiberContactPlanner
from .profiling import print_timings, time_fn
from .quadruped_solver import QuadrupedAcadosSolver
from .references import base_ref_cnt_restricted, base_ref_vel_tracking, hermite_upsample, increment_base_ref_position
from .workloads import FEET

N_SQP_FIRST = 15     # mpc.py:465


class LocomotionMPC:
    def __init__(self, path_urdf: str = "", feet_frame_names=FEET, robot_name: str = "go2", gait_name: str = "trot",
                 joint_ref: Optional[np.ndarray] = None, interactive_goal: bool = False, sim_dt: float = 1.0e-3,
                 height_offset: float = 0., contact_planner: str = "", print_info: bool = True,
                 compute_timings: bool = True, solve_async: bool = False, batch: int = 1, device="cuda:0",
                 n_nodes: Optional[int] = None, force_reference: str = "zero"):
        self.batch = int(batch)
        self.config_gait, self.config_opt, self.config_cost = get_quadruped_config(gait_name, robot_name)
        if n_nodes is not None:                      # BASELINE configs[2] asks for N = 30; the reference runs 25
            self.config_opt.n_nodes = int(n_nodes)
        self.print_info = print_info
        self.height_offset = height_offset
        self.solver = QuadrupedAcadosSolver(path_urdf, list(feet_frame_names), self.config_opt, self.config_cost,
                                            height_offset, print_info, compute_timings, batch=batch, device=device,
                                            force_reference=force_reference)
        self.nq, self.nv, self.nu, self.n_foot = 18, 18, wb.N_JOINTS, 4
        self.joint_ref = np.asarray(joint_ref, float) if joint_ref is not None else wb.Q_HOME.copy()   # mpc.py:73-81
        self._contact_planner_str = contact_planner
        if contact_planner.lower() == "raibert":                                            # mpc.py:75-92
            q0 = np.zeros(18)
            q0[6:] = self.joint_ref
            self.solver.dyn.update_pin(q0, np.zeros(18))
            offset_hip_b = np.array(self.solver.dyn.get_feet_position_w())
            offset_hip_b[:, -1] = 0.
            self.contact_planner = RaiberContactPlanner(list(feet_frame_names), self.solver.dt_nodes, self.config_gait,
                                                        offset_hip_b, y_offset=0.02, x_offset=0.04, foot_size=0.0085,
                                                        cache_cnt=False)
            self.restrict_cnt = True
        else:
            self.contact_planner = ContactPlanner(list(feet_frame_names), self.solver.dt_nodes, self.config_gait)
            self.restrict_cnt = False
        self.solver.set_contact_restriction(self.restrict_cnt)
        self.Kp, self.Kd = self.config_opt.Kp, self.config_opt.Kd
        self.sim_dt = sim_dt
        self.dt_nodes = self.solver.dt_nodes
        self.replanning_freq = self.config_opt.replanning_freq
        self.replanning_steps = int(1 / (self.replanning_freq * sim_dt))                    # mpc.py:113
        self.compute_timings = compute_timings
        self.reset(reset_solver=False)

    def reset(self, reset_solver: bool = True) -> None:                                     # mpc.py:121-169
        if reset_solver:
            self.solver.reset()
        self.first_solve, self.diverged = True, False
        self.sim_step = self.plan_step = self.current_opt_node = self.delay = 0
        sh = (self.batch, 3) if self.batch > 1 else (3,)
        self.v_des, self.w_des = np.zeros(sh), np.zeros(sh)
        self.base_ref_vel_tracking = np.zeros((self.batch, 12) if self.batch > 1 else 12)
        self.n_interp_plan = round(self.config_opt.time_horizon / self.sim_dt)
        self.id_repeat = np.int32(np.linspace(0, 1, self.n_interp_plan) * (self.config_opt.n_nodes - 1))   # mpc.py:142
        self.timings = defaultdict(list)

    def set_command(self, v_des=np.zeros((3,)), w_yaw: float = 0.) -> None:                 # mpc.py:197-202
        self.v_des = np.broadcast_to(np.asarray(v_des, float), self.v_des.shape).copy()
        self.w_des[..., 2] = w_yaw

    def increment_base_ref_position(self):                                                  # mpc.py:204-208
        if self.batch == 1:
            increment_base_ref_position(self.base_ref_vel_tracking, self.v_des, self.w_des, self.sim_dt)
        else:
            for b in range(self.batch):
                increment_base_ref_position(self.base_ref_vel_tracking[b], self.v_des[b], self.w_des[b], self.sim_dt)

    def compute_base_ref_vel_tracking(self, q: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:   # mpc.py:210-272
        T, h = self.config_opt.time_horizon, self.config_gait.nom_height
        if self.batch == 1:
            return base_ref_vel_tracking(q, self.v_des, self.w_des, self.base_ref_vel_tracking, T, h, self.height_offset)
        out = [base_ref_vel_tracking(q[b], self.v_des[b], self.w_des[b], self.base_ref_vel_tracking[b], T, h, self.height_offset)
               for b in range(self.batch)]
        return np.stack([o[0] for o in out]), np.stack([o[1] for o in out])

    def compute_base_ref_cnt_restricted(self, q, contact_locations):                        # mpc.py:274-315
        return base_ref_cnt_restricted(contact_locations, self.config_gait.nom_height, self.height_offset)

    def _replan(self) -> bool:                                                              # mpc.py:171-175
        return self.sim_step % self.replanning_steps == 0

    def set_convergence_on_first_iter(self):                                                # mpc.py:464-473
        """iteration policy of the solve about to run: the first one gets N_SQP_FIRST SQP iterations at a tenth of the
        configured tolerances; the configured policy is restored during the first replanning interval after it"""
        cfg = self.solver.config_opt
        if self.first_solve:
            policy = (N_SQP_FIRST, cfg.nlp_tol / 10., cfg.qp_tol / 10.)
        elif self.sim_step <= self.replanning_steps:
            policy = (cfg.max_iter, cfg.nlp_tol, cfg.qp_tol)
        else:
            return
        for setter, value in zip((self.solver.set_max_iter, self.solver.set_nlp_tol, self.solver.set_qp_tol), policy):
            setter(value)

    def solver_inputs(self, q: np.ndarray, v: np.ndarray):
        """what `optimize` hands to `solver.init` (mpc.py:325-366), without solving"""
        N = self.config_opt.n_nodes
        self.solver.dyn.update_pin(q, v)
        cnt_sequence = self.contact_planner.get_contacts(self.current_opt_node, N + 1)
        swing_peak = self.contact_planner.get_peaks(self.current_opt_node, N + 1) if self.config_opt.opt_peak else None
        cnt_locations = None
        if self.restrict_cnt:
            assert self.batch == 1, "contact-location plans are per rollout: use batch = 1"
            if self._contact_planner_str.lower() == "raibert":
                com_xyz = np.asarray(q[:3], float)          # declared model: centre of mass at the base origin
                self.contact_planner.set_state(q[:3], v[:3], q[3:6][::-1], com_xyz, self.v_des, self.w_des[-1])
            cnt_locations = self.contact_planner.get_locations(self.current_opt_node, N + 1)
            base_ref, base_ref_e = self.compute_base_ref_cnt_restricted(q, cnt_locations)
        else:
            base_ref, base_ref_e = self.compute_base_ref_vel_tracking(q)
        if self.batch > 1:
            cnt_sequence = np.broadcast_to(cnt_sequence, (self.batch,) + cnt_sequence.shape)
            swing_peak = None if swing_peak is None else np.broadcast_to(swing_peak, (self.batch,) + swing_peak.shape)
        joint_ref = self.joint_ref if self.batch == 1 else np.broadcast_to(self.joint_ref, (self.batch, 12))
        return (self.current_opt_node, q, v, base_ref, base_ref_e, joint_ref, self.config_gait.step_height,
                cnt_sequence, cnt_locations, swing_peak)

    @time_fn("optimize")
    def optimize(self, q: np.ndarray, v: np.ndarray):                                       # mpc.py:317-369
        self.solver.init(*self.solver_inputs(q, v))
        return self.solver.solve()

    def interpolate_trajectory_with_derivatives(self, time_traj, positions, velocities, accelerations):   # mpc.py:388-414
        return hermite_upsample(time_traj, positions, velocities, accelerations, self.n_interp_plan)

    def open_loop(self, q0: np.ndarray, v0: np.ndarray, trajectory_time: float):
        """Simulator-free rollout (mpc.py:416-462): replan every `replanning_steps`, interpolate the plan to the
        simulation rate, follow it as if it were the plant.  q0, v0 in the solver's Euler layout.
        Returns q_traj[K(,B),18], one row per simulation step."""
        q, v = np.array(q0, float), np.array(v0, float)
        sim_time, q_rows, v_rows, time_traj = 0.0, [], [], None
        while sim_time <= trajectory_time:
            if sim_time >= (self.current_opt_node + 1) * self.dt_nodes:
                self.current_opt_node += 1
            if self._replan():
                self.set_convergence_on_first_iter()
                q_sol, v_sol, a_sol, f_sol, dt_sol = self.optimize(q, v)
                self.first_solve = False
                time_traj = np.concatenate(([0.], np.cumsum(dt_sol[0] if self.batch > 1 else dt_sol)))
                # interpolate_state_trajectory (mpc.py:371-386): sample 0 is the current state and is dropped
                if self.batch == 1:
                    qp, vp = self.interpolate_trajectory_with_derivatives(time_traj, q_sol, v_sol, a_sol)
                    self.q_plan, self.v_plan = qp[1:], vp[1:]
                else:
                    pl = [self.interpolate_trajectory_with_derivatives(time_traj, q_sol[b], v_sol[b], a_sol[b]) for b in range(self.batch)]
                    self.q_plan, self.v_plan = np.stack([p[0][1:] for p in pl], 1), np.stack([p[1][1:] for p in pl], 1)
                self.a_plan, self.f_plan = np.take(a_sol, self.id_repeat, axis=-2), np.take(f_sol, self.id_repeat, axis=-3)
                self.plan_step = 0
            q, v = self.q_plan[self.plan_step].copy(), self.v_plan[self.plan_step].copy()
            q_rows.append(q)
            v_rows.append(v)
            self._step()
            sim_time = sim_time + self.sim_dt
        self.v_traj = np.stack(v_rows)                       # the velocities that go with the returned rows
        return np.stack(q_rows)

    def replan_clock(self, trajectory_time: float):
        """The float clock of `open_loop` without its solves: (number of simulation steps it runs, optimisation node of each
        of its replans) from the controller's current counters -- what `open_loop_device` hands to the device, so that both
        replan at the same nodes (the reference advances the node when the accumulated float time passes a node time,
        mpc.py:171-186, 427-431)."""
        sim_time, sim_step, node, nodes, steps = 0.0, self.sim_step, self.current_opt_node, [], 0
        while sim_time <= trajectory_time:
            if sim_time >= (node + 1) * self.dt_nodes:
                node += 1
            if sim_step % self.replanning_steps == 0:
                nodes.append(node)
            sim_step += 1
            steps += 1
            sim_time = sim_time + self.sim_dt
        return steps, nodes, node

    def state_rows(self, q_traj: np.ndarray, v_traj: np.ndarray, t0_step: int = 0) -> np.ndarray:
        """states of this controller (Euler layout, [K, 18] each; K, B, 18 for a batch) as the reference's recorded rows
        [phase, v_mj(18), q_mj[2:](17), base_wrt_feet(8)] (DAgger/utils/RolloutMPC.py:221): row j is the state after
        simulation step t0_step + j"""
        from .trajectory_io import convert_to_mujoco
        q_traj, v_traj = np.asarray(q_traj, float), np.asarray(v_traj, float)
        if q_traj.ndim == 3:
            return np.stack([self.state_rows(q_traj[:, b], v_traj[:, b], t0_step) for b in range(q_traj.shape[1])])
        period = self.config_gait.nominal_period
        rows = np.zeros((len(q_traj), 44))
        for j, (q, v) in enumerate(zip(q_traj, v_traj)):
            q_mj, v_mj = convert_to_mujoco(q, v)
            tw = (t0_step + j + 1) * self.sim_dt
            bwf = (q[None, :2] - wb.feet_position_w(q)[:, :2]).reshape(8)
            rows[j] = np.concatenate([[np.round(np.fmod(tw, period) / period, 4)], v_mj, q_mj[2:], bwf])
        return rows

    def open_loop_device(self, q0: np.ndarray, v0: np.ndarray, trajectory_time: float, push: Optional[dict] = None,
                         record_sim_steps: bool = True, terminate_mask: int = 33, collision_height: float = 0.08):
        """`open_loop` with the whole rollout on the device (nmpc_wb_rollout_batch): per replan the problem is assembled from
        the plant state by a kernel, solved with the warm-start shift folded in, and the up-sampled plan is followed for
        `replanning_steps` simulation steps -- one host call, no round trip per replan, the whole batch at once.
        Returns S [B, K, 44] (device): the recorded rows of `state_rows` -- one per simulation step, K as `open_loop`
        runs, or one per replan (the state it starts from) with record_sim_steps=False.  push = {"start", "duration",
        "force": [B, 3]}: velocity impulse F dt / m on the base per replanning interval.  The controller's counters,
        reference and solution views advance as in `open_loop`; `self.failed` holds the NMPC_ROLLOUT_FLAG_* bits,
        `self.q_final` / `self.v_final` the plant state."""
        import ctypes
        import torch
        from . import _lib
        fs = self.solver
        s = fs._device_solver()
        B, N, dev = self.batch, self.config_opt.n_nodes, s.device
        steps, nodes, node_end = self.replan_clock(trajectory_time)
        n_replans = len(nodes)
        cfg_o = self.config_opt
        s.set_max_iter(cfg_o.max_iter); s.set_nlp_tol(cfg_o.nlp_tol); s.set_max_qp_iter(cfg_o.max_qp_iter)
        fs._opts.update(max_iter=cfg_o.max_iter, nlp_tol=cfg_o.nlp_tol, qp_tol=cfg_o.qp_tol)
        cfg = _lib.NmpcWbRolloutCfg(
            n_replans, self.replanning_steps, self.contact_planner.nodes_per_cycle, int(self.first_solve), int(fs.last_node),
            N_SQP_FIRST, cfg_o.nlp_tol / 10.0, cfg_o.nlp_tol, self.sim_dt, cfg_o.time_horizon, self.config_gait.nom_height,
            self.height_offset, float(self.config_gait.step_height), float(push["start"]) if push else 0.0,
            float(push["duration"]) if push else 0.0, int(record_sim_steps), int(fs.force_reference == "gravity_share"),
            float(self.config_gait.nominal_period), int(terminate_mask), float(collision_height))
        t32 = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a, np.float64).reshape(B, -1)), dtype=torch.float32).to(dev).contiguous()
        t64 = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a, np.float64).reshape(B, -1)), dtype=torch.float64).to(dev).contiguous()
        q, v = t32(q0), t32(v0)
        v_des, w_des, ref_state = t64(self.v_des), t64(self.w_des), t64(self.base_ref_vel_tracking)
        gait = torch.as_tensor(np.ascontiguousarray(self.contact_planner.gait_sequence), dtype=torch.int8).to(dev)
        peaks = torch.as_tensor(np.ascontiguousarray(self.contact_planner.peak_swing), dtype=torch.int8).to(dev)
        joint_ref = torch.as_tensor(self.joint_ref, dtype=torch.float32).to(dev).contiguous()
        force = t32(push["force"]) if push else None
        if getattr(self, "_X_dev", None) is None or self.first_solve:
            self._X_dev = torch.zeros(B, N + 1, 42, dtype=torch.float32, device=dev)
            self._U_dev = torch.zeros(B, N, 30, dtype=torch.float32, device=dev)
        rows = n_replans * (self.replanning_steps if record_sim_steps else 1)
        S = torch.empty(B, rows, 44, dtype=torch.float32, device=dev)
        status = torch.zeros(B, dtype=torch.int32, device=dev)
        failed = torch.zeros(B, dtype=torch.int32, device=dev)
        nodes_c = (ctypes.c_int * n_replans)(*nodes)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(s.lib.nmpc_wb_rollout_batch(
            s._h, B, ctypes.byref(cfg), p(gait), p(peaks), ctypes.cast(nodes_c, ctypes.c_void_p), p(q), p(v), p(v_des), p(w_des),
            p(ref_state), p(joint_ref), p(force), p(self._X_dev), p(self._U_dev), p(S), p(status), p(failed),
            ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), s._h, "nmpc_wb_rollout_batch")
        # bookkeeping as open_loop leaves it
        self.first_solve = False
        self.sim_step += steps
        self.current_opt_node = node_end
        fs.last_node = nodes[-1]
        ref = ref_state.cpu().numpy()
        # (open_loop integrates the reference once per simulation step it runs; the device does it per full replanning interval)
        self.base_ref_vel_tracking = ref if self.batch > 1 else ref[0]
        self.failed, self.status_dev = failed, status
        self.q_final, self.v_final = q, v
        fs.parse_sol(self._X_dev.cpu().numpy().astype(np.float64), self._U_dev.cpu().numpy().astype(np.float64))
        return S[:, :steps] if record_sim_steps else S

    def _step(self) -> None:                                                                # mpc.py:183-186
        self.increment_base_ref_position()
        self.sim_step += 1
        self.plan_step += 1

    def print_timings(self):
        print_timings(self.timings)
        self.solver.print_timings()
