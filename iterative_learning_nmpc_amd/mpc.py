"""Batched MPC controller: the host-side mirror of `LocomotionMPC` for B rollouts at once.

Restates the control flow of mpc_controller/mpc.py on top of the batched HIP solver:
    set_command                      mpc.py:197-202
    compute_base_ref_vel_tracking    mpc.py:210-272   (vectorised over the batch, same quantisation)
    optimize                         mpc.py:317-369   (contact flags -> references -> init -> solve)
    set_convergence_on_first_iter    mpc.py:464-473   (15 SQP iterations on the first solve)
    open_loop                        mpc.py:416-462   (simulator-free receding horizon: plant = plan)
    _replan / _step bookkeeping      mpc.py:171-186
    RaiberContactPlanner.get_locations + stance anchoring   contact_planner.py:265-322, solver.py:194-210 (footsteps=True)
    interpolate_state_trajectory     mpc.py:371-414   (record_sim_steps=True: one recorded row per simulation step)
The plant is the declared centroidal model; the whole-body controller is `mpc_wholebody.py`;
MuJoCo is outside this path.  All rollouts of a batch share the
gait clock (they are replanned at the same nodes), which is how the reference generates its perturbed
rollouts from one nominal rollout (data_collection_*_perturbed.py:176-247).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Optional, Tuple

import numpy as np
import torch

from .config import get_quadruped_config
from .contact_planner import ContactPlanner, RaiberContactPlanner
from .profiling import print_timings, time_fn
from .references import _euler_rate_matrix, _hermite, rpy_to_matrix
from .solver import BatchedNmpcSolver
from .workloads import FEET, HIP_OFFSETS, MODEL_CENTROIDAL, model_params

N_SQP_FIRST = 15   # mpc.py:465

# bits of `failed` (include/nmpc.h NMPC_ROLLOUT_FLAG_*); above them 1 + the replan that terminated the rollout
FLAG_SOLVER, FLAG_ROLL, FLAG_PITCH, FLAG_HEIGHT, FLAG_VEL_TRACKING, FLAG_COLLISION = 1, 2, 4, 8, 16, 32
FLAG_MASK, TERM_SHIFT = 0xFF, 8
# what ends a rollout early and makes the data collection discard and redo it: the controller diverged or the robot lies
# on the ground (the reference: mpc.diverged / a collision the simulator does not allow, RolloutMPC.py:404,424-437)
TERMINATE_DEFAULT = FLAG_SOLVER | FLAG_COLLISION
COLLISION_HEIGHT = 0.08      # [decl] base height of a trunk that touches the floor


def sample_pushes(n: int, seed, start: float = 0.0, duration: float = 0.3, magnitude=(50.0, 70.0)) -> dict:
    """n base pushes as the reference's data collection draws them (bc_experimental.yaml:32-35,
    data_collection_force_perturbation.py:213-248): direction uniform in the cube, normalised, magnitude uniform in
    `magnitude` newtons.  seed: anything numpy's SeedSequence takes (e.g. (rank, iteration, attempt))."""
    rng = np.random.default_rng(seed)
    force = rng.uniform(-1.0, 1.0, (n, 3))
    force /= np.linalg.norm(force, axis=1, keepdims=True) + 1e-6
    force *= rng.uniform(magnitude[0], magnitude[1], (n, 1))
    return dict(start=float(start), duration=float(duration), force=force)


def state_row_flags(rows: np.ndarray, v_des: np.ndarray, collision_height: float = COLLISION_HEIGHT) -> np.ndarray:
    """NMPC_ROLLOUT_FLAG_* bits of recorded 19-slot state rows [B, R, 19] (check_unsafe_state_v2,
    Rollout_combined_controller.py:367-431, + the collision height), OR-ed over the R rows: what the device's advance
    kernel raises, in fp32 like it."""
    r = np.asarray(rows, np.float32)
    lim = np.float32(25.0) * np.float32(0.017453292519943295)
    vd = np.asarray(v_des, np.float64).astype(np.float32)
    f = np.zeros(r.shape[:2], np.int32)
    f |= np.where(np.abs(r[:, :, 10]) > lim, FLAG_ROLL, 0)
    f |= np.where(np.abs(r[:, :, 9]) > lim, FLAG_PITCH, 0)
    f |= np.where((r[:, :, 7] < np.float32(0.18)) | (r[:, :, 7] > np.float32(0.45)), FLAG_HEIGHT, 0)
    f |= np.where((np.abs(r[:, :, 1] - vd[:, None, 0]) > np.float32(0.10)) | (np.abs(r[:, :, 2] - vd[:, None, 1]) > np.float32(0.10)),
                  FLAG_VEL_TRACKING, 0)
    f |= np.where(r[:, :, 7] < np.float32(collision_height), FLAG_COLLISION, 0)
    f |= np.where(~(np.abs(r[:, :, 7]) <= np.float32(1e30)), FLAG_SOLVER, 0)
    return np.bitwise_or.reduce(f, axis=1)


def base_ref_vel_tracking_batch(q, v_des, w_des, ref_state, t_horizon, nom_height, height_offset=0.0):
    """Vectorised `compute_base_ref_vel_tracking` (mpc.py:210-272) for q[B,>=4], v_des[B,3],
    w_des[B,3], ref_state[B,12].  Same roundings: np.round(.,2) on position, builtin round(.,1) on yaw,
    np.round(.,1) on the commanded velocity; same crossed-bounds clips."""
    q, v_des, w_des = (np.asarray(a, dtype=np.float64) for a in (q, v_des, w_des))
    B = q.shape[0]
    ref = np.zeros((B, 12))
    ref[:, :2] = np.round(q[:, :2], 2)
    ref[:, 2] = nom_height + height_offset
    ref[:, 3] = [round(float(y), 1) for y in q[:, 3]]
    R = np.stack([rpy_to_matrix(s[3:6][::-1]) for s in ref_state])
    v_glob = np.round(np.einsum("bij,bj->bi", R, v_des), 1)
    ref[:, 6:9] = v_glob
    ref[:, 9:12] = w_des[:, ::-1]
    ref_e = ref.copy()
    R_yaw = np.stack([rpy_to_matrix(w * t_horizon) for w in w_des])
    ref_e[:, 6:9] = np.einsum("bij,bj->bi", R_yaw, ref[:, 6:9])
    reach = v_glob[:, :2] * t_horizon
    ref_e[:, :2] = np.clip(ref_state[:, :2] + reach, -ref[:, :2] + 1.2 * reach, ref[:, :2] + 1.2 * reach)
    yaw_reach = w_des[:, 2] * t_horizon
    yaw_ref = ref_state[:, 3]
    ref_e[:, 3] = np.clip(yaw_ref + yaw_reach, -yaw_ref + 1.5 * yaw_reach, yaw_ref + 1.5 * yaw_reach)
    ref[:, :2] += 0.75 * (ref_e[:, :2] - ref[:, :2])
    ref[:, 3] += 0.75 * (ref_e[:, 3] - ref[:, 3])
    ref_e[:, 8] = 0.0
    ref_e[:, 4:6] = 0.0
    ref[:, 4:6] = 0.0
    ref_e[:, 10:12] = 0.0
    return ref, ref_e


class BatchedLocomotionMPC:
    """B centroidal rollouts driven by one batched NMPC solve per replanning step."""

    def __init__(self, batch: int, gait_name: str = "trot", robot_name: str = "go2", n_nodes: int = 50,
                 device="cuda:0", sim_dt: float = 1.0e-3, height_offset: float = 0.0,
                 compute_timings: bool = True, mass: float = 15.0, inertia=(0.11, 0.27, 0.33),
                 footsteps: bool = False, record_sim_steps: bool = False, terminate_mask: int = TERMINATE_DEFAULT,
                 collision_height: float = COLLISION_HEIGHT):
        """footsteps: stance feet anchored, touch-downs at Raibert targets (False: the feet stay under the initial hips
        for the whole rollout -- the stance-frozen rollouts of round 1).  record_sim_steps: one state row per
        simulation step (the plan up-sampled as mpc.py:371-414) instead of one per replan."""
        self.batch = int(batch)
        self.footsteps, self.record_sim_steps = bool(footsteps), bool(record_sim_steps)
        # early termination (include/nmpc.h, nmpc_rollout_cfg.terminate_mask): a rollout that raises one of these flags is
        # frozen; `invalid_mask` is what open_loop_device_valid discards and redoes (the same bits by default)
        self.terminate_mask, self.collision_height = int(terminate_mask) & FLAG_MASK, float(collision_height)
        self.invalid_mask = self.terminate_mask
        self.config_gait, self.config_opt, self.config_cost = get_quadruped_config(gait_name, robot_name)
        self.n_nodes = int(n_nodes)
        self.height_offset = height_offset
        self.sim_dt = sim_dt
        self.dt_nodes = self.config_opt.time_horizon / self.n_nodes
        self.replanning_freq = self.config_opt.replanning_freq
        self.replanning_steps = int(1 / (self.replanning_freq * sim_dt))                # mpc.py:113
        self.nodes_per_replan = max(1, int(round(self.replanning_steps * sim_dt / self.dt_nodes)))
        self.contact_planner = ContactPlanner(FEET, self.dt_nodes, self.config_gait)
        # footstep targets: the reference's Raibert planner with its offsets (mpc.py:80-91), stateless (cache_cnt=False)
        offs = HIP_OFFSETS.copy(); offs[:, 2] = 0.0
        self.raibert = RaiberContactPlanner(FEET, self.dt_nodes, self.config_gait, offs, y_offset=0.02, x_offset=0.04,
                                            foot_size=0.0085, height_offset=height_offset, cache_cnt=False)
        self.compute_timings = compute_timings
        self.timings = defaultdict(list)

        self.mp = model_params(dt=self.dt_nodes, mass=mass, Ixx=inertia[0], Iyy=inertia[1], Izz=inertia[2])
        self.solver = BatchedNmpcSolver(MODEL_CENTROIDAL, self.n_nodes, self.batch, device, compute_timings)
        self.solver.set_contact_patterns(gait_sequence=self.contact_planner.gait_sequence)   # kernel by gait
        self.device = self.solver.device
        self.solver.set_model_params(self.mp)
        W = np.concatenate([self.config_cost.W_base, self.config_cost.W_cnt_f_reg.ravel()])
        self.solver.set_cost_weights(W, self.config_cost.W_e_base, self.config_cost.reg_eps,
                                     self.config_cost.reg_eps_e)
        self.solver.set_max_qp_iter(self.config_opt.max_qp_iter)
        self.reset()

    # -- state ------------------------------------------------------------------------------------
    def reset(self) -> None:
        B, N = self.batch, self.n_nodes
        self.first_solve = True
        self.sim_step = 0
        self.current_opt_node = 0
        self.v_des = np.zeros((B, 3))
        self.w_des = np.zeros((B, 3))
        self.base_ref_vel_tracking = np.zeros((B, 12))
        self.foot_pos = None
        self.X = torch.zeros(B, N + 1, 12, dtype=torch.float32, device=self.device)
        self.U = torch.zeros(B, N, 12, dtype=torch.float32, device=self.device)
        self.status = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.stats = torch.zeros(B, 4, dtype=torch.float32, device=self.device)
        self.timings = defaultdict(list)
        self.solver.last_node = 0

    def set_command(self, v_des=np.zeros(3), w_yaw=0.0) -> None:
        self.v_des = np.broadcast_to(np.asarray(v_des, dtype=np.float64), (self.batch, 3)).copy()
        self.w_des[:, 2] = w_yaw

    def set_convergence_on_first_iter(self) -> None:
        if self.first_solve:
            self.solver.set_max_iter(N_SQP_FIRST)
            self.solver.set_nlp_tol(self.config_opt.nlp_tol / 10.0)
        elif self.sim_step <= self.replanning_steps:
            self.solver.set_max_iter(self.config_opt.max_iter)
            self.solver.set_nlp_tol(self.config_opt.nlp_tol)

    def _replan(self) -> bool:
        return self.sim_step % self.replanning_steps == 0

    def compute_base_ref_vel_tracking(self, q: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        return base_ref_vel_tracking_batch(q, self.v_des, self.w_des, self.base_ref_vel_tracking,
                                           self.config_opt.time_horizon, self.config_gait.nom_height,
                                           self.height_offset)

    def increment_base_ref_position(self, n_steps: int = 1, rows=None) -> None:
        """mpc.py:204-208, once per simulation step, vectorised over the batch (rows: mask of the rollouts that advance)."""
        if rows is not None and not rows.all():
            keep = self.base_ref_vel_tracking[~rows].copy()
            self.increment_base_ref_position(n_steps)
            self.base_ref_vel_tracking[~rows] = keep
            return
        s = self.base_ref_vel_tracking
        for _ in range(n_steps):
            cr, sr = np.cos(s[:, 5]), np.sin(s[:, 5])
            cp, sp = np.cos(s[:, 4]), np.sin(s[:, 4])
            cy, sy = np.cos(s[:, 3]), np.sin(s[:, 3])
            v = self.v_des
            vx = cy * cp * v[:, 0] + (cy * sp * sr - sy * cr) * v[:, 1] + (cy * sp * cr + sy * sr) * v[:, 2]
            vy = sy * cp * v[:, 0] + (sy * sp * sr + cy * cr) * v[:, 1] + (sy * sp * cr - cy * sr) * v[:, 2]
            s[:, 0] += np.round(vx, 1) * self.sim_dt
            s[:, 1] += np.round(vy, 1) * self.sim_dt
            s[:, 3] += self.w_des[:, 2] * self.sim_dt

    # -- one replanning step --------------------------------------------------------------------
    def build_problem(self, x: np.ndarray):
        """Per-stage references and parameters for the current node (host side, as mpc.py:332-366)."""
        B, N = self.batch, self.n_nodes
        contacts = self.contact_planner.get_contacts(self.current_opt_node, N + 1).astype(np.float64)  # [4,N+1]
        base_ref, base_ref_e = self.compute_base_ref_vel_tracking(x)
        if self.foot_pos is None:   # feet under the hips at the first call (setup_initial_feet_pos, solver.py:194)
            self.foot_pos = x[:, None, :3] * [1, 1, 0] + HIP_OFFSETS[None]
        params = np.zeros((B, N + 1, 16))
        params[:, :, :4] = contacts.T[None]
        if self.footsteps:
            params[:, :, 4:] = self.plan_foot_locations(x, contacts).reshape(B, N + 1, 12)
        else:
            params[:, :, 4:] = self.foot_pos.reshape(B, 1, 12)
        n_stance = np.maximum(contacts[:, :N].sum(0), 1.0)
        yref = np.zeros((B, N, 24))
        yref[:, :, :12] = base_ref[:, None, :]
        yref[:, :, 14::3] = (contacts[:, :N].T * (-self.mp[5] * self.mp[1] / n_stance)[:, None])[None]
        return yref, base_ref_e, params

    def plan_foot_locations(self, x: np.ndarray, contacts: np.ndarray) -> np.ndarray:
        """[B, N+1, 4, 3]: Raibert targets from every touch-down of the window on (contact_planner.py:265-322), then a foot
        in contact at node 0 keeps its current position up to its next swing node (solver.py:194-210, argmin and all)."""
        B, N = self.batch, self.n_nodes
        out = np.zeros((B, N + 1, 4, 3))
        for b in range(B):
            self.raibert.set_state(x[b, :3], x[b, 6:9], x[b, [5, 4, 3]], x[b, :3], self.v_des[b], self.w_des[b, 2])
            locs = self.raibert.get_locations(self.current_opt_node, N + 1)            # [4, N+1, 3]
            for f in range(4):
                if contacts[f, 0]:
                    locs[f, :int(np.argmin(contacts[f]))] = self.foot_pos[b, f]
            out[b] = np.moveaxis(locs, 0, 1)
        return out

    def touch_down(self, params: np.ndarray, rows=None) -> None:
        """feet that stand at the node the plant has reached are where the plan put them (rows: mask of the rollouts
        that advance -- a terminated one is frozen)"""
        npr = self.nodes_per_replan
        for f in range(4):
            on = params[:, npr, f] > 0.5
            if rows is not None:
                on &= rows
            self.foot_pos[on, f] = params[on, npr, 4 + 3 * f:7 + 3 * f]

    def sim_step_rows(self, X: np.ndarray, params: np.ndarray, replan_index: int) -> np.ndarray:
        """[B, replanning_steps, 19]: the plan up-sampled to the simulation rate as `interpolate_state_trajectory` does
        (mpc.py:371-414): positions on cubic Hermite segments through (q_k, qdot_k), velocities through (v_k, a_{k-1})
        with the reference's one-node shift of the accelerations (mpc.py:409-410); sample j at t = (j + 1) sim_dt."""
        B, N, dtn = self.batch, self.n_nodes, self.dt_nodes
        t_nodes = np.arange(N + 1) * dtn
        t_q = (np.arange(self.replanning_steps) + 1) * self.sim_dt
        seg = np.minimum(np.floor(t_q / dtn + 1e-9).astype(int), N - 1)
        period = self.config_gait.nominal_period
        t_w = (replan_index * self.replanning_steps + np.arange(self.replanning_steps) + 1) * self.sim_dt
        phase = np.round(np.fmod(t_w, period) / period, 4)
        rows = np.zeros((B, self.replanning_steps, 19))
        for b in range(B):
            Xb = X[b]
            thd = np.stack([_euler_rate_matrix(Xb[k, 3:6]) @ Xb[k, [11, 10, 9]] for k in range(N + 1)])
            pos = _hermite(t_nodes, Xb[:, :6], np.concatenate([Xb[:, 6:9], thd], axis=1), t_q)
            acc = (Xb[1:, 6:12] - Xb[:-1, 6:12]) / dtn
            vel = _hermite(t_nodes, Xb[:, 6:12], np.concatenate([acc[:1], acc]), t_q)
            stands = params[b, seg, :4] > 0.5                                             # [steps, 4]
            feet = np.where(stands[:, :, None] & self.footsteps, params[b, seg, 4:].reshape(-1, 4, 3), self.foot_pos[b][None])
            bwf = (pos[:, None, :2] - feet[:, :, :2]).reshape(-1, 8)
            rows[b] = np.concatenate([phase[:, None], vel, pos[:, 2:6], bwf], axis=1)
        return rows

    @time_fn("optimize")
    def optimize(self, x: np.ndarray) -> Tuple[torch.Tensor, torch.Tensor]:
        """One batched replanning solve from the states x[B,12]; returns device X[B,N+1,12], U[B,N,12]."""
        s = self.solver
        yref, yref_e, params = self.build_problem(x)
        self._params = params
        if self.first_solve:
            self.X[:] = s.to_device(np.repeat(x[:, None, :], self.n_nodes + 1, axis=1))
            self.U[:] = s.to_device(yref[:, :, 12:])
        # warm start (solver.py:304-322): the shift is folded into the solve
        shift = 0 if (self.first_solve or not self.config_opt.warm_start_sol) else self.current_opt_node - s.last_node
        s.last_node = self.current_opt_node
        s.solve(s.to_device(x), s.to_device(yref), s.to_device(yref_e), s.to_device(params),
                self.X, self.U, self.status, self.stats, shift=shift)
        return self.X, self.U

    # -- simulator-free rollout -------------------------------------------------------------------
    def open_loop(self, x0: np.ndarray, trajectory_time: float, push: Optional[dict] = None):
        """Receding-horizon rollout where the plant follows the plan (mpc.py:416-462).

        Returns (S[B,K,19] float32 device, t[K]): one recorded state per replanning step, in the
        slot order of the reference's recorded state (DAgger/utils/RolloutMPC.py:221):
            [phase, v(6) = rdot, body rates, q[2:] = z, yaw, pitch, roll, base_wrt_feet(8)].
        push = {"start": s, "duration": s, "force": [B,3] N}: a base push applied to the plant as the
        velocity impulse F dt / m per replanning interval (the reference pushes the MuJoCo base,
        data_collection_force_perturbation.py:213-248).
        `self.failed` [B] int32 (device): NMPC_ROLLOUT_FLAG_* bits of the recorded states, and above them 1 + the replan
        that terminated the rollout (`terminate_mask`: frozen from then on, its solves skipped, rows repeated) -- the
        same bookkeeping as `open_loop_device`.
        """
        x = np.array(x0, dtype=np.float64)
        n_replans = int(np.floor(trajectory_time / (self.replanning_steps * self.sim_dt) + 1e-9))
        dt_replan = self.replanning_steps * self.sim_dt
        rec, times = [], []
        flags = np.zeros(self.batch, np.int32)
        skip = torch.zeros(self.batch, dtype=torch.int32, device=self.device)
        self.solver.set_skip(skip, self.terminate_mask)
        try:
            for i in range(n_replans):
                t_now = i * dt_replan
                dead = (flags & self.terminate_mask) != 0
                skip.copy_(torch.as_tensor(flags))
                self.set_convergence_on_first_iter()
                X, _ = self.optimize(x)
                self.first_solve = False
                st = self.status.cpu().numpy()
                flags[~dead & ((st == 1) | (st == 4))] |= FLAG_SOLVER
                if self.record_sim_steps:
                    rows = self.sim_step_rows(X.double().cpu().numpy(), self._params, i)
                    times.extend(t_now + (np.arange(self.replanning_steps) + 1) * self.sim_dt)
                else:
                    rows = self.record_state(x, t_now)[:, None, :]
                    times.append(t_now)
                if dead.any():
                    rows[dead] = rec[-1][dead, -1:, :]
                rec.append(rows)
                flags[~dead] |= state_row_flags(rows[~dead], self.v_des[~dead], self.collision_height)
                newly = ~dead & ((flags & self.terminate_mask) != 0)
                flags[newly] |= (i + 1) << TERM_SHIFT
                go = ~dead & ~newly                                                 # these advance: plant = plan
                x_new = X[:, self.nodes_per_replan, :].double().cpu().numpy()
                # (half a simulation step of slack on both ends of the window, as nmpc_rollout_batch: float-safe)
                if push is not None and push["start"] - 0.5 * self.sim_dt <= t_now < push["start"] + push["duration"] - 0.5 * self.sim_dt:
                    x_new[:, 6:9] += np.asarray(push["force"]) * dt_replan / self.mp[1]
                x[go] = x_new[go]
                if self.footsteps:
                    self.touch_down(self._params, go)
                self.sim_step += self.replanning_steps
                self.current_opt_node += self.nodes_per_replan
                self.increment_base_ref_position(self.replanning_steps, go)
        finally:
            self.solver.set_skip(None, 0)
        self._x_last = x
        self.failed = torch.as_tensor(flags).to(self.device)
        S = torch.as_tensor(np.concatenate(rec, axis=1), dtype=torch.float32).to(self.device).contiguous()
        return S, np.asarray(times)

    def _rollout_cfg(self, n_replans: int, start_node: int, first_solve: bool, push: Optional[dict]):
        import ctypes
        from . import _lib
        return _lib.NmpcRolloutCfg(
            n_replans, self.nodes_per_replan, self.replanning_steps, self.contact_planner.nodes_per_cycle,
            start_node, int(first_solve), N_SQP_FIRST, self.config_opt.nlp_tol / 10.0,
            self.config_opt.nlp_tol, self.sim_dt, self.config_opt.time_horizon, self.config_gait.nom_height,
            self.height_offset, float(push["start"]) if push else 0.0, float(push["duration"]) if push else 0.0,
            int(self.footsteps), int(self.record_sim_steps),
            (ctypes.c_float * 8)(*self.raibert.offset_hip_b[:, :2].ravel().tolist()),
            (ctypes.c_float * 4)(*np.asarray(self.config_gait.stance_ratio, float).tolist()),
            float(self.config_gait.nominal_period), float(self.raibert.foot_size),
            int(self.terminate_mask), float(self.collision_height))

    def _device_rollout(self, n_replans, start_node, first_solve, push, x, v_des, w_des, ref_state, foot, X, U, status):
        """One nmpc_rollout_batch call on explicit device tensors (x, ref_state, foot, X, U are updated in place).
        Returns (S, failed)."""
        import ctypes
        from . import _lib
        s, dev, B = self.solver, self.device, x.shape[0]
        period = self.config_gait.nominal_period
        times = np.arange(n_replans) * self.replanning_steps * self.sim_dt
        phase = np.ascontiguousarray(np.round((times % period) / period, 4), dtype=np.float32)
        cfg = self._rollout_cfg(n_replans, start_node, first_solve, push)
        rows_per_replan = self.replanning_steps if self.record_sim_steps else 1
        if getattr(self, "_gait_dev", None) is None:
            self._gait_dev = torch.as_tensor(np.ascontiguousarray(self.contact_planner.gait_sequence), dtype=torch.int8).to(dev)
        force = s.to_device(np.asarray(push["force"])) if push else None
        S = torch.empty(B, n_replans * rows_per_replan, 19, dtype=torch.float32, device=dev)
        failed = torch.zeros(B, dtype=torch.int32, device=dev)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(s.lib.nmpc_rollout_batch(
            s._h, B, ctypes.byref(cfg), p(self._gait_dev), p(x), p(v_des), p(w_des), p(ref_state), p(foot), p(force),
            phase.ctypes.data_as(ctypes.c_void_p), p(X), p(U), p(S), p(status), p(failed),
            ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), s._h, "nmpc_rollout_batch")
        return S, failed

    def open_loop_device(self, x0: np.ndarray, trajectory_time: float, push: Optional[dict] = None):
        """`open_loop` with the whole rollout on the device: one C call launches every replanning
        step (references, shift, solve, plant update) on the stream -- no host round trip per replan.
        Same return value as `open_loop`; the controller state (X, U, node, reference) advances alike."""
        s, B, dev = self.solver, self.batch, self.device
        dt_replan = self.replanning_steps * self.sim_dt
        n_replans = int(np.floor(trajectory_time / dt_replan + 1e-9))
        if self.foot_pos is None:
            self.foot_pos = np.asarray(x0)[:, None, :3] * [1, 1, 0] + HIP_OFFSETS[None]
        x = s.to_device(x0)
        v_des = torch.as_tensor(self.v_des, dtype=torch.float64).to(dev).contiguous()
        w_des = torch.as_tensor(self.w_des, dtype=torch.float64).to(dev).contiguous()
        ref_state = torch.as_tensor(self.base_ref_vel_tracking, dtype=torch.float64).to(dev).contiguous()
        foot = s.to_device(self.foot_pos.reshape(B, 12))
        S, failed = self._device_rollout(n_replans, self.current_opt_node, self.first_solve, push, x, v_des, w_des, ref_state,
                                         foot, self.X, self.U, self.status)
        # advance the host-side bookkeeping as open_loop does
        self.first_solve = False
        self.sim_step += n_replans * self.replanning_steps
        self.current_opt_node += n_replans * self.nodes_per_replan
        s.last_node = self.current_opt_node - self.nodes_per_replan
        self.base_ref_vel_tracking = ref_state.cpu().numpy()
        self.x_final, self.failed = x, failed          # failed: NMPC_ROLLOUT_FLAG_* bits (include/nmpc.h)
        if self.footsteps:
            self.foot_pos = foot.cpu().numpy().astype(np.float64).reshape(B, 4, 3)
        times = np.arange(n_replans) * dt_replan
        if self.record_sim_steps:
            times = (times[:, None] + (np.arange(self.replanning_steps) + 1) * self.sim_dt).ravel()
        return S, times

    def open_loop_device_valid(self, x0: np.ndarray, trajectory_time: float, push_sampler, nominal=(0,), max_attempts: int = 8,
                               fill_batch: int = 0):
        """Pushed rollouts with the reference's discard-and-redo: a rollout that terminates early is thrown away and rolled
        again from the same initial state with a NEW push, until it runs to the end
        (data_collection_pretrain_omini_vc_policy_1direction_perturbed.py:217-247, `while True: ... if not
        early_termination: break`; RolloutMPC.py:424-437).  Here the whole batch rolls on the device, the rollouts whose
        flags meet `invalid_mask` are gathered into a compacted batch, given new pushes and rolled again -- at most
        `max_attempts` times in all (the reference loops without a cap); what is still invalid then keeps its flags and
        gets no sampling weight in the learning update (parallel.learning_update).

        push_sampler(n, attempt) -> {"start", "duration", "force": [n, 3]}; the rollouts listed in `nominal` are never
        pushed (and never redone).  `fill_batch` > 0: a redo pass over n rollouts rolls max(1, min(fill_batch, batch) // n) candidates
        of each, every one with its own new push, and keeps the FIRST that runs to the end -- the same distribution as the
        reference's one-after-the-other redo (independent pushes, first success kept), but the late passes over a
        handful of rollouts, which take fifty replans each however small they are, collapse into one.
        Call on a freshly reset controller.  Returns (S, t, info) with S, t as
        `open_loop_device` and info = {"attempt_sizes": rollouts run per attempt, "first_attempt": flag counts of the first
        pass}; `self.failed`, `self.x_final`, `self.X`, `self.U`, `self.foot_pos` hold the kept rollouts."""
        assert self.first_solve and self.current_opt_node == 0, "open_loop_device_valid starts from a reset controller"
        s, B, dev = self.solver, self.batch, self.device
        nominal = np.asarray(nominal, dtype=np.int64)
        x0 = np.asarray(x0, dtype=np.float64)
        ref0 = self.base_ref_vel_tracking.copy()
        foot0 = (self.foot_pos if self.foot_pos is not None else x0[:, None, :3] * [1, 1, 0] + HIP_OFFSETS[None]).copy()
        push = push_sampler(B, 0)
        force = np.array(push["force"], dtype=np.float64)
        force[nominal] = 0.0
        S, times = self.open_loop_device(x0, trajectory_time, dict(push, force=force))
        n_replans = int(np.floor(trajectory_time / (self.replanning_steps * self.sim_dt) + 1e-9))
        f0 = self.failed
        count = lambda bit: int((f0 & bit).ne(0).sum().item())
        info = {"attempt_sizes": [B],
                "first_attempt": {"invalid": count(self.invalid_mask), "solver": count(FLAG_SOLVER), "collision": count(FLAG_COLLISION),
                                  "roll": count(FLAG_ROLL), "pitch": count(FLAG_PITCH), "height": count(FLAG_HEIGHT),
                                  "velocity_tracking": count(FLAG_VEL_TRACKING)}}
        keep = torch.ones(B, dtype=torch.bool, device=dev)
        keep[torch.as_tensor(nominal, device=dev)] = False
        for attempt in range(1, max_attempts):
            idx = torch.nonzero((self.failed & self.invalid_mask).ne(0) & keep).flatten()
            n = int(idx.numel())                                   # (the one host round trip of an attempt)
            if n == 0:
                break
            cand = max(1, min(fill_batch, B) // n)                       # candidates per rollout of this pass (the solver holds B problems)
            m = n * cand
            idx_h = np.repeat(idx.cpu().numpy(), cand)                    # candidate j rolls rollout idx[j // cand]
            push_n = push_sampler(m, attempt)
            x = s.to_device(x0[idx_h])
            v_des = torch.as_tensor(self.v_des[idx_h], dtype=torch.float64).to(dev).contiguous()
            w_des = torch.as_tensor(self.w_des[idx_h], dtype=torch.float64).to(dev).contiguous()
            ref_state = torch.as_tensor(ref0[idx_h], dtype=torch.float64).to(dev).contiguous()
            foot = s.to_device(foot0[idx_h].reshape(m, 12))
            Xn = torch.zeros(m, self.n_nodes + 1, 12, dtype=torch.float32, device=dev)
            Un = torch.zeros(m, self.n_nodes, 12, dtype=torch.float32, device=dev)
            stn = torch.zeros(m, dtype=torch.int32, device=dev)
            Sn, failed_n = self._device_rollout(n_replans, 0, True, push_n, x, v_des, w_des, ref_state, foot, Xn, Un, stn)
            # the candidate kept for a rollout: its first valid one, or its first one if none is
            bad = (failed_n & self.invalid_mask).ne(0).view(n, cand)
            first_ok = torch.argmax((~bad).to(torch.int8), dim=1)         # (argmax of an all-zero row is 0)
            pick = torch.arange(n, device=dev) * cand + first_ok
            S[idx] = Sn[pick]
            self.failed[idx] = failed_n[pick]
            self.x_final[idx] = x[pick]
            self.X[idx] = Xn[pick]; self.U[idx] = Un[pick]; self.status[idx] = stn[pick]
            pick_h, rows_h = pick.cpu().numpy(), idx_h[::cand]
            self.base_ref_vel_tracking[rows_h] = ref_state.cpu().numpy()[pick_h]
            if self.footsteps:
                self.foot_pos[rows_h] = foot.cpu().numpy().astype(np.float64).reshape(m, 4, 3)[pick_h]
            info["attempt_sizes"].append(m)
        return S, times, info

    def record_state(self, x: np.ndarray, t: float) -> np.ndarray:
        period = self.config_gait.nominal_period
        phase = np.round((t % period) / period, 4)
        base_wrt_feet = (x[:, None, :2] - self.foot_pos[:, :, :2]).reshape(self.batch, 8)
        return np.concatenate([np.full((self.batch, 1), phase), x[:, 6:9], x[:, 9:12], x[:, 2:6],
                               base_wrt_feet], axis=1)

    def print_timings(self) -> None:
        print_timings(self.timings)
