"""Periodic gait schedule -> per-node contact flags (input generator of the solve).

Restates the behaviour of mpc_controller/utils/contact_planner.py:
  GaitPlanner.__init__/_init_gait_cycle/_init_peak_cycle   :9-118
  get_contacts / get_peaks / get_make_break_contacts        :121-168
  RaiberContactPlanner.get_locations (Raibert heuristic)    :265-322
as table look-ups: the cycle tables are built once, windows are modular gathers, and
`*_batch` variants produce `[B, n_foot, n_nodes]` blocks for a whole batch of rollouts at once
(the reference builds one window per call with np.tile + slicing).
Bit-exact against tests/golden/contact_planner.npz.
"""
from __future__ import annotations

from math import ceil
from typing import List, Sequence, Tuple

import numpy as np

from .config import GaitConfig
from .references import rpy_to_matrix


class GaitPlanner:
    def __init__(self, feet_frame_names: Sequence[str], dt_nodes: float, config_gait: GaitConfig):
        self.feet_frame_names = list(feet_frame_names)
        self.n_foot = len(self.feet_frame_names)
        self.dt_nodes = dt_nodes
        self.config_gait = config_gait
        self.nodes_per_cycle = round(config_gait.nominal_period / dt_nodes)
        npc = self.nodes_per_cycle

        self.gait_sequence = np.zeros((self.n_foot, npc), dtype=np.int8)
        self.switch_cnt = np.zeros((self.n_foot, npc), dtype=np.int8)
        self.cnt_intervals = {f: [] for f in self.feet_frame_names}
        self.swing_intervals = {f: [] for f in self.feet_frame_names}

        touch_down = np.asarray(config_gait.phase_offset, dtype=np.float64)
        lift_off = ((touch_down + np.asarray(config_gait.stance_ratio)) % 1.0).round(2)
        self.switch_phase = np.unique(np.concatenate((lift_off, touch_down)))
        for foot, (name, td, lo) in enumerate(zip(self.feet_frame_names, touch_down, lift_off)):
            first, last = ceil(td * npc), ceil(lo * npc)
            if td < lo:  # stance is one interval inside the cycle
                self.gait_sequence[foot, first:last] = 1
                self.cnt_intervals[name].append((td, lo))
                for a, b in ((lo, 1.0), (0.0, td)):
                    if a != b:
                        self.swing_intervals[name].append((a, b))
            else:        # stance wraps around the end of the cycle
                self.gait_sequence[foot, first:] = 1
                self.gait_sequence[foot, :last] = 1
                for a, b in ((td, 1.0), (0.0, lo)):
                    if a != b:
                        self.cnt_intervals[name].append((a, b))
                self.swing_intervals[name].append((lo, td))
            self.switch_cnt[foot, first] = 1
            self.switch_cnt[foot, last] = -1
        # "peak" flag = swing flag (contact_planner.py:113-118)
        self.peak_swing = (1 - self.gait_sequence).astype(np.int8)

    # -- windows ------------------------------------------------------------------
    def _window(self, table: np.ndarray, i_node, n_nodes: int) -> np.ndarray:
        cols = (np.asarray(i_node)[..., None] % self.nodes_per_cycle + np.arange(n_nodes)) \
            % self.nodes_per_cycle
        out = table[:, cols]                      # [n_foot, (B,) n_nodes]
        return out if out.ndim == 2 else np.moveaxis(out, 0, 1)

    def get_contacts(self, i_node: int, n_nodes: int) -> np.ndarray:
        """int8 [n_foot, n_nodes]; 1 = stance, 0 = swing, starting at node i_node."""
        return self._window(self.gait_sequence, int(i_node), n_nodes)

    def get_peaks(self, i_node: int, n_nodes: int) -> np.ndarray:
        return self._window(self.peak_swing, int(i_node), n_nodes)

    def get_make_break_contacts(self, i_node: int, n_nodes: int) -> Tuple[np.ndarray, np.ndarray]:
        make = (self.switch_cnt == 1).astype(np.int8)
        brk = (self.switch_cnt == -1).astype(np.int8)
        return self._window(make, int(i_node), n_nodes), self._window(brk, int(i_node), n_nodes)

    def get_contacts_batch(self, i_nodes, n_nodes: int) -> np.ndarray:
        """int8 [B, n_foot, n_nodes] for a vector of start nodes."""
        return self._window(self.gait_sequence, np.asarray(i_nodes, dtype=np.int64), n_nodes)

    def get_peaks_batch(self, i_nodes, n_nodes: int) -> np.ndarray:
        return self._window(self.peak_swing, np.asarray(i_nodes, dtype=np.int64), n_nodes)

    # -- phase queries --------------------------------------------------------------
    def _is_in_cnt_phase(self, foot: str, phase: float) -> bool:
        return any(a <= phase < b for a, b in self.cnt_intervals.get(foot, []))

    def _is_in_cnt(self, foot: str, i_node: int) -> bool:
        phase = round((i_node % self.nodes_per_cycle) / self.nodes_per_cycle, 3)
        return self._is_in_cnt_phase(foot, phase)


class ContactPlanner(GaitPlanner):
    """Schedule only; contact locations are left to the solver (contact_planner.py:170-180)."""

    def get_locations(self, i_node: int, n_nodes: int):
        return None


class RaiberContactPlanner(ContactPlanner):
    """Raibert footstep heuristic (contact_planner.py:182-322)."""
    GRAVITY = 9.81
    V_TRACKING = 0.05

    def __init__(self, feet_frame_names, dt_nodes, config_gait, offset_hip_b: np.ndarray,
                 x_offset: float = 0.0, y_offset: float = 0.0, foot_size: float = 0.0,
                 height_offset: float = 0.0, cache_cnt: bool = True):
        super().__init__(feet_frame_names, dt_nodes, config_gait)
        self.foot_size = foot_size
        self.cache_cnt = cache_cnt
        self.height_offset = height_offset
        self.offset_hip_b = offset_hip_b  # adjusted in place, as the reference does (:223-226)
        if self.n_foot == 4:
            self.offset_hip_b[:, 0] += np.array([x_offset, x_offset, -x_offset, -x_offset])
            self.offset_hip_b[:, 1] += np.array([y_offset, -y_offset, y_offset, -y_offset])
        self.planed_cnt = {foot: {} for foot in range(self.n_foot)}
        self.pos = self.v_w = self.euler_rpy = self.com_xyz = self.v_des = self.w_yaw = None

    def set_state(self, pos, v_w, euler_rpy, com_xyz, v_des=np.zeros(3), w_yaw: float = 0.0):
        self.pos, self.v_w, self.euler_rpy = pos, v_w, euler_rpy
        self.com_xyz, self.v_des, self.w_yaw = np.array(com_xyz), np.array(v_des), w_yaw

    def remove_cnt_before(self, i_node: int):
        self.planed_cnt = {f: {n: c for n, c in d.items() if n >= i_node}
                           for f, d in self.planed_cnt.items()}

    def get_locations(self, i_node: int, n_nodes: int) -> np.ndarray:
        """[n_foot, n_nodes, 3]; zeros until a foot's first touch-down in the window."""
        locs = np.zeros((self.n_foot, n_nodes, 3))
        make, _ = self.get_make_break_contacts(i_node, n_nodes)
        com_xy = self.com_xyz[:2]
        com_z = self.com_xyz[-1] - self.height_offset
        v_cmd = self.v_des[:2]
        R_yaw = rpy_to_matrix(np.array([0.0, 0.0, self.euler_rpy[2]]))
        for foot, k in np.argwhere(make == 1):
            node = i_node + k
            if self.cache_cnt and node in self.planed_cnt[foot]:
                locs[foot, k:] = self.planed_cnt[foot][node]
                continue
            t_touch = round(k * self.dt_nodes, 3)
            if t_touch < 0:
                continue
            ratio = self.config_gait.stance_ratio[foot]
            t_stance = self.config_gait.nominal_period * ratio
            hip = com_xy + (R_yaw @ self.offset_hip_b[foot])[:2] + v_cmd * t_touch * (1 + ratio)
            # (max: the reference's sqrt(com_z / g), contact_planner.py:311, is NaN for a base below the ground -- a state its
            #  simulator cannot reach, the centroidal plant can)
            lever = 0.5 * np.sqrt(max(com_z, 0.0) / self.GRAVITY) * v_cmd
            centrifugal = np.array([lever[1] * self.w_yaw, -lever[0] * self.w_yaw])  # (lever,0) x (0,0,w)
            target = np.zeros(3)
            target[:2] = hip + 0.1 * (v_cmd - self.v_w[:2]) + 0.5 * v_cmd * t_stance + centrifugal[:2]
            target[2] = self.foot_size
            locs[foot, k:] = target
            if self.cache_cnt:
                self.planed_cnt[foot][node] = target
        return locs
