"""Trajectory `.npz` files in the reference's rollout schema.

`StateDataRecorder` (DAgger/utils/RolloutMPC.py:102-258) stores one rollout as an `.npz` with the keys
    time q v ctrl feet_pos_w base_wrt_feet state action vc_goals cc_goals contact_vec is_expert
one row per simulation step: `q` (19: position, quaternion wxyz, joints) and `v` (18: world linear velocity, LOCAL angular
velocity, joint rates) in MuJoCo's layout, `ctrl` in the actuator order [FR, FL, RR, RL], `state` = [phase, v(18), q[2:](17),
base_wrt_feet(8)] (44 slots, :221), `action` = (tau_[FL,FR,RL,RR] + kd v_j) / kp + q_j (:250), file names
`traj_nominal_<stamp>.npz` / `traj_<replanning point>_<n>.npz` (:148-166).  `merge_trajs_from_dir`-style consumers
(Behavior_Cloning/utils/data_collection_force_perturbation.py:123-158) read `time` and `state`.

`TrajectoryRecorder` writes that schema from this package's rollouts: states of the whole-body controller in the solver's
Euler layout are converted as `QuadrupedDynamics.convert_to_mujoco` does (mpc_controller/utils/dynamics.py:75-98).
hdf5 (data_collection_locosafedagger.py:69-90) needs h5py, which is not in the image: not built.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np

from .references import euler_derivative_to_local_angular, rpy_to_matrix

KEYS = ("time", "q", "v", "ctrl", "feet_pos_w", "base_wrt_feet", "state", "action", "vc_goals", "cc_goals",
        "contact_vec", "is_expert")
FEET = ("FL", "FR", "RL", "RR")
KP, KD = 20.0, 1.5                     # RolloutMPC.py:33-34
N_STATE = 44


def matrix_to_quaternion_wxyz(R: np.ndarray) -> np.ndarray:
    """unit quaternion (w, x, y, z) of a rotation matrix, w >= 0 branch-free on the largest component"""
    t = np.trace(R)
    cand = np.array([t, R[0, 0], R[1, 1], R[2, 2]])
    i = int(np.argmax(cand))
    if i == 0:
        w = 0.5 * np.sqrt(1.0 + t)
        q = np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])
    else:
        a = i - 1; b = (a + 1) % 3; c = (a + 2) % 3
        s = 0.5 * np.sqrt(1.0 + R[a, a] - R[b, b] - R[c, c])
        q = np.zeros(4)
        q[1 + a] = s
        q[0] = (R[c, b] - R[b, c]) / (4 * s)
        q[1 + b] = (R[b, a] + R[a, b]) / (4 * s)
        q[1 + c] = (R[c, a] + R[a, c]) / (4 * s)
    return q if q[0] >= 0 else -q


def convert_to_mujoco(q_euler: np.ndarray, v_euler: np.ndarray):
    """solver layout [x y z yaw pitch roll joints], [v_lin, Euler rates, joint rates] -> MuJoCo's (dynamics.py:75-98)"""
    q_euler, v_euler = np.asarray(q_euler, float), np.asarray(v_euler, float)
    q_mj = np.zeros(len(q_euler) + 1)
    q_mj[:3] = q_euler[:3]
    q_mj[3:7] = matrix_to_quaternion_wxyz(rpy_to_matrix(q_euler[3:6][::-1]))
    q_mj[7:] = q_euler[6:]
    v_mj = v_euler.copy()
    v_mj[3:6] = euler_derivative_to_local_angular(q_euler[3:6], v_euler[3:6])
    return q_mj, v_mj


def state_row(phase: float, v_mj: np.ndarray, q_mj: np.ndarray, base_wrt_feet: np.ndarray) -> np.ndarray:
    """[phase, v(18), q[2:](17), base_wrt_feet(8)] (RolloutMPC.py:221)"""
    return np.concatenate([np.round([phase], 4), v_mj, q_mj[2:], base_wrt_feet])


def pd_target_action(ctrl_frflrrrl: np.ndarray, v_mj: np.ndarray, q_mj: np.ndarray, kp: float = KP, kd: float = KD) -> np.ndarray:
    """torques in the actuator order [FR, FL, RR, RL] -> PD target in the joint order [FL, FR, RL, RR] (RolloutMPC.py:228-250)"""
    t = np.asarray(ctrl_frflrrrl, float)
    tau = np.concatenate([t[3:6], t[0:3], t[9:12], t[6:9]])
    return (tau + kd * v_mj[6:]) / kp + q_mj[7:]


class TrajectoryRecorder:
    """One rollout in the reference's `.npz` schema."""

    def __init__(self, record_dir: str = "", v_des=np.zeros(3), current_time: float = 0.0, nominal_flag: bool = True,
                 replanning_point: int = 0, nth_traj_per_replanning: int = 0, kp: float = KP, kd: float = KD):
        self.record_dir = record_dir
        self.vc_goals = np.asarray(v_des, float)
        self.current_time = current_time
        self.nominal_flag, self.replanning_point, self.nth = nominal_flag, replanning_point, nth_traj_per_replanning
        self.kp, self.kd = kp, kd
        self.reset()

    def reset(self) -> None:
        self.data: Dict[str, list] = {k: [] for k in KEYS}

    def record(self, time: float, q_mj, v_mj, ctrl, feet_pos_w, contact_vec=(0, 0, 0, 0), is_expert: int = 0,
               phase: float = 0.0, cc_goals: Optional[np.ndarray] = None) -> None:
        """one simulation step (RolloutMPC.py:168-258).  feet_pos_w: [4, 3] in the order FL, FR, RL, RR.  The reference's
        phase is the constant 0 (`get_phase_percentage`, :262-273) and its contact-conditioned goals are noise (:256)."""
        q_mj, v_mj, ctrl = np.array(q_mj, float), np.array(v_mj, float), np.array(ctrl, float)
        feet = np.asarray(feet_pos_w, float).reshape(4, 3)
        base_wrt_feet = (q_mj[None, :3] - feet)[:, :2].reshape(8)
        d = self.data
        d["time"].append(round(time + self.current_time, 4))
        d["q"].append(q_mj); d["v"].append(v_mj); d["ctrl"].append(ctrl)
        d["feet_pos_w"].append(feet.reshape(12)); d["base_wrt_feet"].append(base_wrt_feet)
        d["contact_vec"].append(np.asarray(contact_vec, dtype=np.int64))
        d["state"].append(state_row(phase, v_mj, q_mj, base_wrt_feet))
        d["action"].append(pd_target_action(ctrl, v_mj, q_mj, self.kp, self.kd))
        d["vc_goals"].append(self.vc_goals)
        d["cc_goals"].append(np.zeros(8) if cc_goals is None else np.asarray(cc_goals, float))
        d["is_expert"].append(is_expert)

    def record_solver_state(self, time: float, q_euler, v_euler, ctrl, feet_pos_w, **kw) -> None:
        """a state of the whole-body controller (`LocomotionMPC.open_loop`, Euler layout) as one recorded step"""
        q_mj, v_mj = convert_to_mujoco(q_euler, v_euler)
        self.record(time, q_mj, v_mj, ctrl, feet_pos_w, **kw)

    def file_name(self, stamp: str = "") -> str:
        return f"traj_nominal_{stamp}.npz" if self.nominal_flag else f"traj_{self.replanning_point}_{self.nth}.npz"

    def save(self, stamp: str = "") -> str:
        os.makedirs(self.record_dir or os.getcwd(), exist_ok=True)
        path = os.path.join(self.record_dir or os.getcwd(), self.file_name(stamp))
        np.savez(path, **self.data)
        return path


def load_trajectory(path: str) -> Dict[str, np.ndarray]:
    """a rollout file as arrays, with the schema checked"""
    with np.load(path) as f:
        d = {k: f[k] for k in f.files}
    missing = [k for k in KEYS if k not in d]
    if missing:
        raise ValueError(f"{path}: not a rollout file of the reference's schema, missing {missing}")
    T = len(d["time"])
    for k, dim in (("q", 19), ("v", 18), ("ctrl", 12), ("feet_pos_w", 12), ("base_wrt_feet", 8), ("state", N_STATE),
                   ("action", 12), ("vc_goals", 3), ("cc_goals", 8), ("contact_vec", 4)):
        if d[k].shape != (T, dim):
            raise ValueError(f"{path}: {k} has shape {d[k].shape}, expected {(T, dim)}")
    return d
