"""Host-side reference generation and post-processing of the MPC solve.

Restates (numerically identical, golden vectors in tests/golden/):
  rpy_to_matrix                         pin.rpy.rpyToMatrix as used at mpc_controller/mpc.py:205,227,238
  local_angular_to_euler_derivative     mpc_controller/utils/transform.py:72-78
  euler_derivative_to_local_angular     mpc_controller/utils/transform.py:80-86
  base_ref_vel_tracking                 LocomotionMPC.compute_base_ref_vel_tracking, mpc.py:210-272
  increment_base_ref_position           mpc.py:204-208
  hermite_upsample                      interpolate_trajectory_with_derivatives, mpc.py:388-414
  zero_order_hold_index                 id_repeat, mpc.py:142
  base_ref_cnt_restricted               LocomotionMPC.compute_base_ref_cnt_restricted, mpc.py:274-315
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def rpy_to_matrix(rpy) -> np.ndarray:
    """R = Rz(yaw) Ry(pitch) Rx(roll) for rpy = (roll, pitch, yaw)."""
    r, p, y = (float(a) for a in rpy)
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([
        [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
        [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
        [-sp, cp * sr, cp * cr]])


def _euler_rate_matrix(ypr) -> np.ndarray:
    sx, cx = np.sin(ypr[2]), np.cos(ypr[2])
    sy, cy = np.sin(ypr[1]), np.cos(ypr[1])
    return np.array([[0.0, sx / cy, cx / cy], [0.0, cx, -sx], [1.0, sx * sy / cy, cx * sy / cy]])


def local_angular_to_euler_derivative(ypr_euler, w_local) -> np.ndarray:
    """(yaw,pitch,roll) rates from the body angular velocity (wx,wy,wz)."""
    return _euler_rate_matrix(ypr_euler) @ np.asarray(w_local)


def euler_derivative_to_local_angular(ypr_euler, v_euler) -> np.ndarray:
    """Body angular velocity (wx,wy,wz) from (yaw,pitch,roll) rates."""
    sx, cx = np.sin(ypr_euler[2]), np.cos(ypr_euler[2])
    sy, cy = np.sin(ypr_euler[1]), np.cos(ypr_euler[1])
    M = np.array([[-sy, 0.0, 1.0], [cy * sx, cx, 0.0], [cx * cy, -sx, 0.0]])
    return M @ np.asarray(v_euler)


def base_ref_vel_tracking(q, v_des, w_des, base_ref_state, t_horizon: float, nom_height: float,
                          height_offset: float = 0.0, interactive: bool = False
                          ) -> Tuple[np.ndarray, np.ndarray]:
    """Running and terminal 12-dim base references for velocity tracking.

    q              current configuration, q[:3] position, q[3] yaw
    v_des, w_des   commanded base-frame linear velocity and (.., .., yaw-rate)
    base_ref_state the controller's integrated reference `base_ref_vel_tracking` (12,)
    Keeps the reference's quantisation (np.round to 2 / 1 decimals, builtin round for yaw)
    and its crossed-bounds np.clip, which make the result what it is.
    """
    q, v_des, w_des = np.asarray(q, float), np.asarray(v_des, float), np.asarray(w_des, float)
    ref = np.zeros(12)
    ref[:2] = np.round(q[:2], 2)
    ref[2] = nom_height + height_offset
    ref[3] = round(float(q[3]), 1)
    v_glob = np.round(rpy_to_matrix(base_ref_state[3:6][::-1]) @ v_des, 1)
    ref[6:9] = v_glob
    ref[9:12] = w_des[::-1]

    ref_e = ref.copy()
    ref_e[6:9] = rpy_to_matrix(w_des * t_horizon) @ ref[6:9]
    if interactive:
        pos_ref, yaw_ref = np.round(q[:3], 2), q[3]
    else:
        pos_ref, yaw_ref = base_ref_state[:3], base_ref_state[3]
    reach = v_glob[:2] * t_horizon
    ref_e[:2] = np.clip(pos_ref[:2] + reach, -ref[:2] + 1.2 * reach, ref[:2] + 1.2 * reach)
    yaw_reach = w_des[-1] * t_horizon
    ref_e[3] = np.clip(yaw_ref + yaw_reach, -yaw_ref + 1.5 * yaw_reach, yaw_ref + 1.5 * yaw_reach)
    ref[:2] += 0.75 * (ref_e[:2] - ref[:2])
    ref[3] += 0.75 * (ref_e[3] - ref[3])
    ref_e[8] = 0.0
    ref_e[4:6] = 0.0
    ref[4:6] = 0.0
    ref_e[10:12] = 0.0
    return ref, ref_e


def increment_base_ref_position(base_ref_state, v_des, w_des, sim_dt: float) -> None:
    """Integrate the commanded velocity into the stored reference, in place (mpc.py:204-208)."""
    v_glob = np.round(rpy_to_matrix(base_ref_state[3:6][::-1]) @ np.asarray(v_des, float), 1)
    base_ref_state[:2] += v_glob[:2] * sim_dt
    base_ref_state[3] += w_des[-1] * sim_dt


def _hermite(t_knots, y, dy, t_query):
    """Piecewise cubic Hermite evaluation; y, dy are [K, d]; returns [len(t_query), d]."""
    t_knots = np.asarray(t_knots, float)
    seg = np.clip(np.searchsorted(t_knots, t_query, side="right") - 1, 0, len(t_knots) - 2)
    h = (t_knots[seg + 1] - t_knots[seg])[:, None]
    s = ((t_query - t_knots[seg]) / h[:, 0])[:, None]
    y0, y1, m0, m1 = y[seg], y[seg + 1], dy[seg], dy[seg + 1]
    h00 = (1 + 2 * s) * (1 - s) ** 2
    h10 = s * (1 - s) ** 2
    h01 = s * s * (3 - 2 * s)
    h11 = s * s * (s - 1)
    return h00 * y0 + h10 * h * m0 + h01 * y1 + h11 * h * m1


def hermite_upsample(time_traj, positions, velocities, accelerations, n_interp: int):
    """Cubic-Hermite upsampling of (q, v) and (v, a) to n_interp+1 uniformly spaced samples.

    positions, velocities: [K, d]; accelerations: [K-1, d] (first row is repeated in front,
    mpc.py:409-410).  Returns (pos[n_interp+1, d], vel[n_interp+1, d]).
    """
    time_traj = np.asarray(time_traj, float)
    t_query = np.linspace(time_traj[0], time_traj[-1], n_interp + 1)
    acc = np.concatenate((accelerations[:1], accelerations))
    return (_hermite(time_traj, np.asarray(positions), np.asarray(velocities), t_query),
            _hermite(time_traj, np.asarray(velocities), acc, t_query))


def zero_order_hold_index(n_interp: int, n_nodes: int) -> np.ndarray:
    """Node index held at each interpolated sample (mpc.py:142)."""
    return np.int32(np.linspace(0, 1, n_interp) * (n_nodes - 1))


def base_ref_cnt_restricted(contact_locations, nom_height: float, height_offset: float = 0.0,
                            blend: float = 0.35) -> Tuple[np.ndarray, np.ndarray]:
    """Running and terminal base references of the contact-restricted mode (Raibert plan in hand): the base is sent
    between the centre of the first and the centre of the last COMPLETE set of planned foot locations.

    contact_locations [4, N+1, 3]: the plan, all-zero where a foot has no planned location yet.  The distinct location
    sets are taken in numpy's sorted order (`np.unique(.., axis=1)`), a set counts a foot when all three coordinates of
    that foot are non-zero, and "complete" means "as many feet as the best set has" -- the reference's bincount/argmax
    construction, quirks included: its sets are sorted by value, not by time, so "first"/"last" are the extremes of that
    order.  Running reference: 0.35 first + 0.65 last in x, y; terminal: the last centre; height as configured; everything
    else zero.  Pinned by tests/golden/cnt_restricted.npz (the reference's own outputs)."""
    loc = np.asarray(contact_locations, float)
    sets = np.unique(loc, axis=1)                                   # [4, n_sets, 3]
    feet_planned = np.all(sets != 0.0, axis=-1).sum(axis=0)         # per set: feet with a location
    if feet_planned.any():
        best = np.flatnonzero(feet_planned == feet_planned.max())
        first, last = sets[:, best[0]].mean(axis=0), sets[:, best[-1]].mean(axis=0)
    else:                                                           # nothing planned at all: first and last node as they are
        first, last = loc[:, 0].mean(axis=0), loc[:, -1].mean(axis=0)
    ref, ref_e = np.zeros(12), np.zeros(12)
    ref[:2] = blend * first[:2] + (1.0 - blend) * last[:2]
    ref_e[:2] = last[:2]
    ref[2] = ref_e[2] = nom_height + height_offset
    return ref, ref_e
