"""Host-side kinematics of the declared whole-body quadruped model (NMPC_MODEL_WHOLEBODY, include/nmpc.h).

The reference gets foot positions and the centroidal momentum of the current state from pinocchio
(`QuadrupedDynamics.update_pin / get_feet_position_w`, mpc_controller/utils/dynamics.py:48-51,100-106;
`pin_data.hg`, utils/solver.py:187).  Neither pinocchio nor a URDF is in the image, so the model is the declared
one of DESIGN.md 3.2: trunk + four 3-joint legs with the geometry of `workloads.quadruped_tree()`, single-rigid-body
inertia for the momentum map.  numpy only: these feed the solver's inputs (anchored plane points, x0's momentum
slots); the per-stage kinematics of the solve run on the device.
"""
from __future__ import annotations

import numpy as np

from .references import rpy_to_matrix, euler_derivative_to_local_angular

N_JOINTS = 12
Q_HOME = np.tile([0.0, 0.79, -1.58], 4)       # declared nominal pose: feet under the hips at 0.30 m
LEG_SIGN_X = np.array([1.0, 1.0, -1.0, -1.0])  # FL, FR, RL, RR (main.py:83)
LEG_SIGN_Y = np.array([1.0, -1.0, 1.0, -1.0])
GEOMETRY = dict(hipx=0.19, hipy=0.047, lhip=0.095, l1=0.213, l2=0.213)


def feet_in_base(q_joints: np.ndarray, geom=GEOMETRY) -> np.ndarray:
    """[..., 12] joint angles -> [..., 4, 3] foot positions in the base frame."""
    ql = np.asarray(q_joints, float).reshape(*np.shape(q_joints)[:-1], 4, 3)
    q1, q2, q3 = ql[..., 0], ql[..., 1], ql[..., 2]
    l1, l2 = geom["l1"], geom["l2"]
    vx = -l1 * np.sin(q2) - l2 * np.sin(q2 + q3)
    vz = -l1 * np.cos(q2) - l2 * np.cos(q2 + q3)
    d = LEG_SIGN_Y * geom["lhip"]
    return np.stack([LEG_SIGN_X * geom["hipx"] + vx,
                     LEG_SIGN_Y * geom["hipy"] + d * np.cos(q1) - vz * np.sin(q1),
                     d * np.sin(q1) + vz * np.cos(q1)], axis=-1)


def feet_position_w(q: np.ndarray, geom=GEOMETRY) -> np.ndarray:
    """`get_feet_position_w` (dynamics.py:100-106) for q = [r, yaw, pitch, roll, joints(12)]: [4, 3]."""
    q = np.asarray(q, float)
    R = rpy_to_matrix(q[3:6][::-1])
    return q[:3] + feet_in_base(q[6:], geom) @ R.T


def _rotations(ypr: np.ndarray) -> np.ndarray:
    """[..., 3] (yaw, pitch, roll) -> [..., 3, 3] R = Rz Ry Rx, vectorised `rpy_to_matrix`"""
    y, p, r = ypr[..., 0], ypr[..., 1], ypr[..., 2]
    cy, sy, cp, sp, cr, sr = np.cos(y), np.sin(y), np.cos(p), np.sin(p), np.cos(r), np.sin(r)
    return np.stack([np.stack([cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr], -1),
                     np.stack([sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr], -1),
                     np.stack([-sp, cp * sr, cp * cr], -1)], -2)


def feet_position_w_batch(q: np.ndarray, geom=GEOMETRY) -> np.ndarray:
    """[B, 18] -> [B, 4, 3]"""
    q = np.asarray(q, float)
    return q[:, None, :3] + np.einsum("bij,bfj->bfi", _rotations(q[:, 3:6]), feet_in_base(q[:, 6:], geom))


def centroidal_momentum_batch(q: np.ndarray, v: np.ndarray, mass: float, inertia) -> np.ndarray:
    """[B, 18] x 2 -> [B, 6]"""
    q, v = np.asarray(q, float), np.asarray(v, float)
    sy, cy, sx, cx = np.sin(q[:, 4]), np.cos(q[:, 4]), np.sin(q[:, 5]), np.cos(q[:, 5])
    yd, pd, rd = v[:, 3], v[:, 4], v[:, 5]
    w_body = np.stack([-sy * yd + rd, cy * sx * yd + cx * pd, cx * cy * yd - sx * pd], -1)      # E(theta) thetadot
    return np.concatenate([mass * v[:, :3], np.einsum("bij,bj->bi", _rotations(q[:, 3:6]), np.asarray(inertia, float) * w_body)], -1)


def centroidal_momentum(q: np.ndarray, v: np.ndarray, mass: float, inertia) -> np.ndarray:
    """`pin_data.hg` of the declared model: [m rdot, R I_b E(theta) thetadot] (v[3:6] are Euler rates)."""
    q, v = np.asarray(q, float), np.asarray(v, float)
    R = rpy_to_matrix(q[3:6][::-1])
    w_body = euler_derivative_to_local_angular(q[3:6], v[3:6])
    return np.concatenate([mass * v[:3], R @ (np.asarray(inertia, float) * w_body)])
