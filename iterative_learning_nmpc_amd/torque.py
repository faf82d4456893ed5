"""Torque layer -- host mirror of `QuadrupedDynamics.id_torques` (mpc_controller/utils/dynamics.py:136-163),
`LocomotionMPC._compute_pd_torques` (mpc_controller/mpc.py:592-599) and the recorded action
(DAgger/utils/RolloutMPC.py:228-250) over the C-ABI of include/nmpc_torque.h, with a leading batch axis.

The robot is given as arrays (the reference reads a URDF through pinocchio; neither is in the image): a tree
of 1-DoF joints whose first six are the virtual base joints of the reference's state
[px, py, pz, yaw, pitch, roll, joints]."""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class BatchedTorqueLayer:
    def __init__(self, parent: Sequence[int], joint_type: Sequence[int], axis, placement_R, placement_p, mass, com, inertia,
                 foot_joint: Sequence[int], foot_offset, n_actuated: int, gravity=(0.0, 0.0, -9.81), device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedTorqueLayer needs a HIP device: there is no CPU path")
        self.lib = _lib.load()
        self.device = torch.device(device)
        n, nf = len(parent), len(foot_joint)
        f32 = lambda x, shape: np.ascontiguousarray(np.asarray(x, np.float32).reshape(shape))   # noqa: E731
        i32 = lambda x: np.ascontiguousarray(np.asarray(x, np.int32))                            # noqa: E731
        placement = np.concatenate([f32(placement_R, (n, 9)), f32(placement_p, (n, 3))], axis=1)
        arrays = dict(parent=i32(parent), type=i32(joint_type), axis=f32(axis, (n, 3)), placement=np.ascontiguousarray(placement),
                      mass=f32(mass, (n,)), com=f32(com, (n, 3)), inertia=f32(inertia, (n, 6)),
                      foot_joint=i32(foot_joint), foot_offset=f32(foot_offset, (nf, 3)))
        m = _lib.NmpcTreeModel()
        m.n_joints, m.n_actuated, m.n_feet = n, int(n_actuated), nf
        for k, a in arrays.items():
            setattr(m, k, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int if a.dtype == np.int32 else ctypes.c_float)))
        m.gravity = (ctypes.c_float * 3)(*[float(g) for g in gravity])
        self._h = ctypes.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        rc = self.lib.nmpc_torque_create(ctypes.byref(m), idx, ctypes.byref(self._h))
        if rc:
            raise _lib.NmpcError(f"nmpc_torque_create: {self.lib.nmpc_torque_last_error(None).decode()}")
        self.n, self.nu, self.n_feet = n, int(n_actuated), nf

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self.lib.nmpc_torque_destroy(h)

    def _check(self, rc, what):
        if rc:
            raise _lib.NmpcError(f"{what}: {self.lib.nmpc_torque_last_error(self._h).decode()}")

    def _in(self, t, width, name):
        t = torch.as_tensor(t, dtype=torch.float32, device=self.device).contiguous()
        if t.dim() < 2 or tuple(t.shape[1:]) != tuple(width):
            raise ValueError(f"{name}: expected [B, {', '.join(map(str, width))}], got {tuple(t.shape)}")
        return t

    def id_torques(self, q_plan, v_plan, a_plan, f_plan=None) -> torch.Tensor:
        """dynamics.py:136-163 per robot: q, v, a [B, n]; f [B, n_feet, 3] world-frame contact forces -> [B, nu]."""
        q = self._in(q_plan, (self.n,), "q_plan"); v = self._in(v_plan, (self.n,), "v_plan"); a = self._in(a_plan, (self.n,), "a_plan")
        f = None if f_plan is None else self._in(f_plan, (self.n_feet, 3), "f_plan")
        B = q.shape[0]
        if v.shape[0] != B or a.shape[0] != B or (f is not None and f.shape[0] != B):
            raise ValueError("batch sizes differ")
        tau = torch.empty(B, self.nu, dtype=torch.float32, device=self.device)
        self._check(self.lib.nmpc_id_torques_batch(self._h, B, _ptr(q), _ptr(v), _ptr(a), _ptr(f), _ptr(tau), _stream(self.device)),
                    "nmpc_id_torques_batch")
        return tau

    def compute_pd_torques(self, q, v, torques_ff, q_plan, v_plan, Kp: float, Kd: float) -> torch.Tensor:
        """mpc.py:592-599: torques_ff + Kp (q_plan[-nu:] - q[-nu:]) + Kd (v_plan[-nu:] - v[-nu:])."""
        q = self._in(q, (self.n,), "q"); v = self._in(v, (self.n,), "v")
        qp = self._in(q_plan, (self.n,), "q_plan"); vp = self._in(v_plan, (self.n,), "v_plan")
        ff = None if torques_ff is None else self._in(torques_ff, (self.nu,), "torques_ff")
        tau = torch.empty(q.shape[0], self.nu, dtype=torch.float32, device=self.device)
        self._check(self.lib.nmpc_pd_torques_batch(self._h, q.shape[0], _ptr(ff), _ptr(q), _ptr(v), _ptr(qp), _ptr(vp), float(Kp),
                                                   float(Kd), _ptr(tau), _stream(self.device)), "nmpc_pd_torques_batch")
        return tau

    def pd_target_action(self, tau, q, v, kp: float = 20.0, kd: float = 1.5, actuator_to_joint: Optional[Sequence[int]] = None):
        """RolloutMPC.py:228-250: action = (tau[perm] + kd v_j) / kp + q_j (kp = 20, kd = 1.5 in the reference)."""
        tau = self._in(tau, (self.nu,), "tau"); q = self._in(q, (self.n,), "q"); v = self._in(v, (self.n,), "v")
        perm = None if actuator_to_joint is None else torch.as_tensor(list(actuator_to_joint), dtype=torch.int32, device=self.device)
        if perm is not None and (perm.numel() != self.nu or sorted(perm.tolist()) != list(range(self.nu))):
            raise ValueError("actuator_to_joint must be a permutation of range(nu)")
        out = torch.empty_like(tau)
        self._check(self.lib.nmpc_pd_target_action_batch(self._h, tau.shape[0], _ptr(tau), _ptr(perm), _ptr(q), _ptr(v), float(kp),
                                                         float(kd), _ptr(out), _stream(self.device)), "nmpc_pd_target_action_batch")
        return out
