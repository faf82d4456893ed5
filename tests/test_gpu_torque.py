"""GPU parity of the torque layer (include/nmpc_torque.h) against the fp64 oracle (oracle/torque_oracle.py,
itself checked against Lagrange's equations in tests/test_torque_oracle.py).  fp32 kernel: 1e-5 of the largest
torque of a sample (sums of ~100 products of O(1..100) terms)."""
import numpy as np
import pytest

from iterative_learning_nmpc_amd.workloads import quadruped_tree

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

CTRL_TO_JOINT = [3, 4, 5, 0, 1, 2, 9, 10, 11, 6, 7, 8]          # ctrl [FR, FL, RR, RL] -> joints [FL, FR, RL, RR]


def layer(m):
    """The device layer for the arrays an oracle model holds."""
    from iterative_learning_nmpc_amd.torque import BatchedTorqueLayer
    return BatchedTorqueLayer(m.parent, m.jtype, m.axis, m.R_fix, m.p_fix, m.mass, m.com, m.inertia, m.foot_joint, m.foot_offset,
                              m.nu, gravity=m.gravity)


def batch(m, B, seed):
    rng = np.random.default_rng(seed)
    q = rng.uniform(-1, 1, (B, m.n)); v = rng.uniform(-2, 2, (B, m.n)); a = rng.uniform(-5, 5, (B, m.n))
    f = rng.uniform(-40, 80, (B, len(m.foot_joint), 3))
    return [x.astype(np.float32) for x in (q, v, a, f)]


@pytest.mark.parametrize("perturb,B", [(0.0, 1), (0.0, 257), (0.3, 64), (0.3, 1000)])
def test_id_torques_match_oracle(perturb, B):
    from oracle import torque_oracle as to
    m = to.TreeModel.from_arrays(quadruped_tree(seed=4, perturb=perturb))
    q, v, a, f = batch(m, B, seed=B)
    tau = layer(m).id_torques(q, v, a, f).cpu().numpy()
    ref = to.id_torques_batch(m, *(x.astype(np.float64) for x in (q, v, a, f)))
    assert tau.shape == ref.shape == (B, 12)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert (np.abs(tau - ref) / scale).max() < 1e-5
    no_f = layer(m).id_torques(q, v, a).cpu().numpy()           # f_plan omitted = zero contact forces
    ref0 = to.id_torques_batch(m, *(x.astype(np.float64) for x in (q, v, a, 0 * f)))
    assert (np.abs(no_f - ref0) / np.abs(ref0).max(axis=1, keepdims=True)).max() < 1e-5


def test_general_tree_with_prismatic_joints_and_every_joint_actuated():
    """Not only the quadruped shape: a random tree, prismatic joints inside it, feet on inner bodies, all joints returned."""
    from oracle import torque_oracle as to
    rng = np.random.default_rng(12)
    n = 23
    parent = [-1] + [int(rng.integers(max(0, i - 4), i)) for i in range(1, n)]
    jtype = rng.integers(0, 2, n)
    axis = rng.standard_normal((n, 3)); axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    R = np.stack([to._axis_rotation(*(lambda r: (r / np.linalg.norm(r), rng.uniform(-2, 2)))(rng.standard_normal(3))) for _ in range(n)])
    inertia = np.stack([(lambda A: (A @ A.T + np.eye(3))[np.triu_indices(3)])(0.1 * rng.standard_normal((3, 3))) for _ in range(n)])
    m = to.TreeModel(parent, jtype, axis, R, 0.3 * rng.standard_normal((n, 3)), rng.uniform(0.1, 3.0, n), 0.1 * rng.standard_normal((n, 3)),
                     inertia, foot_joint=[5, 11, 22, 22, 0], foot_offset=0.2 * rng.standard_normal((5, 3)), n_actuated=n,
                     gravity=(0.3, -0.2, -9.7))
    q, v, a, f = batch(m, 96, seed=5)
    tau = layer(m).id_torques(q, v, a, f).cpu().numpy()
    ref = to.id_torques_batch(m, *(x.astype(np.float64) for x in (q, v, a, f)))
    assert (np.abs(tau - ref) / np.abs(ref).max(axis=1, keepdims=True)).max() < 1e-5


def test_pd_law_and_recorded_action_round_trip():
    from oracle import torque_oracle as to
    m = to.TreeModel.from_arrays(quadruped_tree())
    L = layer(m)
    rng = np.random.default_rng(3)
    q, v, qp, vp = (rng.standard_normal((40, m.n)).astype(np.float32) for _ in range(4))
    ff = rng.standard_normal((40, 12)).astype(np.float32)
    tau = L.compute_pd_torques(q, v, ff, qp, vp, 44.0, 5.0).cpu().numpy()
    ref = to.pd_torques(ff.astype(np.float64), q.astype(np.float64), v.astype(np.float64), qp.astype(np.float64), vp.astype(np.float64), 44.0, 5.0, 12)
    assert np.abs(tau - ref).max() < 1e-4 * np.abs(ref).max()
    assert np.allclose(L.compute_pd_torques(q, v, None, qp, vp, 20.0, 1.5).cpu().numpy(),
                       to.pd_torques(0.0, q, v, qp, vp, 20.0, 1.5, 12), rtol=1e-5, atol=1e-5)
    # RolloutMPC.py:228-250 with the reference's actuator order, kp = 20, kd = 1.5
    ctrl = rng.standard_normal((40, 12)).astype(np.float32)
    action = L.pd_target_action(ctrl, q, v, 20.0, 1.5, CTRL_TO_JOINT).cpu().numpy()
    tau_joint = np.concatenate([ctrl[:, 3:6], ctrl[:, 0:3], ctrl[:, 9:], ctrl[:, 6:9]], axis=1)
    assert np.allclose(action, (tau_joint + 1.5 * v[:, 6:]) / 20.0 + q[:, 6:], rtol=1e-6, atol=1e-6)
    # the action is the PD target that reproduces the torque: kp (action - q_j) - kd v_j = tau
    back = L.compute_pd_torques(q, v, None, np.concatenate([q[:, :6], action], axis=1), np.zeros_like(v), 20.0, 1.5).cpu().numpy()
    assert np.allclose(back, tau_joint, rtol=1e-4, atol=1e-4)


def test_model_errors():
    from iterative_learning_nmpc_amd._lib import NmpcError
    from iterative_learning_nmpc_amd.torque import BatchedTorqueLayer
    from oracle import torque_oracle as to
    m = to.TreeModel.from_arrays(quadruped_tree())
    bad_parent = list(m.parent); bad_parent[3] = 7
    with pytest.raises(NmpcError, match="parents"):
        BatchedTorqueLayer(bad_parent, m.jtype, m.axis, m.R_fix, m.p_fix, m.mass, m.com, m.inertia, m.foot_joint, m.foot_offset, m.nu)
    with pytest.raises(NmpcError, match="unit"):
        BatchedTorqueLayer(m.parent, m.jtype, 2 * m.axis, m.R_fix, m.p_fix, m.mass, m.com, m.inertia, m.foot_joint, m.foot_offset, m.nu)
    with pytest.raises(ValueError, match="expected"):
        layer(m).id_torques(np.zeros((2, 17), np.float32), np.zeros((2, 18), np.float32), np.zeros((2, 18), np.float32))
