"""The torque-layer oracle (oracle/torque_oracle.py): the recursive Newton-Euler restatement against
Lagrange's equations built from forward kinematics and geometric Jacobians -- a derivation that shares no
step with the recursion -- and against closed-form facts."""
import numpy as np
import pytest

from iterative_learning_nmpc_amd.workloads import quadruped_tree

from oracle import torque_oracle as to


def sample(m, rng, scale=1.0):
    q = scale * rng.uniform(-1, 1, m.n); v = scale * rng.uniform(-2, 2, m.n); a = scale * rng.uniform(-5, 5, m.n)
    f = rng.uniform(-40, 80, (len(m.foot_joint), 3))
    return q, v, a, f


@pytest.mark.parametrize("perturb", [0.0, 0.3])
def test_newton_euler_equals_lagrange(perturb):
    m = to.TreeModel.from_arrays(quadruped_tree(seed=4, perturb=perturb))
    rng = np.random.default_rng(1)
    for _ in range(4):
        q, v, a, f = sample(m, rng)
        tau, ref = to.id_torques(m, q, v, a, f), to.lagrangian_torques(m, q, v, a, f)
        assert np.abs(tau - ref).max() <= 2e-7 * max(1.0, np.abs(ref).max())


def test_static_stance_carries_the_weight():
    """Standing still on four feet that carry the weight equally: the virtual base joints need no force
    (sum of forces and moments balance up to the off-centre mass), and every term is gravity + J^T f."""
    m = to.TreeModel.from_arrays(quadruped_tree())
    q = np.zeros(m.n); q[2] = 0.4
    q[6:] = np.tile([0.0, 0.7, -1.4], 4)
    weight = m.mass.sum() * 9.81
    f = np.tile([0.0, 0.0, weight / 4], (4, 1))
    tau = to.id_torques(m, q, np.zeros(m.n), np.zeros(m.n), f)
    assert np.abs(tau[:3]).max() < 1e-9                      # net force on the base: zero
    no_f = to.id_torques(m, q, np.zeros(m.n), np.zeros(m.n), np.zeros((4, 3)))
    assert abs(no_f[2] - weight) < 1e-9                      # without feet the z joint carries the weight
    assert np.abs(no_f[:2]).max() < 1e-12


def test_torques_are_affine_in_acceleration_and_force():
    m = to.TreeModel.from_arrays(quadruped_tree(seed=2, perturb=0.2))
    rng = np.random.default_rng(3)
    q, v, a, f = sample(m, rng)
    t0 = to.id_torques(m, q, v, a, f)
    t1 = to.id_torques(m, q, v, a + 1.0, f); t2 = to.id_torques(m, q, v, a + 2.0, f)
    assert np.allclose(t2 - t1, t1 - t0, rtol=0, atol=1e-9)
    g1 = to.id_torques(m, q, v, a, 2 * f); g0 = to.id_torques(m, q, v, a, 0 * f)
    assert np.allclose(g1 - t0, t0 - g0, rtol=0, atol=1e-9)


def test_pd_law():
    rng = np.random.default_rng(0)
    q, v, qp, vp = (rng.standard_normal((5, 18)) for _ in range(4))
    ff = rng.standard_normal((5, 12))
    out = to.pd_torques(ff, q, v, qp, vp, 44.0, 5.0, 12)
    assert np.allclose(out, ff + 44.0 * (qp[:, 6:] - q[:, 6:]) + 5.0 * (vp[:, 6:] - v[:, 6:]))
