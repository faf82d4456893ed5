"""The learning-update oracle (oracle/policy_oracle.py) against vectors produced by the reference's
own network module and training step (tests/golden/make_golden_policy.py)."""
import os

import numpy as np
import pytest

from oracle.policy_oracle import PolicyOracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["bn", "plain"])
def test_policy_oracle_matches_reference_network(name):
    g = np.load(os.path.join(GOLD, f"policy_{name}.npz"))
    n_in, n_out, L, hidden, bn, batch, steps = [int(v) for v in g["dims"]]
    o = PolicyOracle(n_in, n_out, L, hidden, bool(bn), np.float64)
    o.load_torch_state({k[5:]: g[k] for k in g.files if k.startswith("init.")})
    ref_final = PolicyOracle(n_in, n_out, L, hidden, bool(bn), np.float64)
    ref_final.load_torch_state({k[6:]: g[k] for k in g.files if k.startswith("final.")})
    for s in range(steps):
        loss, pred, grad = o.train_step(g["X"][s], g["Y"][s], float(g["lr"]))
        assert abs(loss - g["train_loss"][s]) < 2e-6
        assert np.abs(pred - g["train_pred"][s]).max() < 2e-5
        if s == 0:                                       # gradients of the first step, every tensor
            ref_grad = PolicyOracle(n_in, n_out, L, hidden, bool(bn), np.float64)
            ref_grad.load_torch_state({k[6:]: g[k] for k in g.files if k.startswith("grad0.") or "running" in k and k.startswith("init.")}
                                      | {k[5:]: g[k] for k in g.files if k.startswith("init.") and "running" in k})
            assert np.abs(grad - ref_grad.theta).max() < 2e-6
    # Parameters after three Adam steps.  Adam divides by sqrt(v) ~ |g|, so an entry whose exact gradient
    # is zero -- the bias in front of a BatchNorm, whose mean is subtracted again -- moves by +-lr per step
    # on rounding noise alone: those entries are only bounded, all others agree to a few 1e-6.
    noise = np.zeros(o.n_theta, bool)
    if bn:
        for n_, s_, off in o.items:
            if n_.startswith("b") and n_[1:].isdigit() and int(n_[1:]) < L:
                noise[off:off + int(np.prod(s_))] = True
    d = np.abs(o.theta - ref_final.theta)
    assert d[~noise].max() < 2e-5 and (not noise.any() or d[noise].max() <= 2.01 * steps * float(g["lr"]))
    if bn:
        # the running mean sees the noise-driven bias (momentum 0.1 x up to lr per step); the variance does not
        assert np.abs(o.running_mean - ref_final.running_mean).max() < 0.1 * 3 * steps * float(g["lr"])
        assert np.abs(o.running_var - ref_final.running_var).max() < 1e-6
    # eval-mode forward (running statistics) on the reference's own final parameters
    assert np.abs(ref_final.forward(g["X"][0], train=False) - g["eval_pred"]).max() < 2e-5


def test_philox_known_answer_and_sampler_distribution():
    """Philox-4x32-10 against the Random123 known-answer vector (counter 0, key 0), and the sampler's
    frequencies against the weights."""
    from oracle.policy_oracle import philox4x32_10, weighted_sample
    a, b = philox4x32_10(np.array([0]), 0)
    assert (int(a[0]), int(b[0])) == (0x6627E8D5, 0xE169C58D)
    w = np.ones(1000); w[::10] = 5.0                                # WeightedRandomSampler weights of test_train_policy.py:128-132
    idx = weighted_sample(w, 200000, seed=3)
    assert idx.min() >= 0 and idx.max() < 1000
    share = np.isin(idx, np.arange(0, 1000, 10)).mean()
    assert abs(share - 500.0 / 1400.0) < 0.005                       # 100 x 5 / (900 + 500)
    assert np.array_equal(idx, weighted_sample(w, 200000, seed=3)) and not np.array_equal(idx, weighted_sample(w, 200000, seed=4))
