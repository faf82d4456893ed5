"""GPU parity of the learning update (include/nmpc_policy.h) against the numpy oracle and the
vectors generated from the reference's own network module.  fp32 throughout; tolerances are the
fp32-vs-fp64 floor of sums over 512 terms (1e-5 relative on predictions and gradients-in-effect)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _pair(n_in, n_out, L, hidden, bn, batch_max, seed=0):
    """A DevicePolicy and an fp64 oracle with the same (random, non-trivial) parameters."""
    from iterative_learning_nmpc_amd.policy import DevicePolicy
    from oracle.policy_oracle import PolicyOracle
    pol = DevicePolicy(n_in, n_out, L, hidden, bn, batch_max=batch_max, device="cuda:0", seed=seed)
    o = PolicyOracle(n_in, n_out, L, hidden, bn, np.float64)
    rng = np.random.default_rng(seed + 1)
    theta, rm, rv = (t.cpu().numpy().astype(np.float64) for t in pol.get_parameters())
    for name, shape, off in pol.items:                   # biases, gamma, beta away from their trivial start values
        n = int(np.prod(shape))
        if name.endswith(".b") or name.endswith(".beta"):
            theta[off:off + n] = 0.1 * rng.standard_normal(n)
        if name.endswith(".gamma"):
            theta[off:off + n] = 1.0 + 0.1 * rng.standard_normal(n)
    rm = 0.1 * rng.standard_normal(rm.shape); rv = 1.0 + 0.2 * rng.random(rv.shape)
    pol.set_parameters(theta, rm, rv)
    o.theta[:] = theta; o.running_mean[:] = rm; o.running_var[:] = rv
    return pol, o


@pytest.mark.parametrize("name", ["bn", "plain"])
def test_policy_matches_reference_vectors(name):
    """The reference's own module (CPU fp32): train-mode predictions and losses of three Adam steps,
    final parameters, eval-mode prediction."""
    from iterative_learning_nmpc_amd.policy import DevicePolicy
    g = np.load(os.path.join(GOLD, f"policy_{name}.npz"))
    n_in, n_out, L, hidden, bn, batch, steps = [int(v) for v in g["dims"]]
    pol = DevicePolicy(n_in, n_out, L, hidden, bool(bn), batch_max=batch, seed=None)
    pol.load_state_dict({k[5:]: g[k] for k in g.files if k.startswith("init.")})
    dev = pol.device
    for s in range(steps):
        loss, pred = pol.train_step(torch.tensor(g["X"][s], device=dev), torch.tensor(g["Y"][s], device=dev),
                                    float(g["lr"]), return_pred=True)
        assert abs(loss.item() - g["train_loss"][s]) < 5e-6
        assert np.abs(pred.cpu().numpy() - g["train_pred"][s]).max() < 5e-5
    ref = DevicePolicy(n_in, n_out, L, hidden, bool(bn), batch_max=batch, seed=None)
    ref.load_state_dict({k[6:]: g[k] for k in g.files if k.startswith("final.")})
    th, rm, rv = (t.cpu().numpy() for t in pol.get_parameters())
    th_ref, rm_ref, rv_ref = (t.cpu().numpy() for t in ref.get_parameters())
    noise = np.zeros(pol.n_theta, bool)                  # biases in front of a BatchNorm: exact gradient zero, Adam moves them on noise
    if bn:
        for n_, s_, off in pol.items:
            if n_.endswith(".b") and int(n_.split(".")[1]) < L:
                noise[off:off + int(np.prod(s_))] = True
    d = np.abs(th - th_ref)
    assert d[~noise].max() < 5e-5 and (not noise.any() or d[noise].max() <= 2.01 * steps * float(g["lr"]))
    if bn:
        assert np.abs(rv - rv_ref).max() < 1e-5 and np.abs(rm - rm_ref).max() < 0.1 * 3 * steps * float(g["lr"])
    assert np.abs(ref.forward(torch.tensor(g["X"][0], device=dev)).cpu().numpy() - g["eval_pred"]).max() < 5e-5


@pytest.mark.parametrize("n_in,n_out,L,hidden,bn,B", [(47, 12, 3, 512, True, 256), (47, 12, 3, 512, True, 1000),
                                                       (5, 3, 1, 7, True, 2), (9, 4, 2, 65, False, 33),
                                                       (130, 70, 2, 100, True, 129)])
def test_policy_forward_and_train_step_match_oracle(n_in, n_out, L, hidden, bn, B):
    """Full-size network of the reference's configuration (47 -> 3 x 512 -> 12, BatchNorm, batch 256 / 1000)
    and ragged shapes (tile remainders in every GEMM dimension)."""
    pol, o = _pair(n_in, n_out, L, hidden, bn, batch_max=B + 3)
    rng = np.random.default_rng(5)
    dev = pol.device
    X = rng.standard_normal((B, n_in)); Y = rng.standard_normal((B, n_out))
    y_eval = pol.forward(torch.tensor(X, dtype=torch.float32, device=dev)).cpu().numpy()
    assert rel(y_eval, o.forward(X, train=False)) < 1e-5
    for step in range(2):
        loss, pred = pol.train_step(torch.tensor(X, dtype=torch.float32, device=dev),
                                    torch.tensor(Y, dtype=torch.float32, device=dev), 1e-3, return_pred=True)
        lo, po, go = o.train_step(X, Y, 1e-3)
        # step 0 is a pure fp32-vs-fp64 comparison; step 1 runs on parameters that Adam moved by ~lr in
        # the direction of sign(gradient), which is rounding noise wherever the exact gradient is ~0
        assert abs(loss.item() - lo) < (1e-5 if step == 0 else 2e-3) * max(1.0, lo)
        assert rel(pred.cpu().numpy(), po) < (2e-5 if step == 0 else 3e-3)
        if step == 0:
            th, rm, rv = (t.cpu().numpy() for t in pol.get_parameters())
            d = np.abs(th - o.theta)
            strong = np.abs(go) > 1e-2 * np.abs(go).max()      # an fp32 gradient there is good to ~1e-4 relative -> lr x that
            assert d[strong].max() < 2e-5, d[strong].max()
            assert d.max() <= 2.01e-3                           # everything else moved by at most lr
            if bn:
                assert rel(rv, o.running_var) < 1e-5 and rel(rm, o.running_mean) < 1e-5
    # the updated network still agrees in eval mode (to the Adam noise explained above)
    assert rel(pol.forward(torch.tensor(X, dtype=torch.float32, device=dev)).cpu().numpy(), o.forward(X, train=False)) < 1e-2


@pytest.mark.parametrize("n_in,n_out,L,hidden,bn,B", [(47, 12, 3, 512, True, 1000), (9, 4, 2, 65, False, 33)])
def test_training_is_reproducible_run_to_run(n_in, n_out, L, hidden, bn, B):
    """Five Adam steps from the same parameters on the same batches, twice: parameters, running statistics and losses
    agree bit for bit (the split-K weight-gradient GEMMs sum their partial products in split order, no float atomics)."""
    rng = np.random.default_rng(11)
    X = torch.tensor(rng.standard_normal((5, B, n_in)), dtype=torch.float32, device="cuda:0")
    Y = torch.tensor(rng.standard_normal((5, B, n_out)), dtype=torch.float32, device="cuda:0")
    runs = []
    for _ in range(2):
        pol, _o = _pair(n_in, n_out, L, hidden, bn, batch_max=B)
        losses = [pol.train_step(X[s], Y[s], 1e-3).item() for s in range(5)]
        runs.append((losses, [t.cpu().numpy() for t in pol.get_parameters()]))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert np.array_equal(a, b), (np.abs(a - b).max(), int((a != b).sum()), a.size)


def test_policy_argument_errors():
    from iterative_learning_nmpc_amd import _lib
    from iterative_learning_nmpc_amd.policy import DevicePolicy
    with pytest.raises(_lib.NmpcError):
        DevicePolicy(0, 3, 1, 8, True)
    pol = DevicePolicy(4, 3, 1, 8, True, batch_max=8)
    x = torch.zeros(9, 4, device=pol.device); y = torch.zeros(9, 3, device=pol.device)
    with pytest.raises(_lib.NmpcError, match="batch_max"):
        pol.forward(x)
    with pytest.raises(_lib.NmpcError):
        pol.train_step(x[:1].contiguous(), y[:1].contiguous())          # BatchNorm needs two rows
    with pytest.raises(_lib.NmpcError):
        pol.train_step(x[:4].contiguous(), y[:4].contiguous(), lr=0.0)


@pytest.mark.parametrize("n,num", [(1000, 4096), (5000, 100), (2048 * 3 + 17, 9000), (1, 7)])
def test_weighted_sampler_is_bit_exact(n, num):
    """Index work: the device sampler equals the oracle's restatement (same Philox numbers, same
    inverse-CDF lookup) for the reference's weights (1 or 5: the fp64 prefix sums are exact)."""
    from iterative_learning_nmpc_amd.policy import weighted_sample
    from oracle.policy_oracle import weighted_sample as oracle_sample
    rng = np.random.default_rng(n)
    w = np.where(rng.random(n) < 0.15, 5.0, 1.0).astype(np.float32)
    idx = weighted_sample(torch.tensor(w, device="cuda:0"), num, seed=1234567891011).cpu().numpy()
    assert np.array_equal(idx, oracle_sample(w, num, 1234567891011))


def test_weighted_sampler_follows_tracking_error_weights_and_gathers_batches():
    """nmpc_tracking_error's OOD weights -> sampler -> batch of rows (the reference's data loader,
    test_train_policy.py:128-134): frequencies follow the weights, the gathered rows are the table's."""
    from iterative_learning_nmpc_amd.policy import gather_rows, weighted_sample
    from iterative_learning_nmpc_amd.solver import tracking_error
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    B, T, ns = 40, 50, 19
    S = torch.tensor(rng.normal(0, 1.0, (B, T, ns)), dtype=torch.float32, device=dev)
    err, wgt = tracking_error(S, S[0].contiguous(), threshold=6.0, ood_weight=5.0)
    w = wgt.reshape(-1)
    ood = (w > 1.0)
    assert 0.02 < ood.float().mean().item() < 0.98
    idx = weighted_sample(w, 400000, seed=5)
    hit = ood[idx.long()].float().mean().item()
    expect = (5.0 * ood.sum() / (5.0 * ood.sum() + (~ood).sum())).item()
    assert abs(hit - expect) < 0.005
    table = S.reshape(B * T, ns).contiguous()
    batch = gather_rows(table, idx[:1024].contiguous())
    assert torch.equal(batch, table[idx[:1024].long()])
    bad = gather_rows(table, torch.tensor([3, B * T, -1], dtype=torch.int32, device=dev))      # out of range: not read, NaN
    assert torch.equal(bad[0], table[3]) and torch.isnan(bad[1:]).all()
    # arbitrary positive weights: statistical agreement (prefix sums differ in the last bits)
    w2 = torch.tensor(rng.random(3000).astype(np.float32) + 0.01, device=dev)
    i2 = weighted_sample(w2, 600000, seed=9).cpu().numpy()
    freq = np.bincount(i2, minlength=3000) / 600000.0
    p = (w2 / w2.sum()).cpu().numpy()
    assert np.abs(freq - p).max() < 5 * np.sqrt(p.max() / 600000.0)
