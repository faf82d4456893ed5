"""GPU parity of the whole-body kernel family (BASELINE configs[2]: nx = 42, nu = 30) through the C-ABI against
the CPU oracle on identical seeded inputs.  Tolerance: 1e-5 relative L2 (north_star), asserted against the fp64
oracle; the distance to the fp32 oracle (same declared algorithm in float, its own operation order) is checked too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from iterative_learning_nmpc_amd import workloads as wl  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


def _solver(w, B, dev, **opts):
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    s = BatchedNmpcSolver(w.model_id, w.N, B, dev)
    s.set_model_params(w.mp)
    s.set_cost_weights(w.W, w.W_e, w.meta["reg"], w.meta["reg_e"])
    s.set_max_iter(opts.get("max_sqp_iter", 1))
    s.set_max_qp_iter(opts.get("n_ipm", 6))
    s.set_nlp_tol(opts.get("nlp_tol", 0.0))
    return s


def _gpu_solve(s, w, shift=0, X=None, U=None):
    t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params")}
    Xd, Ud = s.to_device(w.X if X is None else X), s.to_device(w.U if U is None else U)
    Xd, Ud, st, stats = s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], Xd, Ud, shift=shift)
    torch.cuda.synchronize()
    return Xd.cpu().numpy(), Ud.cpu().numpy(), st.cpu().numpy(), stats.cpu().numpy()


def _oracle_solve(o, w, X=None, U=None, **opts):
    kw = dict(max_sqp_iter=1, n_ipm=6, yref_per_stage=int(w.yref.ndim == 3), reg=w.meta["reg"], reg_e=w.meta["reg_e"])
    kw.update(opts)
    return o.solve_batch(w.model_id, w.N, w.mp, o.opt(**kw), w.W, w.W_e, w.x0, w.yref, w.yref_e, w.params,
                         w.X if X is None else X, w.U if U is None else U)


# (n_ipm, sqp): no inequalities / the reference's steady-state policy (mpc_opt.py:25-27) / a few SQP iterations /
# the reference's first-solve policy (mpc.py:464-473)
@pytest.mark.parametrize("n_ipm,sqp", [(0, 1), (6, 1), (6, 3), (6, 15)])
def test_wholebody_solve_parity(dev, oracle64, oracle32, n_ipm, sqp):
    B = 64
    w = wl.wholebody_trot(B=B, N=30, seed=0)
    s = _solver(w, B, dev, n_ipm=n_ipm, max_sqp_iter=sqp)
    X, U, st, stats = _gpu_solve(s, w)
    Xo, Uo, sto, statso = _oracle_solve(oracle64, w, n_ipm=n_ipm, max_sqp_iter=sqp)
    assert np.array_equal(st, sto)
    assert np.isfinite(X).all() and np.isfinite(U).all()
    # measured: 7.3e-6 / 3.5e-6 at (6, 1); 6e-7 / 1e-6 at (6, 3): the iteration contracts rounding differences
    assert rel(X, Xo) < 1e-5 and rel(U, Uo) < 1e-5, (rel(X, Xo), rel(U, Uo))
    assert np.allclose(stats[:, 0], statso[:, 0], rtol=2e-5)                   # cost at the last linearisation
    X32, U32, st32, _ = _oracle_solve(oracle32, w, n_ipm=n_ipm, max_sqp_iter=sqp)
    # fp32 oracle: same algorithm in float with its own summation order -- both sit at the fp32 floor around fp64
    assert rel(X, X32) < 2e-5 and rel(U, U32) < 2e-5, (rel(X, X32), rel(U, U32))
    assert rel(X32, Xo) < 1e-5, rel(X32, Xo)


def test_wholebody_active_friction_pyramid(dev, oracle64):
    """Low friction and a commanded forward speed: pyramid faces are active in the solution; parity holds."""
    B = 32
    w = wl.wholebody_trot(B=B, N=30, seed=3)
    w.mp = w.mp.copy(); w.mp[6] = 0.15
    w.yref = w.yref.copy(); w.yref[:, :, 6] = 1.0
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=4)
    X, U, st, _ = _gpu_solve(s, w)
    Xo, Uo, sto, _ = _oracle_solve(oracle64, w, n_ipm=6, max_sqp_iter=4)
    assert np.array_equal(st, sto)
    f = Uo[:, :, 18:].reshape(B, 30, 4, 3)
    stance = w.params[:, :30, :4] > 0.5
    ratio = np.abs(f[..., 0]) / np.maximum(0.15 * f[..., 2], 1e-9)
    assert (ratio[stance] > 0.9).any(), "test needs binding pyramid faces"
    assert rel(X, Xo) < 1e-5 and rel(U, Uo) < 1e-5, (rel(X, Xo), rel(U, Uo))
    assert (np.abs(U[:, :, 18:].reshape(B, 30, 4, 3)[~stance]) < 1e-3).all()     # swing feet carry no force


def test_wholebody_any_contact_pattern(dev, oracle64):
    """The elimination has static variants for the patterns of a trot (diagonal pairs, four-foot stance, flight) and the
    full-mask code for everything else: random per-problem, per-stage patterns (pace, bound, three-legged ...) exercise
    every path, including the hand-over between variants from stage to stage."""
    B = 48
    w = wl.wholebody_trot(B=B, N=30, seed=13)
    rng = np.random.default_rng(5)
    c = (rng.random((B, 31, 4)) < 0.6).astype(np.float64)
    c[: B // 3] = w.params[: B // 3, :, :4]                                  # a third keeps its trot schedule
    w.params = w.params.copy()
    w.params[:, :, :4] = c
    w.params[:, :, 4:8] = 1.0 - c
    n_st = np.maximum(c[:, :30].sum(-1, keepdims=True), 1.0)
    w.U = w.U.copy(); w.yref = w.yref.copy()
    w.U[:, :, 18:] = 0.0
    w.U[:, :, 20::3] = c[:, :30] * (-w.mp[5] * w.mp[1]) / n_st
    w.yref[:, :, 52:64] = w.U[:, :, 18:]
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=2)
    X, U, st, _ = _gpu_solve(s, w)
    Xo, Uo, sto, _ = _oracle_solve(oracle64, w, n_ipm=6, max_sqp_iter=2)
    assert np.array_equal(st, sto)
    assert rel(X, Xo) < 1e-5 and rel(U, Uo) < 1e-5, (rel(X, Xo), rel(U, Uo))


@pytest.mark.parametrize("B,N", [(1, 30), (3, 25), (65, 30), (5, 7), (2, 64), (2, 43), (2, 44), (3, 6), (2, 5)])
def test_wholebody_odd_batches_and_horizons(dev, oracle64, B, N):
    """ragged batches; the reference's own horizon (25 nodes, mpc_opt.py:11-13); the 64-lane limit of lane = stage; the horizons
    either side of the LDS budget that keeps the gains of two (N <= 43) or one (N >= 44) backward stages on the CU, and horizons
    no longer than that"""
    w = wl.wholebody_trot(B=B, N=N, seed=7)
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=2)
    X, U, st, stats = _gpu_solve(s, w)
    Xo, Uo, sto, statso = _oracle_solve(oracle64, w, n_ipm=6, max_sqp_iter=2)
    assert np.array_equal(st, sto)
    assert rel(X, Xo) < 1e-5 and rel(U, Uo) < 1e-5, (rel(X, Xo), rel(U, Uo))
    # cost at the last linearisation, every node counted (N = 64: node 64 has no lane of its own), and the iteration count
    assert np.allclose(stats[:, 0], statso[:, 0], rtol=2e-5) and np.array_equal(stats[:, 3], statso[:, 3])


def test_wholebody_foot_placement_cost(dev, oracle64):
    """the reference's contact-restricted cost (pos_cost with W_foot_displacement = 1e3, solver.py:128-137,272-273) in the
    solved model: eight more rows of the dense residual Jacobian, inside the two K tiles the contraction runs anyway.
    Parity with the oracle; and a handle whose weights go back to zero solves exactly what a fresh handle solves (the rows
    of the Jacobian image are exact zeros again)."""
    B = 32
    w = wl.wholebody_trot(B=B, N=30, seed=8, foot_placement=1.0e3)
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=3)
    X, U, st, stats = _gpu_solve(s, w)
    Xo, Uo, sto, statso = _oracle_solve(oracle64, w, n_ipm=6, max_sqp_iter=3)
    assert np.array_equal(st, sto)
    assert rel(X, Xo) < 1e-5 and rel(U, Uo) < 1e-5, (rel(X, Xo), rel(U, Uo))
    assert np.allclose(stats[:, 0], statso[:, 0], rtol=2e-5)
    w0 = wl.wholebody_trot(B=B, N=30, seed=8)
    assert rel(X, _oracle_solve(oracle64, w0, n_ipm=6, max_sqp_iter=3)[0]) > 1e-3          # the cost term does something
    s.set_cost_weights(w0.W, w0.W_e, w0.meta["reg"], w0.meta["reg_e"])
    X0, U0, _, _ = _gpu_solve(s, w0)
    fresh = _solver(w0, B, dev, n_ipm=6, max_sqp_iter=3)
    Xf, Uf, _, _ = _gpu_solve(fresh, w0)
    assert np.array_equal(X0, Xf) and np.array_equal(U0, Uf)


def test_wholebody_stage_constant_reference(dev, oracle64):
    """yref[B, ny]: one reference row for all stages, as the reference sets it (solver.py:169)"""
    B = 8
    w = wl.wholebody_trot(B=B, N=30, seed=2)
    w.yref = np.ascontiguousarray(w.yref[:, 0, :])
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=2)
    X, U, st, _ = _gpu_solve(s, w)
    Xo, Uo, sto, _ = _oracle_solve(oracle64, w, n_ipm=6, max_sqp_iter=2)
    assert np.array_equal(st, sto)
    assert rel(X, Xo) < 1e-5 and rel(U, Uo) < 1e-5


def test_wholebody_warm_start_shift_folded(dev, oracle64):
    """receding horizon: solve, then shift by one node and solve again (solver.py:304-322 then :396-403); the folded
    shift equals shifting the oracle's own previous solution first"""
    B = 16
    w = wl.wholebody_trot(B=B, N=30, seed=5)
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=1)
    X1, U1, _, _ = _gpu_solve(s, w)
    X2, U2, st2, _ = _gpu_solve(s, w, shift=1, X=X1, U=U1)
    Xo1, Uo1, _, _ = _oracle_solve(oracle64, w)
    Xs, Us = oracle64.shift_warm_start(Xo1, Uo1, 1)
    Xo2, Uo2, sto2, _ = _oracle_solve(oracle64, w, X=Xs, U=Us)
    assert np.array_equal(st2, sto2)
    assert rel(X2, Xo2) < 1e-5 and rel(U2, Uo2) < 1e-5, (rel(X2, Xo2), rel(U2, Uo2))
    # shift of the whole horizon: nothing of the old inputs survives
    X3, U3, _, _ = _gpu_solve(s, w, shift=30, X=X1, U=U1)
    Xs, Us = oracle64.shift_warm_start(Xo1, Uo1, 30)
    Xo3, Uo3, _, _ = _oracle_solve(oracle64, w, X=Xs, U=Us)
    assert rel(X3, Xo3) < 1e-5 and rel(U3, Uo3) < 1e-5


def test_wholebody_shift_zeroes_forces_and_keeps_accelerations(dev, oracle64):
    """solver.py:316-322: the warm-start shift moves a[:, :n_warm_start] and f[:, :n_warm_start] and zeroes f[:, n_warm_start:] --
    the acceleration tail keeps the previous solution's values.  The device shift (nmpc_shift_warm_start) and the oracle's agree
    bit for bit (the facade's `warm_start_solver` on its views against the oracle: tests/test_facade.py)."""
    B, N, shift = 3, 30, 4
    rng = np.random.default_rng(21)
    X = rng.standard_normal((B, N + 1, 42)).astype(np.float32)
    U = rng.standard_normal((B, N, 30)).astype(np.float32)
    w = wl.wholebody_trot(B=B, N=N, seed=0)
    s = _solver(w, B, dev)
    Xd, Ud = s.to_device(X), s.to_device(U)
    s.warm_start_solver(Xd, Ud, shift)
    torch.cuda.synchronize()
    Xo, Uo = oracle64.shift_warm_start(X, U, shift)
    assert np.array_equal(Xd.cpu().numpy(), Xo.astype(np.float32)) and np.array_equal(Ud.cpu().numpy(), Uo.astype(np.float32))
    assert np.array_equal(Uo[:, N - shift:, :18], U[:, N - shift:, :18]) and (Uo[:, N - shift:, 18:] == 0).all()
    assert np.array_equal(Uo[:, :N - shift], U[:, shift:])


def test_wholebody_solves_are_bit_reproducible(dev):
    """Two solves of the same inputs give the same bits, run to run and across the contact patterns of a batch -- the guard
    for the two scheduling hazards DESIGN.md 5b records (an inline-asm v_readlane behind a VALU write, MFMA results joined
    behind a uniform branch), which showed up as a run-to-run varying 5e-5 error."""
    B = 256
    w = wl.wholebody_trot(B=B, N=30, seed=4)
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=2)
    ref = None
    for _ in range(3):
        X, U, st, stats = _gpu_solve(s, w)
        if ref is None:
            ref = (X, U, st, stats)
        assert np.array_equal(X, ref[0]) and np.array_equal(U, ref[1]) and np.array_equal(st, ref[2]) and np.array_equal(stats, ref[3])
    X, U, _, _ = _gpu_solve(s, w, shift=1, X=ref[0], U=ref[1])
    X2, U2, _, _ = _gpu_solve(s, w, shift=1, X=ref[0], U=ref[1])
    assert np.array_equal(X, X2) and np.array_equal(U, U2)


def test_wholebody_early_exit_and_status(dev, oracle64):
    """status 0 once max|step| < nlp_tol, else 2; the same problems stop at the same iteration as the oracle"""
    B = 24
    w = wl.wholebody_trot(B=B, N=30, seed=9, sigma_joint=0.05)
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=12, nlp_tol=2e-2)
    X, U, st, stats = _gpu_solve(s, w)
    Xo, Uo, sto, statso = _oracle_solve(oracle64, w, n_ipm=6, max_sqp_iter=12, nlp_tol=2e-2)
    assert set(np.unique(sto)) <= {0, 2}
    same = stats[:, 3] == statso[:, 3]
    # a problem whose step norm crosses the tolerance within rounding may stop one iteration apart: enumerate them
    for b in np.nonzero(~same)[0]:
        assert abs(stats[b, 3] - statso[b, 3]) == 1 and abs(statso[b, 1] - 2e-2) < 2e-3, (b, stats[b], statso[b])
    assert np.array_equal(st[same], sto[same])
    assert rel(X[same], Xo[same]) < 1e-5 and rel(U[same], Uo[same]) < 1e-5


def test_wholebody_nan_input_is_reported_not_propagated(dev):
    B = 4
    w = wl.wholebody_trot(B=B, N=30, seed=1)
    w.x0 = w.x0.copy(); w.x0[2, 7] = np.nan
    w.X = w.X.copy(); w.X[2, :, 7] = np.nan
    s = _solver(w, B, dev)
    X, U, st, _ = _gpu_solve(s, w)
    assert st[2] == 1 and (st[[0, 1, 3]] == 2).all()
    assert np.isfinite(X[[0, 1, 3]]).all()


def test_wholebody_full_size_properties(dev, oracle64):
    """BASELINE configs[2] at full size (B = 8192, N = 30): size-independent properties of one steady-state solve,
    and the first problems against the oracle."""
    B = 8192
    w = wl.wholebody_trot(B=B, N=30, seed=11)
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=1)
    X, U, st, stats = _gpu_solve(s, w)
    assert (st == 2).all() and np.isfinite(X).all() and np.isfinite(U).all()
    # a full step lands on the measured state: X[:, 0] = x0
    assert np.abs(X[:, 0] - w.x0).max() < 1e-5
    # the step satisfies the LINEARISED dynamics exactly: kinematic rows are linear, so after a full step
    # q+ = q + dt v+ and v+ = v + dt a hold to rounding on the new iterate
    dt = w.mp[0]
    vn = X[:, :-1, 18:36] + dt * U[:, :, :18]
    assert np.abs(X[:, 1:, 18:36] - vn).max() < 2e-3 * max(1.0, np.abs(vn).max())
    assert np.abs(X[:, 1:, :18] - (X[:, :-1, :18] + dt * X[:, 1:, 18:36])).max() < 1e-3
    # swing feet end without force, stance feet inside their pyramid (interior point: strictly)
    f = U[:, :, 18:].reshape(B, 30, 4, 3)
    stance = w.params[:, :30, :4] > 0.5
    assert np.abs(f[~stance]).max() < 1e-2
    assert (np.abs(f[..., 0])[stance] <= 0.8 * f[..., 2][stance] + 1e-3).all()
    n = 8
    import copy
    w8 = copy.copy(w)
    for k in ("x0", "yref", "yref_e", "params", "X", "U"):
        setattr(w8, k, getattr(w, k)[:n])
    Xo, Uo, sto, _ = _oracle_solve(oracle64, w8)
    assert rel(X[:n], Xo) < 1e-5 and rel(U[:n], Uo) < 1e-5


@pytest.mark.parametrize("precision,sqp,lo,hi", [(3, 1, 0.0, 1e-5), (3, 15, 0.0, 1e-5),
                                                 (1, 1, 1e-3, 1e-2), (1, 15, 8e-4, 8e-3), (2, 1, 4e-5, 5e-4), (2, 15, 0.0, 1e-5)])
def test_wholebody_mixed_precision(dev, oracle64, precision, sqp, lo, hi):
    """BASELINE configs[4] on the model that has a dense Gauss-Newton contraction: the scaled residual Jacobian in bf16,
    Q~ = Js'Js on the bf16 matrix pipe (v_mfma_f32_16x16x16_bf16) with fp32 accumulation, fp32 Riccati.
    Precision 3, the shipped recommendation: Js = hi + mid + lo (three bf16 numbers carry the 24 bits of the fp32
    value), six products -- INSIDE the 1e-5 bar at the steady-state policy the bench runs (1 SQP x 6 IPM) and at the
    first-solve policy.  Documented negatives kept as bands (DESIGN.md 7, B = 64): plain bf16 (precision 1) sits at 3.5e-3
    after one SQP iteration and 2.4e-3 converged -- 8 bits of mantissa in the Hessian; two-way split bf16 (precision 2) at
    1.4e-4 after one iteration and 3.7e-6 converged."""
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    B = 64
    w = wl.wholebody_trot(B=B, N=30, seed=0)
    s = BatchedNmpcSolver(w.model_id, w.N, B, dev, precision=precision)
    s.set_model_params(w.mp)
    s.set_cost_weights(w.W, w.W_e, w.meta["reg"], w.meta["reg_e"])
    s.set_max_iter(sqp)
    X, U, st, _ = _gpu_solve(s, w)
    Xo, Uo, sto, _ = _oracle_solve(oracle64, w, max_sqp_iter=sqp)
    eX, eU = rel(X, Xo), rel(U, Uo)
    print(f"precision {precision}, {sqp} SQP: rel-L2 X {eX:.2e} U {eU:.2e}")
    assert np.array_equal(st, sto)
    assert lo <= eX < hi and eU < hi, (eX, eU)


def test_wholebody_api_limits(dev):
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    from iterative_learning_nmpc_amd._lib import NmpcError
    with pytest.raises(NmpcError):
        BatchedNmpcSolver(wl.MODEL_WHOLEBODY, 65, 4, dev)            # lane = stage phases: N <= 64
    with pytest.raises(NmpcError):
        BatchedNmpcSolver(wl.MODEL_WHOLEBODY, 30, 4, dev, precision=4)
    with pytest.raises(NmpcError):
        BatchedNmpcSolver(wl.MODEL_CENTROIDAL, 50, 4, dev, precision=2)      # split bf16 exists for the whole-body contraction only
    with pytest.raises(NmpcError):
        BatchedNmpcSolver(wl.MODEL_CENTROIDAL, 50, 4, dev, precision=3)
    w = wl.wholebody_trot(B=2, N=30, seed=0)
    s = _solver(w, 2, dev)
    s.set_line_search(True)
    with pytest.raises(NmpcError):
        _gpu_solve(s, w)                                               # the whole-body model takes full steps


def _robot_state(seed=0):
    from iterative_learning_nmpc_amd import wholebody as wbk
    rng = np.random.default_rng(seed)
    q = np.zeros(18); v = np.zeros(18)
    q[:2] = rng.normal(0, 0.05, 2); q[2] = 0.29; q[3:6] = rng.normal(0, 0.05, 3); q[6:] = wbk.Q_HOME + rng.normal(0, 0.1, 12)
    v[:6] = rng.normal(0, 0.2, 6); v[6:] = rng.normal(0, 0.2, 12)
    return q, v


def test_facade_optimize_is_the_oracle_solution_of_its_packed_problem(dev, oracle64):
    """`LocomotionMPC.optimize` through the reference-shaped facade (mpc.py:317-369 -> solver.py:355-429): the first
    solve with the reference's first-solve policy (15 SQP, tolerances / 10: mpc.py:464-473), then a warm-started
    steady-state solve one node later -- both equal the oracle's solution of the arrays `update_solver` packed."""
    from iterative_learning_nmpc_amd.mpc_wholebody import LocomotionMPC
    mpc = LocomotionMPC(print_info=False, n_nodes=30, device=dev, force_reference="gravity_share")
    mpc.set_command(np.array([0.3, 0.0, 0.0]), 0.0)
    q, v = _robot_state(0)
    s = mpc.solver

    def oracle_of_packed(max_sqp, nlp_tol):
        p = s._problem
        W, We = s._W, s._W_e
        opt = oracle64.opt(max_sqp_iter=max_sqp, n_ipm=6, nlp_tol=nlp_tol, yref_per_stage=1, reg=mpc.config_cost.reg_eps,
                           reg_e=mpc.config_cost.reg_eps_e)
        return oracle64.solve_batch(2, 30, s.mp, opt, W, We, p["x0"], p["yref"], p["yref_e"], p["params"], p["X"], p["U"])

    mpc.set_convergence_on_first_iter()
    q_sol, v_sol, a_sol, f_sol, dt_sol = mpc.optimize(q, v)
    assert q_sol.shape == (31, 18) and v_sol.shape == (31, 18) and a_sol.shape == (30, 18) and f_sol.shape == (30, 4, 3)
    assert np.allclose(dt_sol, s.dt_nodes) and np.allclose(q_sol[0], q, atol=1e-5)
    Xo, Uo, sto, statso = oracle_of_packed(15, 0.01)
    assert s.status[0] == sto[0] and s.stats[0, 3] == statso[0, 3]
    assert rel(q_sol, Xo[0, :, :18]) < 1e-5 and rel(a_sol, Uo[0, :, :18]) < 1e-5 and rel(f_sol.reshape(30, 12), Uo[0, :, 18:]) < 1e-5
    mpc.first_solve = False
    mpc.sim_step, mpc.current_opt_node = mpc.replanning_steps, 1
    mpc.set_convergence_on_first_iter()
    q2, v2, a2, f2, _ = mpc.optimize(q_sol[1].copy(), v_sol[1].copy())
    Xo2, Uo2, sto2, _ = oracle_of_packed(1, 0.1)
    assert s.last_node == 1 and s.status[0] == sto2[0]
    assert rel(q2, Xo2[0, :, :18]) < 1e-5 and rel(f2.reshape(30, 12), Uo2[0, :, 18:]) < 1e-5
    # the same packed problem through the batched solver: the facade adds nothing of its own
    X, U, st, _ = _gpu_solve(s._device_solver(), type("W", (), dict(
        x0=s._problem["x0"], yref=s._problem["yref"], yref_e=s._problem["yref_e"], params=s._problem["params"],
        X=s._problem["X"], U=s._problem["U"])))
    assert np.array_equal(X[0, :, :18].astype(np.float64), q2)


def test_facade_open_loop_keeps_the_robot_up(dev):
    """0.2 s of the reference's simulator-free rollout (mpc.py:416-462): six replans, plan followed at 1 kHz"""
    from iterative_learning_nmpc_amd import wholebody as wbk
    from iterative_learning_nmpc_amd.mpc_wholebody import LocomotionMPC
    mpc = LocomotionMPC(print_info=False, device=dev, force_reference="gravity_share")   # [decl] the facade's default is the reference's zero
    mpc.set_command(np.array([0.2, 0.0, 0.0]), 0.0)
    q0 = np.zeros(18); q0[2] = 0.30; q0[6:] = wbk.Q_HOME
    traj = mpc.open_loop(q0, np.zeros(18), 0.2)
    assert traj.shape[0] in (200, 201) and traj.shape[1] == 18 and np.isfinite(traj).all()   # float clock, as the reference's loop
    assert traj[:, 2].min() > 0.22 and np.abs(traj[:, 3:6]).max() < 0.3
    assert traj[-1, 0] > 0.01                                                      # it walks forward
    # five replans (steps 0, 40, .., 160); the node counter follows the float clock of the reference's loop
    assert mpc.current_opt_node in (4, 5) and mpc.solver.last_node in (3, 4) and not mpc.first_solve


@pytest.mark.parametrize("force_reference", ["gravity_share", "zero"])
def test_wholebody_device_rollout_equals_host_driven_open_loop(dev, force_reference):
    """the reference's own problem through the reference's own loop (LocomotionMPC.open_loop,
    mpc.py:416-462) from ONE host call -- per replan the problem is assembled on the device from the plant state, solved
    with the warm-start shift folded in, the plan up-sampled and followed -- against the host-driven loop of the facade
    (numpy helpers golden-pinned; same device solver): twelve replans, every recorded 44-slot row."""
    from iterative_learning_nmpc_amd import wholebody as wbk
    from iterative_learning_nmpc_amd.mpc_wholebody import LocomotionMPC
    T = 0.45
    q0 = np.zeros(18); q0[2] = 0.30; q0[3] = 0.05; q0[6:] = wbk.Q_HOME
    v0 = np.zeros(18); v0[0] = 0.1
    out = {}
    for mode in ("host", "device"):
        mpc = LocomotionMPC(print_info=False, device=dev, force_reference=force_reference)
        mpc.set_command(np.array([0.25, 0.05, 0.0]), 0.1)
        if mode == "host":
            q_traj = mpc.open_loop(q0, v0, T)
            out[mode] = (mpc.state_rows(q_traj, mpc.v_traj), mpc.current_opt_node, mpc.solver.last_node, mpc.base_ref_vel_tracking.copy(),
                         mpc.solver.q_sol_euler.copy())
        else:
            S = mpc.open_loop_device(q0, v0, T)
            torch.cuda.synchronize()
            out[mode] = (S.cpu().numpy()[0], mpc.current_opt_node, mpc.solver.last_node, mpc.base_ref_vel_tracking.copy(),
                         mpc.solver.q_sol_euler.copy())
            assert int(mpc.failed.cpu().numpy()[0] & 0xFF & ~16) == 0
    (Sh, nh, lh, rh, qh), (Sd, nd, ld, rd, qd) = out["host"], out["device"]
    assert Sh.shape == Sd.shape and Sh.shape[1] == 44 and Sh.shape[0] >= 440
    assert (nh, lh) == (nd, ld), ((nh, lh), (nd, ld))                        # same float clock, same replanning nodes
    assert np.array_equal(Sd[:, 0], Sh[:, 0].astype(np.float32))               # gait phase
    e = rel(Sd, Sh)
    print(f"whole-body device rollout vs host-driven open_loop ({force_reference}): rel-L2 {e:.2e} over {Sh.shape[0]} rows")
    assert e < 2e-5, e
    assert rel(qd, qh) < 2e-5                                                  # the last plan, in the facade's views
    if force_reference == "gravity_share":
        assert Sd[:, 19].min() > 0.22                                          # it stays up


def test_wholebody_device_rollouts_batch_and_termination(dev):
    """a batch of pushed whole-body rollouts: batch-size independence (a rollout is bit for bit what it is alone), flags,
    early termination on the collision height, finite rows"""
    from iterative_learning_nmpc_amd import wholebody as wbk
    from iterative_learning_nmpc_amd.mpc_wholebody import LocomotionMPC
    B, T = 24, 0.8
    rng = np.random.default_rng(2)
    q0 = np.zeros((B, 18)); q0[:, 2] = 0.30; q0[:, 6:] = wbk.Q_HOME + rng.normal(0, 0.03, (B, 12))
    v0 = np.zeros((B, 18))
    force = rng.uniform(-1, 1, (B, 3)); force /= np.linalg.norm(force, axis=1, keepdims=True); force *= rng.uniform(50, 70, (B, 1))
    force[0] = 0.0
    force[1] = [0.0, 0.0, -70.0]
    push = dict(start=0.2, duration=0.3, force=force)
    mpc = LocomotionMPC(print_info=False, device=dev, batch=B, n_nodes=30, force_reference="gravity_share")
    mpc.set_command(np.array([0.2, 0.0, 0.0]), 0.0)
    S = mpc.open_loop_device(q0, v0, T, push=push, record_sim_steps=False)
    torch.cuda.synchronize()
    Sn, f = S.cpu().numpy(), mpc.failed.cpu().numpy()
    assert Sn.shape[0] == B and Sn.shape[1] in (20, 21) and Sn.shape[2] == 44 and np.isfinite(Sn).all()     # (the float clock of the reference's loop)
    assert (f & 1 == 0).all(), f                                               # no solver failure
    assert f[0] >> 8 == 0 and (f[0] & 0x6F) == 0                               # the unpushed rollout: nothing but velocity tracking at the start
    quat = Sn[:, :, 20:24]
    assert np.allclose(np.linalg.norm(quat, axis=-1), 1.0, atol=1e-5)
    for b in np.nonzero(f >> 8)[0]:                                             # terminated ones are frozen from the terminating row on
        i = (f[b] >> 8) - 1
        assert (Sn[b, i:] == Sn[b, i]).all() and Sn[b, i, 19] < 0.08
    n = 3
    small = LocomotionMPC(print_info=False, device=dev, batch=n, n_nodes=30, force_reference="gravity_share")
    small.set_command(np.array([0.2, 0.0, 0.0]), 0.0)
    Ss = small.open_loop_device(q0[:n], v0[:n], T, push=dict(push, force=force[:n]), record_sim_steps=False)
    assert np.array_equal(Ss.cpu().numpy(), Sn[:n]) and np.array_equal(small.failed.cpu().numpy(), f[:n])


def test_wholebody_device_rollouts_full_size(dev):
    """BASELINE-sized: 8192 pushed whole-body rollouts of 2 s (50 replans) from one call per replan, no host round trip inside
    a rollout.  Size-independent properties: finite rows, unit quaternions, no solver failure, terminated rollouts frozen below
    the collision height, copies of one rollout anywhere in the batch bit for bit equal, the first 16 bit for bit what they are
    in a batch of 16."""
    from iterative_learning_nmpc_amd import wholebody as wbk
    from iterative_learning_nmpc_amd.mpc_wholebody import LocomotionMPC
    B, T = 8192, 2.0
    rng = np.random.default_rng(5)
    q0 = np.zeros((B, 18)); q0[:, 2] = 0.30; q0[:, 6:] = wbk.Q_HOME + rng.normal(0, 0.03, (B, 12))
    v0 = np.zeros((B, 18))
    force = rng.uniform(-1, 1, (B, 3)); force /= np.linalg.norm(force, axis=1, keepdims=True); force *= rng.uniform(50, 70, (B, 1))
    force[0] = 0.0
    copies = np.array([1000, 4095, 4096, 8191])                  # rollout 7, again, in other waves / rounds of the launch
    q0[copies], force[copies] = q0[7], force[7]
    push = dict(start=0.2, duration=0.3, force=force)
    mpc = LocomotionMPC(print_info=False, device=dev, batch=B, n_nodes=30, force_reference="gravity_share")
    mpc.set_command(np.array([0.2, 0.0, 0.0]), 0.0)
    S = mpc.open_loop_device(q0, v0, T, push=push, record_sim_steps=False)
    torch.cuda.synchronize()
    Sn, f = S.cpu().numpy(), mpc.failed.cpu().numpy()
    assert Sn.shape[0] == B and Sn.shape[1] in (50, 51) and Sn.shape[2] == 44 and np.isfinite(Sn).all()
    assert (f & 1 == 0).all(), np.nonzero(f & 1)[0][:10]                      # no solver failure
    assert f[0] >> 8 == 0
    assert np.allclose(np.linalg.norm(Sn[:, :, 20:24], axis=-1), 1.0, atol=1e-5)
    term = np.nonzero(f >> 8)[0]
    print(f"whole-body rollouts, full size: {len(term)} of {B} terminated early, flag counts",
          {name: int((f & bit != 0).sum()) for name, bit in (("roll", 2), ("pitch", 4), ("height", 8), ("vel", 16), ("collision", 32), ("joint", 64))})
    for b in term[:64]:
        i = (f[b] >> 8) - 1
        assert (Sn[b, i:] == Sn[b, i]).all() and Sn[b, i, 19] < 0.08
    assert len(term) < B // 2
    for cidx in copies:
        assert np.array_equal(Sn[cidx], Sn[7]) and f[cidx] == f[7]
    n = 16
    small = LocomotionMPC(print_info=False, device=dev, batch=n, n_nodes=30, force_reference="gravity_share")
    small.set_command(np.array([0.2, 0.0, 0.0]), 0.0)
    Ss = small.open_loop_device(q0[:n], v0[:n], T, push=dict(push, force=force[:n]), record_sim_steps=False)
    assert np.array_equal(Ss.cpu().numpy(), Sn[:n]) and np.array_equal(small.failed.cpu().numpy(), f[:n])


def test_wholebody_refuses_misaligned_arrays(dev):
    """include/nmpc.h: arrays of the whole-body model are read in 8 B (parameters: 16 B) pieces -- a pointer off those boundaries is
    an argument error, not a fault; a view that starts at a problem boundary is fine."""
    from iterative_learning_nmpc_amd import workloads as wl
    from iterative_learning_nmpc_amd._lib import NmpcError
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    w = wl.wholebody_trot(B=4, N=30, seed=9)
    s = BatchedNmpcSolver(w.model_id, w.N, w.B, dev)
    s.set_model_params(w.mp); s.set_cost_weights(w.W, w.W_e, w.meta["reg"], w.meta["reg_e"])
    t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
    X, U, st, _ = s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], t["X"].clone(), t["U"].clone())
    assert (st.cpu().numpy() != 1).all()
    # the last three problems as views into the same storage: aligned, and the same result as in the full batch
    s3 = BatchedNmpcSolver(w.model_id, w.N, 3, dev)
    s3.set_model_params(w.mp); s3.set_cost_weights(w.W, w.W_e, w.meta["reg"], w.meta["reg_e"])
    X3, U3, st3, _ = s3.solve(*(t[k][1:] for k in ("x0", "yref", "yref_e", "params")), t["X"][1:].clone(), t["U"][1:].clone())
    assert torch.equal(X3, X[1:]) and torch.equal(U3, U[1:])
    # one float off
    flat = torch.zeros(t["X"].numel() + 1, dtype=torch.float32, device=dev)
    Xoff = flat[1:].view_as(t["X"])
    Xoff.copy_(t["X"])
    assert Xoff.data_ptr() % 8 == 4
    with pytest.raises(NmpcError, match="aligned"):
        s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], Xoff, t["U"].clone())
