"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on identical seeded inputs.

Tolerance (BASELINE.json north_star): 1e-5 relative L2 on trajectories.  It is asserted against the
fp64 oracle where the fp32 arithmetic allows it and otherwise against the fp32 oracle, with the
measured fp32-vs-fp64 floor quoted next to each bound.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


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
    s.set_cost_weights(w.W, w.W_e, w.meta.get("reg", 1e-6), w.meta.get("reg_e", 1e-5))
    s.set_max_iter(opts.get("max_sqp_iter", 1))
    s.set_max_qp_iter(opts.get("n_ipm", 6))
    s.set_nlp_tol(opts.get("nlp_tol", 0.0))
    s.set_line_search(opts.get("line_search", 0))
    return s


def _gpu_solve(s, w):
    t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
    X, U, st, stats = s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], t["X"], t["U"])
    torch.cuda.synchronize()
    return X.cpu().numpy(), U.cpu().numpy(), st.cpu().numpy(), stats.cpu().numpy()


def _oracle_solve(o, w, **opts):
    kw = dict(max_sqp_iter=1, n_ipm=6, yref_per_stage=1, reg=w.meta.get("reg", 1e-6),
              reg_e=w.meta.get("reg_e", 1e-5))
    kw.update(opts)
    return o.solve_batch(w.model_id, w.N, w.mp, o.opt(**kw), w.W, w.W_e, w.x0, w.yref, w.yref_e,
                         w.params, w.X, w.U)


# ------------------------------------------------------------------------------- LQ core
@pytest.mark.parametrize("nx,nu,N", [(12, 12, 50), (4, 2, 20), (7, 5, 13), (15, 16, 9), (1, 1, 3)])
def test_riccati_matches_oracle(dev, oracle64, nx, nu, N):
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    from iterative_learning_nmpc_amd.workloads import MODEL_CENTROIDAL
    rng = np.random.default_rng(nx * 100 + nu)
    B = 8
    def spd(n, m):
        M = rng.normal(0, 1, (B, m, n, n))
        return M @ M.transpose(0, 1, 3, 2) / n + np.eye(n) * 0.5
    Q, R = spd(nx, N + 1), spd(nu, N)
    q, r = rng.normal(0, 1, (B, N + 1, nx)), rng.normal(0, 1, (B, N, nu))
    A = np.eye(nx) + rng.normal(0, 0.3, (B, N, nx, nx)) / np.sqrt(nx)
    Bm = rng.normal(0, 0.5, (B, N, nx, nu))
    d, dx0 = rng.normal(0, 0.1, (B, N, nx)), rng.normal(0, 1, (B, nx))
    s = BatchedNmpcSolver(MODEL_CENTROIDAL, N, B, dev)
    args = [s.to_device(a) for a in (Q, R, q, r, A, Bm, d, dx0)]
    dX, dU, st = s.riccati(*args)
    torch.cuda.synchronize()
    dX, dU, st = dX.cpu().numpy(), dU.cpu().numpy(), st.cpu().numpy()
    assert (st == 0).all()
    for b in range(B):
        ref = oracle64.riccati(Q[b], R[b], q[b], r[b], A[b], Bm[b], d[b], dx0[b])
        assert ref["status"] == 0
        assert rel(dX[b], ref["dX"]) < 2e-5, (b, rel(dX[b], ref["dX"]))
        assert rel(dU[b], ref["dU"]) < 2e-5, (b, rel(dU[b], ref["dU"]))


def test_riccati_flags_indefinite_pivot(dev):
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    N, nx, nu, B = 4, 3, 2, 2
    s = BatchedNmpcSolver(0, N, B, dev)
    Q = np.tile(np.eye(nx), (B, N + 1, 1, 1))
    R = np.tile(np.eye(nu), (B, N, 1, 1))
    R[1, 2] = -10 * np.eye(nu)          # problem 1 has a negative-definite input weight
    z = lambda *sh: np.zeros(sh)
    A = np.tile(np.eye(nx), (B, N, 1, 1))
    Bm = np.ones((B, N, nx, nu)) * 0.1
    args = [s.to_device(a) for a in (Q, R, z(B, N + 1, nx), z(B, N, nu), A, Bm, z(B, N, nx), np.ones((B, nx)))]
    _, _, st = s.riccati(*args)
    assert st.cpu().numpy().tolist() == [0, 4]


# ------------------------------------------------------------------------------- linearisation
def test_centroidal_stage_tiles_match_oracle(dev, oracle64):
    from iterative_learning_nmpc_amd import workloads as wl
    w = wl.centroidal_trot(B=4, N=50, seed=3)
    w.X += np.random.default_rng(0).normal(0, 0.02, w.X.shape)     # non-trivial defects
    s = _solver(w, 4, dev, n_ipm=0)
    _gpu_solve(s, w)
    for b in (0, 3):
        for k in (0, 17, 49):
            xn, A, Bj = oracle64.dynamics(1, w.mp, w.X[b, k], w.U[b, k], w.params[b, k])
            At, Bt = s.debug_tile(b, k, 0), s.debug_tile(b, k, 1)
            assert np.abs(At[:12, :12] - A).max() < 2e-5 * max(1.0, np.abs(A).max())
            assert np.abs(Bt[:12, :12] - Bj).max() < 2e-5 * max(1.0, np.abs(Bj).max())
            assert np.abs(At[:12, 12] - (xn - w.X[b, k + 1])).max() < 1e-5
            assert At[12, 12] == 1.0 and np.all(At[12, :12] == 0) and np.all(At[13:, :] == 0)
            assert np.all(Bt[12:, :] == 0) and np.all(Bt[:, 12:] == 0)


# ------------------------------------------------------------------------------- full solves
def test_double_integrator_lq_exact(dev, oracle64):
    """Config 1: an LQ problem, one SQP iteration is the exact optimum."""
    from iterative_learning_nmpc_amd import workloads as wl
    w = wl.double_integrator(B=16, N=20, seed=0)
    s = _solver(w, 16, dev, n_ipm=0)
    X, U, st, stats = _gpu_solve(s, w)
    Xo, Uo, sto, _ = _oracle_solve(oracle64, w, n_ipm=0)
    assert rel(X, Xo) < 1e-5 and rel(U, Uo) < 1e-5
    assert (st == 2).all() and (sto == 2).all()
    # a second iteration does not move the solution: converged -> status 0 with a tolerance
    s.set_max_iter(3); s.set_nlp_tol(1e-3)
    X2, U2, st2, stats2 = _gpu_solve(s, w)
    assert (st2 == 0).all() and (stats2[:, 3] == 2).all()
    assert rel(X2, Xo) < 1e-5


def test_double_integrator_box_constraints(dev, oracle64, oracle32):
    from iterative_learning_nmpc_amd import workloads as wl
    w = wl.double_integrator(B=16, N=20, seed=1, umax=1.0)
    s = _solver(w, 16, dev, n_ipm=8)
    X, U, st, _ = _gpu_solve(s, w)
    Xo, Uo, _, _ = _oracle_solve(oracle64, w, n_ipm=8)
    assert np.abs(U).max() <= 1.0 + 1e-4          # box respected
    assert np.abs(Uo).max() > 0.9                   # and it is active
    assert rel(X, Xo) < 1e-5 and rel(U, Uo) < 1e-5, (rel(X, Xo), rel(U, Uo))      # measured 2.9e-7 / 4.4e-7 (tools/parity_floor.py)


def _within_tolerance(e, floor):
    """The stated bar, 1e-5 relative L2 against the fp64 oracle.  The fp32 oracle is the same algorithm in float with
    the CPU's summation order -- the kernels contract in the order of the matrix instruction, so neither is bit-equal
    to the other; where the CPU's own fp32 error approaches the bar (measured: 7.8e-6 after three SQP iterations,
    1.65e-5 with a binding friction pyramid, tools/parity_floor.py) the device may sit at 1.5 x that floor."""
    return e < 1e-5 or e < 1.5 * floor


# (6, 15) is the reference's first-solve policy (6 IPM x 15 SQP, mpc.py:464-473)
@pytest.mark.parametrize("n_ipm,sqp", [(0, 1), (6, 1), (6, 3), (0, 15), (6, 15)])
def test_centroidal_solve_parity(dev, oracle64, oracle32, n_ipm, sqp):
    """Config 2 shapes (nx=nu=12, N=50) at a batch the oracle finishes in seconds."""
    from iterative_learning_nmpc_amd import workloads as wl
    B = 64
    w = wl.centroidal_trot(B=B, N=50, seed=0)
    s = _solver(w, B, dev, n_ipm=n_ipm, max_sqp_iter=sqp)
    X, U, st, stats = _gpu_solve(s, w)
    X64, U64, st64, stats64 = _oracle_solve(oracle64, w, n_ipm=n_ipm, max_sqp_iter=sqp)
    X32, U32, _, _ = _oracle_solve(oracle32, w, n_ipm=n_ipm, max_sqp_iter=sqp)
    floor = max(rel(X32, X64), rel(U32, U64))         # what fp32 itself costs on the CPU
    eX, eU = rel(X, X64), rel(U, U64)
    print(f"n_ipm={n_ipm} sqp={sqp}: gpu-vs-f64 X {eX:.2e} U {eU:.2e}; gpu-vs-f32 X {rel(X, X32):.2e} "
          f"U {rel(U, U32):.2e}; f32-vs-f64 floor {floor:.2e}")
    assert np.array_equal(st, st64)
    # measured gpu-vs-fp64 (X / U): 4.2e-6 / 9e-7, 4.1e-6 / 1.0e-6, 9.6e-6 / 1.2e-6 (fp32 oracle itself 7.8e-6), 9.7e-7 / 2.2e-7,
    # 1.0e-6 / 2.3e-7: fifteen iterations contract the rounding differences, they do not amplify them
    assert _within_tolerance(eX, rel(X32, X64)) and eU < 1e-5, (eX, eU, floor)
    assert rel(X, X32) < 1e-5 and rel(U, U32) < 1e-5                 # and against the fp32 oracle (measured <= 6.2e-6)
    assert np.allclose(stats[:, 0], stats64[:, 0], rtol=1e-4)       # cost at linearisation
    assert np.array_equal(stats[:, 3], stats64[:, 3])               # iteration count


def test_centroidal_active_friction(dev, oracle64, oracle32):
    """Low friction makes the pyramid active: constraints hold, parity at the fp32 floor of an
    ill-conditioned barrier system (bound documented in DESIGN.md 3.3)."""
    from iterative_learning_nmpc_amd import workloads as wl
    B = 32
    w = wl.centroidal_trot(B=B, N=50, seed=1)
    w.mp[6] = 0.3
    s = _solver(w, B, dev, n_ipm=6)
    X, U, st, _ = _gpu_solve(s, w)
    X64, U64, _, _ = _oracle_solve(oracle64, w, n_ipm=6)
    f = U.reshape(B, 50, 4, 3)
    c = np.moveaxis(w.params[:, :50, :4], 2, 2)
    viol = np.maximum(np.abs(f[..., :2]).max(-1) - 0.3 * f[..., 2], 0) * c
    assert viol.max() < 1e-3
    X32, U32, _, _ = _oracle_solve(oracle32, w, n_ipm=6)
    # measured 1.84e-5 / 4.6e-6 with the fp32 oracle itself at 1.65e-5 / 4.6e-6: the stiff barrier system, not the kernel
    assert _within_tolerance(rel(X, X64), rel(X32, X64)) and rel(U, U64) < 1e-5, (rel(X, X64), rel(U, U64), rel(X32, X64))


def test_centroidal_line_search_and_status(dev, oracle64):
    from iterative_learning_nmpc_amd import workloads as wl
    B = 16
    w = wl.centroidal_trot(B=B, N=50, seed=2, warm="zero")
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=2, line_search=1)
    X, U, st, stats = _gpu_solve(s, w)
    X64, U64, st64, stats64 = _oracle_solve(oracle64, w, n_ipm=6, max_sqp_iter=2, line_search=1)
    # every problem of the batch backtracks to the same step length as the oracle (a merit value within rounding of the
    # acceptance threshold could flip one; none does on this seed, and the assertion is on the whole batch)
    assert np.array_equal(stats[:, 2], stats64[:, 2])
    assert rel(X, X64) < 1e-5 and rel(U, U64) < 1e-5, (rel(X, X64), rel(U, U64))       # measured 8.9e-7 / 2.4e-7
    # NaN input -> status 1 for that problem only
    w2 = wl.centroidal_trot(B=4, N=50, seed=5)
    w2.x0[2, 0] = np.nan
    s2 = _solver(w2, 4, dev)
    _, _, st2, _ = _gpu_solve(s2, w2)
    assert st2.tolist() == [2, 2, 1, 2]


def test_warm_start_shift(dev, oracle32):
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    rng = np.random.default_rng(0)
    B, N = 5, 50
    s = BatchedNmpcSolver(1, N, B, dev)
    X, U = rng.normal(size=(B, N + 1, 12)).astype(np.float32), rng.normal(size=(B, N, 12)).astype(np.float32)
    for shift in (0, 1, 2, 7, 50, 60):
        Xd, Ud = s.to_device(X), s.to_device(U)
        s.warm_start_solver(Xd, Ud, shift)
        Xo, Uo = oracle32.shift_warm_start(X, U, shift)
        assert np.array_equal(Xd.cpu().numpy(), Xo) and np.array_equal(Ud.cpu().numpy(), Uo)


# ------------------------------------------------------------------------------- tracking error
@pytest.mark.parametrize("B,T,ns", [(5, 60, 44), (3, 1, 2), (7, 129, 13), (64, 500, 44)])
def test_tracking_error_bit_exact(dev, oracle32, B, T, ns):
    from iterative_learning_nmpc_amd.solver import tracking_error
    rng = np.random.default_rng(B + T)
    Snom = rng.normal(0, 1, (T, ns)).astype(np.float32)
    S = (Snom[None] + rng.normal(0, 0.62, (B, T, ns))).astype(np.float32)
    S[:, :, 0] += 100.0
    err, wgt = tracking_error(torch.from_numpy(S).to(dev), torch.from_numpy(Snom).to(dev))
    ref = oracle32.tracking_error(S, Snom)
    assert np.array_equal(err.cpu().numpy(), ref)           # fp32, same operation order: bit-exact
    assert np.array_equal(wgt.cpu().numpy(), np.where(ref > 4.0, 5.0, 1.0).astype(np.float32))


def test_tracking_error_golden_ood_selection(dev, golden_dir):
    """The OOD samples the reference selects (tests/golden/tracking_error.npz) are the ones whose
    GPU tracking error exceeds the threshold."""
    import os
    from iterative_learning_nmpc_amd.solver import tracking_error
    g = np.load(os.path.join(golden_dir, "tracking_error.npz"))
    S, Snom = g["s_pert"].astype(np.float32), g["s_nom"].astype(np.float32)
    err, wgt = tracking_error(torch.from_numpy(S).to(dev), torch.from_numpy(Snom).to(dev),
                              threshold=float(g["threshold"]))
    matched = np.isclose(g["t_pert"], g["t_nom"][None], atol=1e-9)   # rows with a nominal counterpart
    sel = (wgt.cpu().numpy() == 5.0) & matched
    margin = np.abs(err.cpu().numpy() - 4.0) > 1e-4                    # ignore fp32 ties at the threshold
    assert np.array_equal(sel[margin], g["ood_selected"][margin])


# ------------------------------------------------------------------------------- controller / rollouts
def test_open_loop_rollouts_match_oracle_driven_loop(dev, oracle64):
    """`BatchedLocomotionMPC.open_loop` (mirror of mpc.py:416-462) against the same host loop
    driven by the CPU oracle: 8 replans (first one 15 SQP iterations), nominal + pushed rollouts."""
    from iterative_learning_nmpc_amd.mpc import BatchedLocomotionMPC, N_SQP_FIRST
    from iterative_learning_nmpc_amd.solver import tracking_error
    B, T = 6, 0.32
    rng = np.random.default_rng(7)
    x0 = np.zeros((B, 12)); x0[:, 2] = 0.3
    force = np.zeros((B, 3)); force[1:] = rng.uniform(-1, 1, (B - 1, 3)) * 60.0     # rollout 0 = nominal
    push = dict(start=0.08, duration=0.12, force=force)

    mpc = BatchedLocomotionMPC(B, n_nodes=50, device=dev)
    mpc.set_command(np.array([0.2, 0.0, 0.0]), 0.0)
    S, t = mpc.open_loop(x0, T, push)
    torch.cuda.synchronize()
    assert S.shape == (B, 8, 19) and len(t) == 8 and mpc.current_opt_node == 16
    assert (mpc.status.cpu().numpy() != 1).all() and np.isfinite(S.cpu().numpy()).all()

    # the same loop with the oracle as the solver
    ref = BatchedLocomotionMPC(B, n_nodes=50, device=dev)
    ref.set_command(np.array([0.2, 0.0, 0.0]), 0.0)
    x, X, U, rec = x0.copy(), None, None, []
    for i in range(8):
        yref, yref_e, params = ref.build_problem(x)
        first = X is None
        if first:
            X, U = np.repeat(x[:, None, :], 51, axis=1), yref[:, :, 12:].copy()
        else:
            X, U = oracle64.shift_warm_start(X, U, ref.nodes_per_replan)
        opt = oracle64.opt(max_sqp_iter=N_SQP_FIRST if first else 1, n_ipm=6, yref_per_stage=1,
                           nlp_tol=(0.01 if first else 0.1), reg=ref.config_cost.reg_eps, reg_e=ref.config_cost.reg_eps_e)
        W = np.concatenate([ref.config_cost.W_base, ref.config_cost.W_cnt_f_reg.ravel()])
        X, U, _, _ = oracle64.solve_batch(1, 50, ref.mp, opt, W, ref.config_cost.W_e_base, x, yref, yref_e, params, X, U)
        rec.append(ref.record_state(x, i * 0.04))
        x = X[:, ref.nodes_per_replan].copy()
        if push["start"] <= i * 0.04 < push["start"] + push["duration"]:
            x[:, 6:9] += force * 0.04 / ref.mp[1]
        ref.sim_step += ref.replanning_steps
        ref.current_opt_node += ref.nodes_per_replan
        ref.increment_base_ref_position(ref.replanning_steps)
    S_ref = np.stack(rec, axis=1)
    assert rel(S.cpu().numpy(), S_ref) < 1e-4, rel(S.cpu().numpy(), S_ref)

    # tracking error of every rollout against the nominal one (rollout 0)
    err, w = tracking_error(S, S[0].contiguous())
    err = err.cpu().numpy()
    assert np.all(err[0] == 0) and err[1:, -1].min() > 1e-3
    assert np.allclose(err, oracle64.tracking_error(S.cpu().numpy(), S[0].cpu().numpy()), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------- edge cases
@pytest.mark.parametrize("model,B,N", [(1, 1, 50), (1, 3, 7), (1, 5, 70), (1, 2, 64), (1, 2, 65), (0, 2, 1), (0, 7, 100)])
def test_odd_batches_and_horizons(dev, oracle64, model, B, N):
    """Ragged sizes: B = 1, odd B, N below / above the 64-lane stage loop, N = 1."""
    from iterative_learning_nmpc_amd import workloads as wl
    w = wl.centroidal_trot(B=B, N=N, seed=11) if model == 1 else wl.double_integrator(B=B, N=N, seed=11, umax=1.5)
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=2)
    X, U, st, stats = _gpu_solve(s, w)
    Xo, Uo, sto, statso = _oracle_solve(oracle64, w, n_ipm=6, max_sqp_iter=2)
    assert np.array_equal(st, sto)
    assert rel(X, Xo) < 1e-5 and rel(U, Uo) < 1e-5, (rel(X, Xo), rel(U, Uo))           # measured <= 2.3e-6


def test_per_problem_early_exit(dev, oracle64):
    """nlp_tol > 0: every problem stops at its own iteration; status 0 and the iteration count match
    the oracle, and finished problems are left untouched by later launches."""
    from iterative_learning_nmpc_amd import workloads as wl
    B = 24
    w = wl.centroidal_trot(B=B, N=50, seed=8)
    w.X[B // 2:] += np.random.default_rng(1).normal(0, 0.05, w.X[B // 2:].shape)   # harder half
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=10, nlp_tol=0.5)
    X, U, st, stats = _gpu_solve(s, w)
    Xo, Uo, sto, statso = _oracle_solve(oracle64, w, n_ipm=6, max_sqp_iter=10, nlp_tol=0.5)
    assert len(set(statso[:, 3].tolist())) > 1, "test needs problems stopping at different iterations"
    # every problem stops at the oracle's iteration (a step norm within rounding of the tolerance could flip one: the
    # closest on this seed is 3 % away from it)
    assert np.array_equal(stats[:, 3], statso[:, 3])
    assert np.array_equal(st, sto) and (st == 0).all()
    assert rel(X, Xo) < 1e-5 and rel(U, Uo) < 1e-5                                      # measured 9e-7 / 1.9e-7


def test_handle_reuse_after_riccati_and_argument_errors(dev, oracle64):
    from iterative_learning_nmpc_amd import _lib, workloads as wl
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    B = 4
    w = wl.centroidal_trot(B=B, N=50, seed=4)
    s = _solver(w, B, dev)
    X1, U1, _, _ = _gpu_solve(s, w)
    # a dense-LQ call on the same handle dirties the tile workspace; the next solve must not care
    rng = np.random.default_rng(0)
    lq = [np.tile(np.eye(9), (B, 51, 1, 1)), np.tile(np.eye(5), (B, 50, 1, 1)), rng.normal(size=(B, 51, 9)),
          rng.normal(size=(B, 50, 5)), np.tile(np.eye(9), (B, 50, 1, 1)), rng.normal(size=(B, 50, 9, 5)),
          rng.normal(size=(B, 50, 9)), rng.normal(size=(B, 9))]
    s.riccati(*[s.to_device(a) for a in lq])
    X2, U2, _, _ = _gpu_solve(s, w)
    assert np.array_equal(X1, X2) and np.array_equal(U1, U2)          # and the solve is deterministic
    # argument errors surface as exceptions with the library's message
    big = wl.centroidal_trot(B=B + 1, N=50, seed=4)
    with pytest.raises(_lib.NmpcError, match="B_max"):
        _gpu_solve(s, big)
    with pytest.raises(ValueError):
        s.solve(s.to_device(w.x0[:, :5]), s.to_device(w.yref), s.to_device(w.yref_e), s.to_device(w.params),
                s.to_device(w.X), s.to_device(w.U))
    fresh = BatchedNmpcSolver(1, 50, B, dev)
    with pytest.raises(_lib.NmpcError, match="not set"):
        _gpu_solve(fresh, w)
    with pytest.raises(_lib.NmpcError):
        fresh.set_model_params(wl.model_params(dt=0.0))    # dt = 0
    # an empty batch is a no-op
    e = {k: s.to_device(getattr(w, k)[:0]) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
    s.solve(e["x0"], e["yref"], e["yref_e"], e["params"], e["X"], e["U"])


def test_full_size_properties(dev):
    """BASELINE config 2 at full size (B = 1024): size-independent properties instead of the oracle --
    dynamics residual after convergence, friction pyramid, zero swing forces, x_0 = x0, permutation
    equivariance of the batch."""
    from iterative_learning_nmpc_amd import workloads as wl
    B = 1024
    w = wl.centroidal_trot(B=B, N=50, seed=21)
    s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=8)
    X, U, st, stats = _gpu_solve(s, w)
    assert (st != 1).all() and (st != 4).all() and np.isfinite(X).all()
    assert np.abs(X[:, 0] - w.x0).max() < 1e-5
    c = w.params[:, :50, :4]
    f = U.reshape(B, 50, 4, 3)
    assert np.abs(f * (1 - c)[..., None]).max() < 1e-3                     # swing feet carry nothing
    assert (np.abs(f[..., :2]).max(-1) - 0.8 * f[..., 2] <= 1e-2)[c > 0.5].all()   # pyramid, mu = 0.8
    # full-step Gauss-Newton converges for the bulk of the batch (hard initial rates oscillate at the
    # ~30 N level on the oracle too; globalisation is opt-in)
    assert np.mean(stats[:, 1] < 1.0) > 0.9
    # vertical momentum balance of the declared model along the solution
    vz_next = X[:, :-1, 8] + w.mp[0] * ((f[..., 2] * c).sum(-1) / w.mp[1] + w.mp[5])
    assert np.abs(vz_next - X[:, 1:, 8]).max() < 5e-3
    perm = np.random.default_rng(0).permutation(B)
    w2 = wl.centroidal_trot(B=B, N=50, seed=21)
    for k in ("x0", "yref", "yref_e", "params", "X", "U"):
        setattr(w2, k, getattr(w2, k)[perm])
    X2, U2, _, _ = _gpu_solve(s, w2)
    assert np.array_equal(X2, X[perm]) and np.array_equal(U2, U[perm])


def test_device_rollout_matches_host_driven_rollout(dev):
    """nmpc_rollout_batch (everything on the device, one host call) against open_loop (host loop that
    builds references with numpy and calls the same solver): same recorded states, same final state,
    same integrated reference, incl. a yaw-rate command and a push."""
    from iterative_learning_nmpc_amd.mpc import BatchedLocomotionMPC
    B, T = 9, 0.40
    rng = np.random.default_rng(3)
    x0 = np.zeros((B, 12)); x0[:, 2] = 0.3
    x0[:, :2] = rng.normal(0, 0.03, (B, 2)); x0[:, 3] = rng.normal(0, 0.2, B)
    force = rng.uniform(-1, 1, (B, 3)) * 50.0; force[0] = 0
    push = dict(start=0.12, duration=0.08, force=force)
    out = []
    for mode in ("host", "device"):
        mpc = BatchedLocomotionMPC(B, n_nodes=50, device=dev)
        mpc.set_command(np.array([0.25, 0.05, 0.0]), 0.3)
        mpc.base_ref_vel_tracking[:, :2] = x0[:, :2]
        mpc.base_ref_vel_tracking[:, 3] = x0[:, 3]
        roll = mpc.open_loop if mode == "host" else mpc.open_loop_device
        S, t = roll(x0, T, push)
        S2, t2 = roll(mpc.x_final.double().cpu().numpy() if mode == "device" else mpc._x_last, 0.12, None)   # continue warm
        torch.cuda.synchronize()
        out.append((S.cpu().numpy(), S2.cpu().numpy(), mpc.base_ref_vel_tracking.copy(), mpc.current_opt_node,
                    mpc.X.cpu().numpy()))
    (Sh, Sh2, refh, nodeh, Xh), (Sd, Sd2, refd, noded, Xd) = out
    assert Sh.shape == Sd.shape == (B, 10, 19) and Sd2.shape == (B, 3, 19) and nodeh == noded == 26
    assert np.allclose(refh, refd, rtol=0, atol=1e-12)
    assert rel(Sd, Sh) < 2e-5 and rel(Sd2, Sh2) < 2e-5 and rel(Xd, Xh) < 2e-5, (rel(Sd, Sh), rel(Sd2, Sh2), rel(Xd, Xh))
    assert np.abs(Sd[1:, -1] - Sd[0, -1]).max() > 1e-3          # the pushes did something


# ----------------------------------------------------------------------------- kernel variants
@pytest.mark.gpu
@pytest.mark.parametrize("model,N,opts", [(1, 50, dict(n_ipm=6, max_sqp_iter=2)), (1, 70, dict(n_ipm=6)),
                                          (0, 20, dict(n_ipm=6)), (1, 50, dict(n_ipm=6, line_search=1)),
                                          (1, 64, dict(n_ipm=6)), (1, 63, dict(n_ipm=6)), (1, 130, dict(n_ipm=3)),
                                          (0, 70, dict(n_ipm=6))])
def test_lean_variant_is_bit_identical_to_resident(dev, monkeypatch, model, N, opts):
    """The lean-LDS variant of the QP kernel (stage arrays in the workspace, stage-major rows, the blend of the step folded into
    the interior-point phase) runs the same arithmetic in the same order as the resident one: trajectories, status and
    stats agree bit for bit -- at one stage per lane (N <= 64, node 64 without a lane of its own) and beyond."""
    from iterative_learning_nmpc_amd import workloads as wl
    B = 6
    w = wl.centroidal_trot(B=B, N=N, seed=13) if model == 1 else wl.double_integrator(B=B, N=N, seed=13, umax=1.2)
    out = {}
    for variant in ("resident", "lean"):
        monkeypatch.setenv("NMPC_QP_VARIANT", variant)       # read by nmpc_create
        out[variant] = _gpu_solve(_solver(w, B, dev, **opts), w)
    for a, b in zip(out["resident"], out["lean"]):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["shift", "early_exit", "nan_with_shift", "nan_without_shift"])
def test_lean_variant_equals_resident_on_the_step_and_write_back_paths(dev, monkeypatch, case):
    """The two variants end a solve through different code (the lean one takes a node's rows through registers: norm, step,
    store to the caller's X, U): same trajectories, status and stats with a folded warm-start shift, with problems that
    stop early at nlp_tol, and with a NaN problem -- which leaves the (shifted) previous solution in X, U in both."""
    from iterative_learning_nmpc_amd import workloads as wl
    B, N = 7, 50
    w = wl.centroidal_trot(B=B, N=N, seed=23)
    if case.startswith("nan"):
        w.x0 = w.x0.copy(); w.x0[3, 1] = np.nan
    shift = 0 if case == "nan_without_shift" else 2
    out = {}
    for variant in ("resident", "lean"):
        monkeypatch.setenv("NMPC_QP_VARIANT", variant)
        s = _solver(w, B, dev, n_ipm=6, max_sqp_iter=4 if case == "early_exit" else 1, nlp_tol=0.2 if case == "early_exit" else 0.0)
        t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
        X, U, st, stats = s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], t["X"], t["U"], shift=shift)
        torch.cuda.synchronize()
        out[variant] = (X.cpu().numpy(), U.cpu().numpy(), st.cpu().numpy(), stats.cpu().numpy())
    for a, b in zip(out["resident"], out["lean"]):
        assert np.array_equal(a, b, equal_nan=True)
    st = out["lean"][2]
    if case.startswith("nan"):
        assert st[3] == 1 and (np.delete(st, 3) == 2).all()
        if shift == 0:                                            # untouched
            assert np.array_equal(out["lean"][0][3], w.X[3].astype(np.float32))
        else:                                                     # the shifted previous solution
            assert np.array_equal(out["lean"][0][3, 1:N - shift + 1], w.X[3, 1 + shift:N + 1].astype(np.float32))
            assert (out["lean"][1][3, N - shift:] == 0).all()
    if case == "early_exit":
        assert (st == 0).any() or (st == 2).all()                 # converged problems report NMPC_STATUS_OK


@pytest.mark.gpu
def test_batch_beyond_one_wave_per_simd_runs_in_rounds(dev, oracle32, monkeypatch):
    """B > 4 waves/CU x CUs: the library's own choice (the two-waves-per-SIMD variant beyond one problem per SIMD) equals
    the resident kernel forced onto the same batch, which runs it in rounds, bit for bit -- and both match the oracle on
    the tail of the batch."""
    from iterative_learning_nmpc_amd import workloads as wl
    monkeypatch.delenv("NMPC_QP_VARIANT", raising=False)
    B = 1100
    w = wl.centroidal_trot(B=B, N=50, seed=17)
    X, U, st, _ = _gpu_solve(_solver(w, B, dev), w)
    monkeypatch.setenv("NMPC_QP_VARIANT", "resident")
    Xr, Ur, str_, _ = _gpu_solve(_solver(w, B, dev), w)
    assert np.array_equal(X, Xr) and np.array_equal(U, Ur) and np.array_equal(st, str_)
    sel = slice(B - 24, B)                                    # the tail of the batch: the second round
    ws = wl.centroidal_trot(B=B, N=50, seed=17)
    Xo, Uo, sto, _ = oracle32.solve_batch(ws.model_id, ws.N, ws.mp, oracle32.opt(yref_per_stage=1, reg=ws.meta["reg"], reg_e=ws.meta["reg_e"]),
                                          ws.W, ws.W_e, ws.x0[sel], ws.yref[sel], ws.yref_e[sel], ws.params[sel], ws.X[sel], ws.U[sel])
    assert rel(X[sel], Xo) < 2e-5 and rel(U[sel], Uo) < 2e-5
    assert np.array_equal(st[sel], sto)


@pytest.mark.gpu
def test_mixed_precision_barrier_product(dev, oracle64):
    """BASELINE configs[4]: precision = 1 computes the interior-point barrier product G'DG | G'v with
    ONE bf16 MFMA (inputs rounded to 8 bits of mantissa) and everything else in fp32.  The measured
    outcome is negative and the test pins it: the barrier terms dominate Huu near active constraints,
    their 0.4 % rounding perturbs every Newton direction, and after six interior-point iterations the
    trajectories differ from the fp64 oracle by 8e-2 (X) / 3e-2 (U) relative -- four orders above the
    fp32 path, for 2 % more throughput.  Emulating the rounding in fp32 arithmetic gives
    the same deviation, so it is the precision, not the instruction."""
    from iterative_learning_nmpc_amd import workloads as wl
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    B = 16
    w = wl.centroidal_trot(B=B, N=50, seed=2)
    s = BatchedNmpcSolver(w.model_id, w.N, B, dev, precision=1)
    s.set_model_params(w.mp)
    s.set_cost_weights(w.W, w.W_e, w.meta["reg"], w.meta["reg_e"])
    X, U, st, _ = _gpu_solve(s, w)
    Xo, Uo, sto, _ = _oracle_solve(oracle64, w)
    eX, eU = rel(X, Xo), rel(U, Uo)
    print(f"mixed precision: rel-L2 X {eX:.2e} U {eU:.2e}")
    assert np.array_equal(st, sto)
    # the documented deviation (8e-2 / 3e-2 on this workload), within a factor of two either way
    assert 4e-2 < eX < 1.6e-1 and 1.5e-2 < eU < 6e-2, (eX, eU)


@pytest.mark.gpu
@pytest.mark.parametrize("shift,sqp", [(1, 1), (3, 2), (50, 1), (7, 1)])
def test_fused_shift_solve_equals_shift_then_solve(dev, shift, sqp):
    """nmpc_shift_solve_batch reads the previous solution through the shift's index map; the result
    is bit-identical to nmpc_shift_warm_start followed by nmpc_solve_batch."""
    from iterative_learning_nmpc_amd import workloads as wl
    B = 5
    w = wl.centroidal_trot(B=B, N=50, seed=31)
    rng = np.random.default_rng(5)
    w.X = (w.X + 0.01 * rng.standard_normal(w.X.shape)).astype(np.float32)      # a previous solution with structure
    w.U = (w.U + 0.5 * rng.standard_normal(w.U.shape)).astype(np.float32)
    s = _solver(w, B, dev, max_sqp_iter=sqp)
    t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params")}
    Xa, Ua = s.to_device(w.X), s.to_device(w.U)
    s.warm_start_solver(Xa, Ua, shift)
    s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], Xa, Ua)
    Xb, Ub = s.to_device(w.X), s.to_device(w.U)
    _, _, stb, _ = s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], Xb, Ub, shift=shift)
    torch.cuda.synchronize()
    assert torch.equal(Xa, Xb) and torch.equal(Ua, Ub)


@pytest.mark.gpu
def test_all_contact_patterns_kernel_matches_default_and_oracle(dev, oracle64):
    """Random contact patterns per stage (every subset of the four feet occurs): the default kernel
    (static bodies for the trot patterns, run-time-mask fallback for the rest) and the kernel with a
    static body per pattern (nmpc_set_contact_patterns) solve the same problems; both match the oracle."""
    from iterative_learning_nmpc_amd import workloads as wl
    B, N = 24, 50
    w = wl.centroidal_trot(B=B, N=N, seed=23)
    rng = np.random.default_rng(7)
    flags = rng.integers(0, 2, size=(B, N + 1, 4)).astype(np.float32)
    flags[:, :, :][flags.sum(-1) == 0] = np.array([1, 0, 0, 1], np.float32)        # keep some support most of the time
    flags[:, ::7] = 0.0                                                              # ... and some flight stages
    w.params[:, :, 0:4] = flags
    n_st = np.maximum(flags[:, :N].sum(-1, keepdims=True), 1.0)
    fz = (-w.mp[5] * w.mp[1]) / n_st                                                 # weight shared by the stance feet
    for i in range(4):
        w.yref[:, :, 12 + 3 * i: 14 + 3 * i] = 0.0
        w.yref[:, :, 14 + 3 * i] = (fz[..., 0] * flags[:, :N, i]).astype(np.float32)
    w.U[:] = w.yref[:, :, 12:]
    assert len(np.unique((flags[:, :N] * np.array([1, 2, 4, 8])).sum(-1))) == 16
    out = {}
    for allp in (False, True):
        s = _solver(w, B, dev)
        assert s.set_contact_patterns(all_patterns=allp) == allp
        out[allp] = _gpu_solve(s, w)
    Xo, Uo, sto, _ = _oracle_solve(oracle64, w)
    for allp in (False, True):
        X, U, st, _ = out[allp]
        assert np.array_equal(st, sto)
        assert rel(X, Xo) < 3e-5 and rel(U, Uo) < 3e-5, (allp, rel(X, Xo), rel(U, Uo))
    assert rel(out[True][0], out[False][0]) < 3e-5


def test_contact_pattern_choice_from_gait_table():
    """Host side: a trot keeps the default kernel, a pace or a crawl asks for all patterns."""
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    common = BatchedNmpcSolver.COMMON_CONTACT_PATTERNS
    pat = lambda g: set((np.asarray(g)[0] | (np.asarray(g)[1] << 1) | (np.asarray(g)[2] << 2) | (np.asarray(g)[3] << 3)).tolist())
    trot = np.array([[1, 1, 0, 0], [0, 0, 1, 1], [0, 0, 1, 1], [1, 1, 0, 0]])
    pace = np.array([[1, 1, 0, 0], [0, 0, 1, 1], [1, 1, 0, 0], [0, 0, 1, 1]])
    assert pat(trot) <= common and not pat(pace) <= common


@pytest.mark.gpu
def test_device_rollout_is_the_same_under_every_kernel_variant(dev, monkeypatch):
    """Device-resident rollouts (cold start, warm replans with the folded shift, push) through the lean
    LDS layout and through the all-contact-patterns kernel reproduce the default kernel's states:
    bit for bit for the layout (same arithmetic), to rounding for the pattern set (same stage bodies
    for a trot, other code only around them)."""
    from iterative_learning_nmpc_amd.mpc import BatchedLocomotionMPC
    B, T = 5, 0.32
    rng = np.random.default_rng(11)
    x0 = np.zeros((B, 12)); x0[:, 2] = 0.3
    x0[:, :2] = rng.normal(0, 0.03, (B, 2))
    push = dict(start=0.08, duration=0.08, force=rng.uniform(-1, 1, (B, 3)) * 40.0)

    def roll(variant, all_patterns):
        if variant: monkeypatch.setenv("NMPC_QP_VARIANT", variant)
        else: monkeypatch.delenv("NMPC_QP_VARIANT", raising=False)
        mpc = BatchedLocomotionMPC(B, n_nodes=50, device=dev)
        assert mpc.solver.set_contact_patterns(gait_sequence=mpc.contact_planner.gait_sequence) is False   # a trot
        mpc.solver.set_contact_patterns(all_patterns=all_patterns)
        mpc.set_command(np.array([0.3, 0.0, 0.0]), 0.0)
        S, _ = mpc.open_loop_device(x0, T, push)
        torch.cuda.synchronize()
        assert int((mpc.failed & 1).sum().item()) == 0          # bit 0: solver failure (NMPC_ROLLOUT_FLAG_SOLVER)
        return S.cpu().numpy()

    ref = roll(None, False)
    assert np.array_equal(roll("lean", False), ref)
    assert np.array_equal(roll("resident", False), ref)
    assert rel(roll(None, True), ref) < 2e-5


@pytest.mark.gpu
def test_option_argument_errors(dev):
    from iterative_learning_nmpc_amd import workloads as wl, _lib
    w = wl.centroidal_trot(B=2, N=50, seed=0)
    s = _solver(w, 2, dev)
    t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
    with pytest.raises(_lib.NmpcError, match="negative shift"):
        s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], t["X"], t["U"], shift=-1)
    s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], t["X"], t["U"], shift=10 ** 6)     # clamps to N: U becomes the zero tail
    torch.cuda.synchronize()
    assert s.lib.nmpc_set_contact_patterns(None, 1) != 0
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    with pytest.raises(_lib.NmpcError):
        BatchedNmpcSolver(1, 50, 2, dev, precision=7)
