"""Measured parity of the centroidal / double-integrator solve against the fp64 and fp32 oracles, case by case
(the numbers the gates of tests/test_gpu_parity.py are set from)."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch
from oracle.oracle import Oracle
from iterative_learning_nmpc_amd import workloads as wl
from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, float) - b) / np.linalg.norm(b))
o64, o32 = Oracle('f64'), Oracle('f32')
dev = torch.device('cuda:0')
def run(name, w, **opts):
    B = w.B
    s = BatchedNmpcSolver(w.model_id, w.N, B, dev)
    s.set_model_params(w.mp); s.set_cost_weights(w.W, w.W_e, w.meta.get('reg', 1e-6), w.meta.get('reg_e', 1e-5))
    s.set_max_iter(opts.get('max_sqp_iter', 1)); s.set_max_qp_iter(opts.get('n_ipm', 6)); s.set_nlp_tol(opts.get('nlp_tol', 0.0))
    s.set_line_search(opts.get('line_search', 0))
    t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
    X, U, st, stats = s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], t["X"], t["U"])
    torch.cuda.synchronize()
    X, U, st, stats = X.cpu().numpy(), U.cpu().numpy(), st.cpu().numpy(), stats.cpu().numpy()
    kw = dict(max_sqp_iter=1, n_ipm=6, yref_per_stage=int(w.yref.ndim == 3), reg=w.meta.get('reg', 1e-6), reg_e=w.meta.get('reg_e', 1e-5)); kw.update(opts)
    r = {}
    for nm, o in (('f64', o64), ('f32', o32)):
        r[nm] = o.solve_batch(w.model_id, w.N, w.mp, o.opt(**kw), w.W, w.W_e, w.x0, w.yref, w.yref_e, w.params, w.X, w.U)
    X64, U64, st64, s64 = r['f64']; X32, U32, st32, s32 = r['f32']
    same = (stats[:, 3] == s64[:, 3]) & (stats[:, 2] == s64[:, 2])
    print(f"{name:34s} gpu-f64 X {rel(X, X64):.2e} U {rel(U, U64):.2e} | gpu-f32 X {rel(X, X32):.2e} U {rel(U, U32):.2e} | f32-f64 X {rel(X32, X64):.2e} U {rel(U32, U64):.2e}"
          f" | st eq {np.array_equal(st, st64)} same-path {same.sum()}/{B} f32 same-path {((s32[:, 3] == s64[:, 3]) & (s32[:, 2] == s64[:, 2])).sum()}/{B}"
          + (f" | same-path gpu-f64 X {rel(X[same], X64[same]):.2e}" if not same.all() and same.any() else ""))
run("di box ipm8", wl.double_integrator(B=16, N=20, seed=1, umax=1.0), n_ipm=8)
for n_ipm, sqp in [(0, 1), (6, 1), (6, 3), (0, 15), (6, 15)]:
    run(f"centroidal ipm{n_ipm} sqp{sqp}", wl.centroidal_trot(B=64, N=50, seed=0), n_ipm=n_ipm, max_sqp_iter=sqp)
w = wl.centroidal_trot(B=32, N=50, seed=1); w.mp[6] = 0.3
run("centroidal mu0.3", w, n_ipm=6)
run("centroidal line search zero-warm", wl.centroidal_trot(B=16, N=50, seed=2, warm="zero"), n_ipm=6, max_sqp_iter=2, line_search=1)
for model, B, N in [(1, 1, 50), (1, 3, 7), (1, 5, 70), (1, 2, 64), (1, 2, 65), (0, 2, 1), (0, 7, 100)]:
    w = wl.centroidal_trot(B=B, N=N, seed=11) if model == 1 else wl.double_integrator(B=B, N=N, seed=11, umax=1.5)
    run(f"odd model{model} B{B} N{N}", w, n_ipm=6, max_sqp_iter=2)
w = wl.centroidal_trot(B=24, N=50, seed=8); w.X[12:] += np.random.default_rng(1).normal(0, 0.05, w.X[12:].shape)
run("early exit nlp_tol 0.5", w, n_ipm=6, max_sqp_iter=10, nlp_tol=0.5)
