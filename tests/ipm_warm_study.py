#!/usr/bin/env python3
"""Study, on the CPU oracle only: what a warm-started interior point buys.

    python tests/ipm_warm_study.py

For the centroidal trot workload at mu = 0.8 (the bench) and mu = 0.3 (active friction pyramids) it runs the receding-horizon
sequence  first solve (15 SQP) -> K replans (shift by two nodes, plant = plan, 1 SQP x 6 IPM)  three ways -- cold-started
interior point (shipped), warm across the SQP iterations of a call and across replans, and a reference whose QPs are solved
to convergence (60 IPM iterations, tau_min 1e-6) -- and reports per replan
  * the distance of the 6-iteration step from the converged-QP step taken from the SAME linearisation point,
  * the distance fp32 oracle <-> fp64 oracle of the same solve (the fp32 floor).
Not a test (no assertions): the numbers go to DESIGN.md."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterative_learning_nmpc_amd import workloads as wl  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def run(mu, floors, K=8, B=128, shift=2):
    o64, o32 = Oracle("f64"), Oracle("f32")
    w = wl.centroidal_trot(B=B, N=50, seed=0)
    mp = w.mp.copy(); mp[6] = mu
    if mu < 0.5:
        w.yref = w.yref.copy(); w.yref[:, :, 6] = 0.8                 # ask for speed: the pyramids bind
    base = dict(yref_per_stage=1, reg=w.meta["reg"], reg_e=w.meta["reg_e"])
    args = lambda X, U, x0: (1, 50, mp, None, w.W, w.W_e, x0, w.yref, w.yref_e, w.params, X, U)

    def solve(o, opt, X, U, x0, state=None):
        a = list(args(X, U, x0)); a[3] = o.opt(**opt)
        return o.solve_batch(*a, ipm_state=state)

    out = {}
    for mode in ("cold", "warm"):
        warm = dict(ipm_warm=1, ws_s_floor=floors[0], ws_lam_floor=floors[1]) if mode == "warm" else {}
        st64, st32 = {}, {}
        X, U, _, _ = solve(o64, dict(base, max_sqp_iter=15, nlp_tol=0.01, **warm), w.X, w.U, w.x0, st64)
        X32, U32, _, _ = solve(o32, dict(base, max_sqp_iter=15, nlp_tol=0.01, **warm), w.X, w.U, w.x0, st32)
        x0 = w.x0
        qp_err, fl_err = [], []
        for k in range(K):
            x0 = X[:, shift].copy()
            Xs, Us = o64.shift_warm_start(X, U, shift)
            ws = dict(ws_have=1, ws_shift=shift, **warm) if mode == "warm" else {}
            # converged QP from the same linearisation point (cold, many iterations, tiny barrier floor)
            Xc, Uc, _, _ = solve(o64, dict(base, max_sqp_iter=1, n_ipm=60, tau_min=1e-7), Xs, Us, x0)
            s64 = {k_: v.copy() for k_, v in st64.items()}
            s32 = {k_: v.astype(np.float32) for k_, v in st64.items()}           # the same warm start in both precisions
            Xn, Un, stn, _ = solve(o64, dict(base, max_sqp_iter=1, **ws), Xs, Us, x0, s64)
            X3, U3, _, _ = solve(o32, dict(base, max_sqp_iter=1, **ws), Xs, Us, x0, s32)
            qp_err.append((rel(Xn - Xs, Xc - Xs), rel(Un - Us, Uc - Us)))
            fl_err.append((rel(X3, Xn), rel(U3, Un)))
            X, U, st64 = Xn, Un, s64
        out[mode] = (np.array(qp_err), np.array(fl_err))
    return out


if __name__ == "__main__":
    np.set_printoptions(precision=2, linewidth=200)
    for mu in (0.8, 0.3):
        for floors in ((1e-2, 1e-2), (1e-1, 1e-1), (1e-3, 1e-3)):
            r = run(mu, floors)
            print(f"mu = {mu}, floors s/lam = {floors}")
            for mode in ("cold", "warm"):
                q, f = r[mode]
                print(f"  {mode}: step vs converged QP (X, U) per replan: {q[:, 0]}  {q[:, 1]}")
                print(f"  {mode}: fp32 vs fp64 of the same solve (X, U):   {f[:, 0]}  {f[:, 1]}")
