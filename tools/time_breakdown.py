#!/usr/bin/env python3
"""Diagnostic: cost of one interior-point iteration and of the fixed part of a solve call, from
timing the production library at different iteration counts (no stamps, nothing perturbed)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iterative_learning_nmpc_amd import workloads as wl
from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
res = {}
for N in (50, 25):
    w = wl.centroidal_trot(B=B, N=N, seed=0)
    s = BatchedNmpcSolver(w.model_id, w.N, B, "cuda:0")
    s.set_model_params(w.mp)
    s.set_cost_weights(w.W, w.W_e, w.meta["reg"], w.meta["reg_e"])
    t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
    for n_ipm in (6, 3, 1):
        s.set_max_qp_iter(n_ipm)
        X, U = t["X"].clone(), t["U"].clone()
        for _ in range(5):
            s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], X, U)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], X, U)
        e1.record()
        torch.cuda.synchronize()
        res[(N, n_ipm)] = e0.elapsed_time(e1) / 40
        print(f"N={N} n_ipm={n_ipm}: {res[(N, n_ipm)]:.4f} ms per solve call")
for N in (50, 25):
    per = (res[(N, 6)] - res[(N, 3)]) / 3
    print(f"N={N}: one IPM iteration {per * 1e3:.1f} us = {per * 1e3 / N:.3f} us per stage; fixed part {(res[(N, 6)] - 6 * per) * 1e3:.1f} us")
