#!/usr/bin/env python3
"""Diagnostic (not collected by pytest): distribution of the GPU-vs-fp64-oracle trajectory error
over several seeds, for the library named by NMPC_HIP_LIB (default: the in-tree build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from iterative_learning_nmpc_amd import workloads as wl
from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
from oracle.oracle import Oracle

o = Oracle("f64")
o32 = Oracle("f32")
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
B = 128
for seed in range(4):
    w = wl.centroidal_trot(B=B, N=50, seed=seed)
    s = BatchedNmpcSolver(w.model_id, w.N, B, "cuda:0")
    s.set_model_params(w.mp)
    s.set_cost_weights(w.W, w.W_e, w.meta["reg"], w.meta["reg_e"])
    t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
    X, U, st, _ = s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], t["X"], t["U"])
    torch.cuda.synchronize()
    opt = o.opt(yref_per_stage=1, reg=w.meta["reg"], reg_e=w.meta["reg_e"])
    Xo, Uo, _, _ = o.solve_batch(w.model_id, w.N, w.mp, opt, w.W, w.W_e, w.x0, w.yref, w.yref_e, w.params, w.X, w.U)
    X32, U32, _, _ = o32.solve_batch(w.model_id, w.N, w.mp, opt, w.W, w.W_e, w.x0, w.yref, w.yref_e, w.params, w.X, w.U)
    Xg = X.cpu().numpy().astype(np.float64)
    per = np.array([rel(Xg[b], Xo[b]) for b in range(B)])
    per32 = np.array([rel(X32[b].astype(np.float64), Xo[b]) for b in range(B)])
    print(f"seed {seed}: GPU batch {rel(Xg, Xo):.2e} median {np.median(per):.2e} max {per.max():.2e} | "
          f"fp32 CPU batch {rel(X32.astype(np.float64), Xo):.2e} median {np.median(per32):.2e} max {per32.max():.2e}")
