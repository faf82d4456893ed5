#!/usr/bin/env python3
"""Diagnostic: where a solve spends its cycles.  Builds a stamped variant of the library
(-DNMPC_STAMPS) into gpurun_out/, runs one batch and prints the share of each phase.
Read SHARES, not lengths: the stamps forbid overlaps the production kernel has."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", "libnmpc_hip_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
extra = [x for x in sys.argv[2:] if x.startswith("-D")]
if os.environ.get("NMPC_STAMPS_LIB"):     # a stamped library built beforehand (hipcc cross-compiles off the GPU box)
    out = os.environ["NMPC_STAMPS_LIB"]
else:
  subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-mfma-vgpr-form", "-DNMPC_STAMPS", *extra,
                "-o", out, *[os.path.join(ROOT, "iterative_learning_nmpc_amd", "csrc", f)
                             for f in ("nmpc_api.hip", "nmpc_policy.hip", "nmpc_dataset.hip", "nmpc_torque.hip")]], check=True)
os.environ["NMPC_HIP_LIB"] = out
sys.path.insert(0, ROOT)
import ctypes  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from iterative_learning_nmpc_amd import workloads as wl  # noqa: E402
from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = wl.centroidal_trot(B=B, N=50, seed=0)
s = BatchedNmpcSolver(w.model_id, w.N, B, "cuda:0")
s.set_model_params(w.mp)
s.set_cost_weights(w.W, w.W_e, w.meta["reg"], w.meta["reg_e"])
dbg = torch.zeros(B, 16, device="cuda:0")
s.lib.nmpc_debug_set_buffer(s._h, ctypes.c_void_p(dbg.data_ptr()))
t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
for _ in range(3):
    s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], t["X"], t["U"])
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.float64)
names = ["linearise", "ipm upd+coef", "backward", "forward", "last ipm upd", "step+store"]
tot = d[:, :6].sum(1).mean()
print(f"B={B} {extra}: mean cycles per wave {tot:.0f}")
tw = d[:, :6].sum(1)
print(f"  per-wave total: min {tw.min():.0f}  p50 {np.median(tw):.0f}  p90 {np.percentile(tw, 90):.0f}  max {tw.max():.0f}  (the kernel lasts as long as its slowest wave)")
bw = d[:, 2]
print(f"  per-wave backward: min {bw.min():.0f}  p50 {np.median(bw):.0f}  max {bw.max():.0f}")
for i, n in enumerate(names):
    print(f"  {n:12s} {d[:, i].mean():10.0f} cycles  {100 * d[:, i].mean() / tot:5.1f} %")
seg = ["products PA..Hxx", "LDS -> columns", "LDL' elimination", "columns -> LDS -> acc", "K, P+, Acl", "stores + loop", "cost tiles (+IPM MFMA)"]
n_stage = 6 * 50
print("backward stage segments (cycles per stage; stamps serialise, read shares):")
for i, n in enumerate(seg):
    print(f"  {n:24s} {d[:, 8 + i].mean() / n_stage:8.0f}")
