"""Cycle shares of the whole-body QP kernel's segments (diagnostic build -DNMPC_WB_STAMPS):
   hipcc ... -DNMPC_WB_STAMPS -o tools/_dbg/libnmpc_stamps.so ; NMPC_HIP_LIB=tools/_dbg/libnmpc_stamps.so python tools/wb_stamps.py [B]"""
import sys, os; sys.path.insert(0, '.')
import numpy as np, torch
from iterative_learning_nmpc_amd import workloads as wl
from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = wl.wholebody_trot(B=min(B, 256), N=30, seed=0)
rep = lambda a: np.concatenate([a] * (B // a.shape[0]))
s = BatchedNmpcSolver(w.model_id, w.N, B, "cuda:0")
s.set_model_params(w.mp); s.set_cost_weights(w.W, w.W_e, w.meta['reg'], w.meta['reg_e'])
t = {k: s.to_device(rep(getattr(w, k))) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
for _ in range(3):
    s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], t["X"], t["U"], shift=1)
torch.cuda.synchronize()
L = s.debug_wb_layout()
acc = np.stack([s.debug_workspace(b, L['js'], 24) for b in range(0, B, max(1, B // 32))])
m = acc.mean(0)
names = ["P+/K + prefetch", "synthesis", "PA, PB", "Hux, Huu -> LDS", "H -> LDS", "LDL", "H', W -> tiles", "Y = W Hux", "(last stage tail)",
         "forward sweeps", "IPM updates", "step + write-back", "prologue Q~ = Js'Js"]
tot = m[:13].sum()
print(f"B={B}: {tot:.0f} cycles per wave and solve call; per backward stage {m[:8].sum() / 180:.0f}")
for n, v in zip(names, m):
    print(f"  {n:22s} {v:12.0f}  {100 * v / tot:5.1f} %   per stage {v / 180:8.0f}")
print(f"  loop back edge (slot 19) {m[19] / 180:.0f} per stage; interior-point phase of every sweep (slot 20) {m[20]:.0f} per solve = {m[20] / 180:.0f} per stage; sweep set-up (slot 21) {m[21]:.0f} per solve = {m[21] / 180:.0f} per stage")
print(f"  tail of the backward stage (slots 16-18; slot 0 then holds the loop top only): P+ products {m[16] / 180:.0f}, mirror {m[17] / 180:.0f}, K products + stores {m[18] / 180:.0f}, loop top {m[0] / 180:.0f} per stage")
print(f"  forward stage split (separate slots, not in the table's 'forward sweeps'): du = K dx {m[13] / 180:.0f}, dx+ {m[14] / 180:.0f}, hand-over {m[15] / 180:.0f} per stage")
print(f"  interior-point phase, finer (slots 22, 23, 20; per sweep): inputs + step lengths {m[22] / 6:.0f}, reductions + update + stores {m[23] / 6:.0f}, blend + barrier terms {m[20] / 6:.0f}")
