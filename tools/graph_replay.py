"""Diagnostic: one bench step (warm-start shift + solve) replayed from a captured HIP graph against eager launches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iterative_learning_nmpc_amd import workloads as wl
from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
B=1024
w = wl.centroidal_trot(B=B, N=50, seed=0)
s = BatchedNmpcSolver(w.model_id, w.N, B, "cuda:0")
s.set_model_params(w.mp); s.set_cost_weights(w.W, w.W_e, w.meta["reg"], w.meta["reg_e"])
t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
def step():
    s.warm_start_solver(t["X"], t["U"], 1)
    s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], t["X"], t["U"])
for _ in range(5): step()
torch.cuda.synchronize()
def timeit(f, n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
print("eager ms/step", timeit(step))
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    step()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        step()
torch.cuda.synchronize()
print("graph ms/step", timeit(g.replay))
