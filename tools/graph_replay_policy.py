"""Diagnostic: one policy training step (include/nmpc_policy.h) replayed from a captured HIP graph against eager
launches.  Timing only: the replay repeats the captured Adam bias correction, so the parameters it produces are
not those of the real sequence of steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iterative_learning_nmpc_amd.policy import DevicePolicy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pol = DevicePolicy(47, 12, 3, 512, True, batch_max=B, seed=0)
X = torch.randn(B, 47, device="cuda:0"); Y = torch.randn(B, 12, device="cuda:0")
def step():
    pol.train_step(X, Y, 1e-3)
for _ in range(10): step()
torch.cuda.synchronize()
def timeit(f, n=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
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
