#!/usr/bin/env python3
"""Bit-identity check between two builds of libnmpc_hip.so (a kernel rewrite that claims to keep every rounding):
    NMPC_HIP_LIB=tools/_ab/lib_x.so python tools/ab_bits.py <tag>      -> gpurun_out/bits_<tag>.json (digests)
    python tools/ab_bits.py --compare tagA tagB
Solves fixed seeded batches of both model families (steady-state and multi-iteration policies, folded shift) and hashes
X, U, status, stats."""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def digests():
    import torch
    from iterative_learning_nmpc_amd import workloads as wl
    from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver
    out = {}
    cases = [("wb", wl.wholebody_trot(B=96, N=30, seed=1), 1), ("wb3", wl.wholebody_trot(B=40, N=25, seed=2), 3),
             ("wbfp", wl.wholebody_trot(B=24, N=30, seed=4, foot_placement=1e3), 2),
             ("c", wl.centroidal_trot(B=128, N=50, seed=1), 1), ("c3", wl.centroidal_trot(B=64, N=50, seed=2), 3)]
    for name, w, sqp in cases:
        s = BatchedNmpcSolver(w.model_id, w.N, w.B, "cuda:0")
        s.set_model_params(w.mp); s.set_cost_weights(w.W, w.W_e, w.meta["reg"], w.meta["reg_e"]); s.set_max_iter(sqp)
        t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
        X, U, st, stats = s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], t["X"], t["U"])
        X, U, st, stats = s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], X, U, shift=1)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for a in (X, U, st, stats):
            h.update(a.cpu().numpy().tobytes())
        out[name] = h.hexdigest()
    return out


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        a, b = (json.load(open(os.path.join(ROOT, "gpurun_out", f"bits_{t}.json"))) for t in sys.argv[2:4])
        same = {k: a[k] == b[k] for k in a}
        print(same)
        sys.exit(0 if all(same.values()) else 1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(digests(), open(os.path.join(ROOT, "gpurun_out", f"bits_{sys.argv[1]}.json"), "w"), indent=1)
