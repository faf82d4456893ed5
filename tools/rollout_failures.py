#!/usr/bin/env python3
"""Diagnostic: which solves fail in the pushed rollouts of bench.py --rollouts, and why.

Drives the configs[3] rollouts from the host (BatchedLocomotionMPC.open_loop's loop, footsteps on) with the pushes of
bench.py, and records per rollout: the replan index and status of the first failed solve, the step norms of the solves
before it, the base state at that replan -- and dumps the complete inputs of the first failing solve of a few rollouts
(x0, yref, yref_e, params, X/U before the shift, shift) so that the solve can be replayed on the CPU.

    python tools/rollout_failures.py --batch 1024 --out gpurun_out/rollfail
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from iterative_learning_nmpc_amd.mpc import BatchedLocomotionMPC  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--time", type=float, default=2.0)
    ap.add_argument("--line-search", type=int, default=0)
    ap.add_argument("--dump", type=int, default=24, help="failing solves to dump")
    ap.add_argument("--out", default="gpurun_out/rollfail")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    B = a.batch
    rng = np.random.default_rng(a.seed)                      # bench.py rollout_mode, rank 0
    x = np.zeros((B, 12)); x[:, 2] = 0.3
    force = rng.uniform(-1, 1, (B, 3)); force /= np.linalg.norm(force, axis=1, keepdims=True) + 1e-6
    force *= rng.uniform(50, 70, (B, 1))
    force[0] = 0.0
    push = dict(start=0.2, duration=0.3, force=force)
    mpc = BatchedLocomotionMPC(B, n_nodes=50, device="cuda:0", footsteps=True)
    mpc.solver.set_line_search(bool(a.line_search))
    mpc.set_command(np.array([0.3, 0.0, 0.0]), 0.0)
    stash = {}
    build = mpc.build_problem

    def build_and_keep(xx):
        out = build(xx)
        stash["yref"], stash["yref_e"], stash["params"] = (np.array(v, np.float32) for v in out)
        return out
    mpc.build_problem = build_and_keep

    dt_replan = mpc.replanning_steps * mpc.sim_dt
    n_replans = int(np.floor(a.time / dt_replan + 1e-9))
    first_fail = np.full(B, -1)
    fail_status = np.zeros(B, int)
    stepn = np.zeros((B, n_replans), np.float32)
    cost = np.zeros((B, n_replans), np.float32)
    states = np.zeros((B, n_replans, 12))
    dumped = 0
    for i in range(n_replans):
        t_now = i * dt_replan
        states[:, i] = x
        mpc.set_convergence_on_first_iter()
        Xp, Up = mpc.X.clone(), mpc.U.clone()
        node_before, last_before, first = mpc.current_opt_node, mpc.solver.last_node, mpc.first_solve
        X, _ = mpc.optimize(x)
        torch.cuda.synchronize()
        st = mpc.status.cpu().numpy()
        stats = mpc.stats.cpu().numpy()
        stepn[:, i], cost[:, i] = stats[:, 1], stats[:, 0]
        new = np.where(((st == 1) | (st == 4)) & (first_fail < 0))[0]
        first_fail[new], fail_status[new] = i, st[new]
        for b in new[: max(0, a.dump - dumped)]:
            np.savez(os.path.join(a.out, f"fail_b{b}_r{i}.npz"), x0=x[b].astype(np.float32), yref=stash["yref"][b],
                     yref_e=stash["yref_e"][b], params=stash["params"][b], X=Xp[b].cpu().numpy(), U=Up[b].cpu().numpy(),
                     shift=0 if first else node_before - last_before, status=st[b], replan=i, force=force[b],
                     states=states[b, : i + 1], stepn=stepn[b, : i + 1])
            dumped += 1
        mpc.first_solve = False
        x = X[:, mpc.nodes_per_replan, :].double().cpu().numpy()
        mpc.touch_down(mpc._params)
        if push["start"] <= t_now < push["start"] + push["duration"]:
            x[:, 6:9] += force * dt_replan / mpc.mp[1]
        # a failed problem keeps NaN in x: keep it finite so the host helpers do not choke (the rollout is already counted)
        bad = ~np.isfinite(x).all(axis=1)
        x[bad] = states[bad, i]
        mpc.sim_step += mpc.replanning_steps
        mpc.current_opt_node += mpc.nodes_per_replan
        mpc.increment_base_ref_position(mpc.replanning_steps)
    failed = first_fail >= 0
    lim = np.deg2rad(25.0)
    summary = {
        "batch": B, "failed": int(failed.sum()), "status_nan": int((fail_status == 1).sum()), "status_qp": int((fail_status == 4).sum()),
        "first_fail_replan_hist": np.bincount(first_fail[failed], minlength=n_replans).tolist(),
        "line_search": a.line_search,
    }
    rows = []
    for b in np.where(failed)[0]:
        i = first_fail[b]
        s = states[b, i]
        rows.append({"b": int(b), "replan": int(i), "status": int(fail_status[b]), "force": force[b].round(1).tolist(),
                     "z": float(s[2]), "yaw_pitch_roll_deg": np.rad2deg(s[3:6]).round(1).tolist(), "v": s[6:9].round(2).tolist(),
                     "w": s[9:12].round(2).tolist(), "stepn_before": stepn[b, max(0, i - 3): i + 1].round(3).tolist(),
                     "unsafe_before": bool((np.abs(states[b, : i + 1, 4:6]) > lim).any() or (states[b, : i + 1, 2] < 0.18).any()
                                           or (states[b, : i + 1, 2] > 0.45).any())})
    summary["failures_after_unsafe_state"] = int(sum(r["unsafe_before"] for r in rows))
    # how far the surviving rollouts go: largest step norm, attitude, height
    ok = ~failed
    summary["ok_max_stepn"] = float(np.nanmax(stepn[ok])) if ok.any() else None
    summary["ok_unsafe_attitude"] = int((np.abs(states[ok][:, :, 4:6]) > lim).any(axis=(1, 2)).sum())
    summary["ok_unsafe_height"] = int(((states[ok][:, :, 2] < 0.18) | (states[ok][:, :, 2] > 0.45)).any(axis=1).sum())
    json.dump({"summary": summary, "failures": rows}, open(os.path.join(a.out, "summary.json"), "w"), indent=1)
    print(json.dumps(summary))
    for r in rows[:40]:
        print(r)


if __name__ == "__main__":
    main()
