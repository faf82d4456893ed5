"""Golden vectors for `compute_base_ref_cnt_restricted` (mpc_controller/mpc.py:274-315) from the reference itself.

Run in the build container only (needs /root/reference; never on the GPU box):
    python tests/golden/make_golden_cnt_restricted.py
Writes tests/golden/cnt_restricted.npz: contact-location plans [4, N+1, 3] (Raibert plans of the reference's own planner,
plans with unplanned all-zero locations, fully planned ones, a degenerate all-zero plan) and the two base references the
reference computes from them.  Placeholder modules for the absent third-party packages as in make_golden.py."""
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, REF, _Anything, _install_stubs  # noqa: E402


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    from mpc_controller.config.quadruped.mpc_gait import GaitConfigFactory
    from mpc_controller.mpc import LocomotionMPC
    from mpc_controller.utils.contact_planner import RaiberContactPlanner
    rng = np.random.default_rng(77)
    feet = ["FL_foot", "FR_foot", "RL_foot", "RR_foot"]
    hips = np.array([[0.1934, 0.142, 0.0], [0.1934, -0.142, 0.0], [-0.1934, 0.142, 0.0], [-0.1934, -0.142, 0.0]])
    plans, heights = [], []
    for i in range(6):                                   # plans of the reference's Raibert planner (zeros before a touch-down)
        pl = RaiberContactPlanner(feet, 0.04, GaitConfigFactory.get("trot" if i % 2 == 0 else "crawl"), hips.copy(),
                                  y_offset=0.02, x_offset=0.04, foot_size=0.0085, cache_cnt=False)
        pos = rng.normal(0, 0.2, 3) + [0, 0, 0.3]
        pl.set_state(pos, rng.normal(0, 0.2, 3), rng.normal(0, 0.1, 3), pos, np.array([rng.uniform(0.05, 0.4), rng.uniform(-0.1, 0.1), 0.0]),
                     rng.uniform(-0.3, 0.3))
        plans.append(pl.get_locations(int(rng.integers(0, 40)), 26))
    full = rng.normal(0, 0.3, (4, 26, 3))                # every location planned
    plans.append(full)
    part = full.copy(); part[1, :9] = 0.0; part[3, 20:] = 0.0      # unplanned stretches at either end
    plans.append(part)
    rep = np.repeat(rng.normal(0, 0.3, (4, 3, 3)), [10, 9, 7], axis=1)   # few distinct sets, repeated (what a plan looks like)
    plans.append(rep)
    plans.append(np.zeros((4, 26, 3)))                   # nothing planned
    out = {}
    for i, loc in enumerate(plans):
        mpc = object.__new__(LocomotionMPC)
        mpc.executor, mpc.velocity_goal = _Anything(), None
        mpc.config_gait = types.SimpleNamespace(nom_height=0.30 if i % 2 else 0.33)
        mpc.height_offset = 0.0 if i % 3 else 0.015
        b, be = mpc.compute_base_ref_cnt_restricted(np.zeros(19), loc.copy())
        out[f"loc{i}"] = loc
        out[f"ref{i}"] = np.stack([b, be])
        heights.append([mpc.config_gait.nom_height, mpc.height_offset])
    out["heights"] = np.array(heights)
    np.savez_compressed(os.path.join(OUT, "cnt_restricted.npz"), **out)
    print("written", len(plans), "cases")


if __name__ == "__main__":
    main()
