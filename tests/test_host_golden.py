"""Host-side helpers against golden vectors generated from the reference itself
(tests/golden/make_golden.py imports the reference's modules and records inputs/outputs).
Integer tables must match bit for bit; float64 helpers to 1e-12."""
import json
import os

import numpy as np
import pytest

from iterative_learning_nmpc_amd import references as refs
from iterative_learning_nmpc_amd.config import (CostConfigFactory, GaitConfigFactory, get_quadruped_config)
from iterative_learning_nmpc_amd.contact_planner import ContactPlanner, RaiberContactPlanner

FEET = ["FL_foot", "FR_foot", "RL_foot", "RR_foot"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_config_constants_match_reference(golden_dir):
    with open(os.path.join(golden_dir, "config.json")) as f:
        g = json.load(f)
    for gait_name in ("trot", "slow_trot"):
        gait, opt, cost = get_quadruped_config(gait_name, "go2")
        ref = g[gait_name]
        for k, v in ref["gait"].items():
            assert np.allclose(getattr(gait, k), v) if not isinstance(v, str) else getattr(gait, k) == v, k
        for k, v in ref["opt"].items():
            if k == "dt_nodes":
                assert opt.get_dt_nodes() == v
            elif k == "dt_bounds":
                assert list(opt.get_dt_bounds()) == v
            elif k == "hpipm_mode":
                assert opt.hpipm_mode.name == v
            elif k == "opt_dt_scale":
                assert list(opt.opt_dt_scale) == v
            else:
                assert getattr(opt, k) == v, k
        for k, v in ref["cost"].items():
            mine = getattr(cost, k)
            assert (mine == v) if isinstance(v, str) else np.allclose(mine, v, rtol=1e-15, atol=0), k
    for name, ref in g["gaits"].items():
        mine = GaitConfigFactory.get(name)
        for k, v in ref.items():
            assert (getattr(mine, k) == v) if isinstance(v, str) else np.allclose(getattr(mine, k), v), (name, k)
    # the headline constants quoted in SURVEY.md 8c
    _, opt, _ = get_quadruped_config("trot", "go2")
    assert (opt.get_dt_nodes(), opt.get_dt_bounds(), opt.n_nodes, opt.max_iter, opt.max_qp_iter,
            opt.nlp_tol, opt.qp_tol) == (0.04, (0.02, 0.07), 25, 1, 6, 0.1, 0.01)
    with pytest.raises(ValueError):
        CostConfigFactory.get("go2", "gallop")
    with pytest.raises(ValueError):
        GaitConfigFactory.get("gallop")


def test_gait_tables_bit_exact(golden_dir):
    g = _load(golden_dir, "contact_planner.npz")
    for ci, case in enumerate(g["cases"]):
        gait, dt = str(case).split(":")
        pl = ContactPlanner(FEET, float(dt), GaitConfigFactory.get(gait))
        assert pl.nodes_per_cycle == int(g[f"c{ci}_npc"])
        for tab in ("gait_sequence", "switch_cnt", "peak_swing"):
            mine = getattr(pl, tab)
            assert mine.dtype == np.int8 and np.array_equal(mine, g[f"c{ci}_{tab}"]), (case, tab)
        for qi, (i_node, n) in enumerate(g["queries"]):
            assert np.array_equal(pl.get_contacts(int(i_node), int(n)), g[f"c{ci}_q{qi}_contacts"])
            assert np.array_equal(pl.get_peaks(int(i_node), int(n)), g[f"c{ci}_q{qi}_peaks"])
            mk, bk = pl.get_make_break_contacts(int(i_node), int(n))
            assert np.array_equal(mk, g[f"c{ci}_q{qi}_make"]) and np.array_equal(bk, g[f"c{ci}_q{qi}_break"])
        # batched windows are the stacked single windows
        nodes = g["queries"][:, 0]
        stacked = np.stack([pl.get_contacts(int(i), 26) for i in nodes])
        assert np.array_equal(pl.get_contacts_batch(nodes, 26), stacked)
        assert np.array_equal(pl.get_peaks_batch(nodes, 26), 1 - stacked)


def test_trot_table_known_answer():
    """FL = 000000111111, FR = 111111000000, window at node 3."""
    pl = ContactPlanner(FEET, 0.04, GaitConfigFactory.get("trot"))
    rows = ["".join(map(str, r)) for r in pl.gait_sequence]
    assert rows == ["000000111111", "111111000000", "111111000000", "000000111111"]
    assert "".join(map(str, pl.get_contacts(3, 26)[0])) == "00011111100000011111100000"
    assert pl._is_in_cnt("FR_foot", 2) and not pl._is_in_cnt("FL_foot", 2)


def test_raibert_locations(golden_dir):
    g = _load(golden_dir, "raibert.npz")
    for i, row in enumerate(g["inputs"]):
        pos, v_w, rpy, com, v_des, w_yaw, i_node = row[:3], row[3:6], row[6:9], row[9:12], row[12:15], row[15], int(row[16])
        pl = RaiberContactPlanner(FEET, 0.04, GaitConfigFactory.get("trot"), g["hips"].copy(),
                                  y_offset=0.02, x_offset=0.04, foot_size=0.0085, cache_cnt=False)
        pl.set_state(pos, v_w, rpy, com, v_des, w_yaw)
        assert np.allclose(pl.get_locations(i_node, 26), g[f"loc{i}"], rtol=0, atol=1e-12)


def test_base_references(golden_dir):
    g = _load(golden_dir, "references.npz")
    for i, row in enumerate(g["ref_in"]):
        q, v_des, w_des, state = row[:18], row[18:21], row[21:24], row[24:36]
        h_off = 0.0 if i % 3 else 0.02
        ref, ref_e = refs.base_ref_vel_tracking(q, v_des, w_des, state.copy(), 1.0, 0.30, h_off)
        assert np.allclose(ref, g["ref_out"][i, 0], rtol=0, atol=1e-12), i
        assert np.allclose(ref_e, g["ref_out"][i, 1], rtol=0, atol=1e-12), i
        st = state.copy()
        refs.increment_base_ref_position(st, v_des, w_des, 1.0e-3)
        assert np.allclose(st, g["inc_out"][i], rtol=0, atol=1e-15)
    # known answer of SURVEY.md 8c, including the crossed-bounds clip (0.36) and 0.15 -> 0.2 rounding
    q = np.zeros(18); q[2] = 0.3
    ref, ref_e = refs.base_ref_vel_tracking(q, [0.3, 0, 0], np.zeros(3), np.zeros(12), 1.0, 0.30)
    assert np.allclose(ref[[0, 2, 6]], [0.27, 0.3, 0.3]) and np.isclose(ref_e[0], 0.36)
    ref, _ = refs.base_ref_vel_tracking(q, [0.15, 0, 0], np.zeros(3), np.zeros(12), 1.0, 0.30)
    assert ref[6] == 0.2


def test_hermite_upsampling(golden_dir):
    g = _load(golden_dir, "references.npz")
    for name in ("a", "b", "c"):
        pos, vel = refs.hermite_upsample(g[f"{name}_t"], g[f"{name}_pos"], g[f"{name}_vel"],
                                         g[f"{name}_acc"], int(g[f"{name}_n"]))
        assert pos.shape == g[f"{name}_ipos"].shape
        assert np.allclose(pos, g[f"{name}_ipos"], rtol=0, atol=1e-11)
        assert np.allclose(vel, g[f"{name}_ivel"], rtol=0, atol=1e-11)
    assert np.array_equal(refs.zero_order_hold_index(1000, 25), g["id_repeat_1000_25"])


def test_euler_rate_maps(golden_dir):
    g = _load(golden_dir, "transform.npz")
    for ypr, w, a, b in zip(g["ypr"], g["w"], g["to_euler"], g["to_local"]):
        assert np.allclose(refs.local_angular_to_euler_derivative(ypr, w), a, rtol=0, atol=1e-14)
        assert np.allclose(refs.euler_derivative_to_local_angular(ypr, w), b, rtol=0, atol=1e-14)
        # the two maps are inverses of each other
        assert np.allclose(refs.euler_derivative_to_local_angular(ypr, a), w, atol=1e-12)


def test_base_ref_cnt_restricted(golden_dir):
    """`LocomotionMPC.compute_base_ref_cnt_restricted` (mpc.py:274-315): base references from a contact-location plan --
    Raibert plans of the reference's own planner, plans with unplanned (all-zero) stretches, fully planned, nothing
    planned.  Bit for bit the reference's outputs, and the facade's method is this function."""
    from iterative_learning_nmpc_amd.references import base_ref_cnt_restricted
    from iterative_learning_nmpc_amd.mpc_wholebody import LocomotionMPC
    g = np.load(os.path.join(golden_dir, "cnt_restricted.npz"))
    n = g["heights"].shape[0]
    assert n >= 10
    for i in range(n):
        h, off = g["heights"][i]
        ref, ref_e = base_ref_cnt_restricted(g[f"loc{i}"], h, off)
        assert np.array_equal(ref, g[f"ref{i}"][0]) and np.array_equal(ref_e, g[f"ref{i}"][1]), i
    mpc = LocomotionMPC(print_info=False)
    ref, ref_e = mpc.compute_base_ref_cnt_restricted(np.zeros(18), g["loc0"])
    want = base_ref_cnt_restricted(g["loc0"], mpc.config_gait.nom_height, mpc.height_offset)
    assert np.array_equal(ref, want[0]) and np.array_equal(ref_e, want[1])
