#!/usr/bin/env python3
"""Golden trajectory file of the rollout recorder (the `.npz` schema of DAgger/utils/RolloutMPC.py:135-166),
written by the reference's own `StateDataRecorder.record` / `.save` (RolloutMPC.py:102-258).

    python tests/golden/make_golden_trajectory.py        # needs /root/reference; writes trajectory_recorder.npz

MuJoCo, mj_pin, pinocchio and contact_tamp are absent from the image; none of them takes part in what the recorder
stores.  They are inert placeholder modules (as in make_golden.py), with two exceptions that the recorder calls:
`mj_pin.abstract.DataRecorder` (a base class holding the two constructor arguments) and `mj_pin.utils.mj_frame_pos`
(returns the foot position the synthetic `mj_data` carries).  The fixture holds data only: the synthetic simulator
samples that were fed in, and the arrays the reference wrote."""
import os
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
import make_golden as mg  # noqa: E402  (the placeholder-module machinery)

REF = mg.REF


def main():
    mg._install_stubs()
    abstract = types.ModuleType("mj_pin.abstract")

    class DataRecorder:                                   # holds what the reference's subclass passes up
        def __init__(self, record_dir="", record_step=1):
            self.record_dir, self.record_step = record_dir, record_step

        def get_date_time_str(self):
            return "fixture"

    abstract.DataRecorder = DataRecorder
    abstract.VisualCallback = type("VisualCallback", (), {"__init__": lambda self, *a, **k: None})
    abstract.PinController = type("PinController", (), {"__init__": lambda self, *a, **k: None})
    abstract.__getattr__ = lambda name: mg._Anything()
    sys.modules["mj_pin.abstract"] = abstract
    utils = types.ModuleType("mj_pin.utils")
    utils.get_robot_description = lambda name: SimpleNamespace(xml_path="")
    utils.mj_frame_pos = lambda model, data, name: data.feet[name]
    utils.pin_frame_pos = lambda *a, **k: None
    utils.__getattr__ = lambda name: mg._Anything()
    sys.modules["mj_pin.utils"] = utils
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "DAgger", "utils"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_rollout_mpc", os.path.join(REF, "DAgger/utils/RolloutMPC.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    rng = np.random.default_rng(21)
    T = 40
    v_des = np.array([0.15, 0.0, 0.0])
    geom = {"FL": 20, "FR": 32, "RL": 44, "RR": 56}
    inputs = dict(qpos=np.zeros((T, 19)), qvel=np.zeros((T, 18)), ctrl=np.zeros((T, 12)), time=np.zeros(T),
                  feet=np.zeros((T, 4, 3)), contact=np.zeros((T, 4), np.int64), is_expert=np.zeros(T, np.int64))
    with tempfile.TemporaryDirectory() as d:
        rec = ref.StateDataRecorder(d, 1, v_des, current_time=0.5, nominal_flag=False, replanning_point=3, nth_traj_per_replanning=2)
        np.random.seed(7)                                 # the recorder draws its contact-conditioned goals from numpy's global stream
        for t in range(T):
            q = rng.normal(0, 0.3, 19); q[2] = 0.3 + 0.01 * rng.normal()
            quat = rng.normal(size=4); q[3:7] = quat / np.linalg.norm(quat)
            v, ctrl = rng.normal(0, 0.5, 18), rng.normal(0, 5.0, 12)
            feet = q[:3] + rng.normal(0, 0.2, (4, 3))
            cnt = rng.integers(0, 2, 4)
            contacts = [SimpleNamespace(geom1=0, geom2=geom[n]) if i % 2 == 0 else SimpleNamespace(geom1=geom[n], geom2=0)
                        for i, n in enumerate(("FL", "FR", "RL", "RR")) if cnt[i]]
            mj_data = SimpleNamespace(qpos=q, qvel=v, ctrl=ctrl, time=round(0.001 * (t + 1), 4), ncon=len(contacts), contact=contacts,
                                      feet={n: feet[i] for i, n in enumerate(("FL", "FR", "RL", "RR"))})
            expert = int(rng.integers(0, 2))
            rec.record(mj_data, is_expert=expert)
            for k, val in (("qpos", q), ("qvel", v), ("ctrl", ctrl), ("time", mj_data.time), ("feet", feet), ("contact", cnt), ("is_expert", expert)):
                inputs[k][t] = val
        path = rec.save()
        assert os.path.basename(path) == "traj_3_2.npz", path
        written = dict(np.load(path))
    out = {f"in.{k}": v for k, v in inputs.items()}
    out.update({f"ref.{k}": v for k, v in written.items()})
    out["in.v_des"], out["in.current_time"], out["in.kp_kd"] = v_des, np.float64(0.5), np.array([ref.kp, ref.kd])
    np.savez_compressed(os.path.join(here, "trajectory_recorder.npz"), **out)
    print("written", {k: v.shape for k, v in written.items()})


if __name__ == "__main__":
    main()
