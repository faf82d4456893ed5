"""Trajectory `.npz` schema against a file written by the reference's own `StateDataRecorder`
(tests/golden/make_golden_trajectory.py)."""
import os

import numpy as np

from iterative_learning_nmpc_amd import trajectory_io as tio


def test_recorder_writes_what_the_reference_recorder_wrote(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "trajectory_recorder.npz"))
    ref = {k[4:]: g[k] for k in g.files if k.startswith("ref.")}
    assert tuple(ref) == tio.KEYS                                   # key set and order of the reference's file
    kp, kd = g["in.kp_kd"]
    rec = tio.TrajectoryRecorder(str(tmp_path), g["in.v_des"], float(g["in.current_time"]), nominal_flag=False,
                                 replanning_point=3, nth_traj_per_replanning=2, kp=kp, kd=kd)
    T = len(g["in.time"])
    for t in range(T):
        rec.record(float(g["in.time"][t]), g["in.qpos"][t], g["in.qvel"][t], g["in.ctrl"][t], g["in.feet"][t],
                   contact_vec=g["in.contact"][t], is_expert=int(g["in.is_expert"][t]), phase=0.0, cc_goals=ref["cc_goals"][t])
    path = rec.save()
    assert os.path.basename(path) == "traj_3_2.npz"
    mine = tio.load_trajectory(path)
    for k in tio.KEYS:
        assert mine[k].shape == ref[k].shape and mine[k].dtype == ref[k].dtype, (k, mine[k].dtype, ref[k].dtype)
        assert np.array_equal(mine[k], ref[k]), k                     # float64 arithmetic in the reference's order: bit for bit
    assert mine["state"].shape == (T, 44) and np.all(mine["state"][:, 0] == 0)   # the reference's phase slot is constant 0
    assert tio.TrajectoryRecorder("", nominal_flag=True).file_name("x") == "traj_nominal_x.npz"


def test_solver_layout_round_trip_to_mujoco_layout():
    """convert_to_mujoco (dynamics.py:75-98): quaternion of R = Rz Ry Rx, local angular velocity from Euler rates"""
    from iterative_learning_nmpc_amd.references import local_angular_to_euler_derivative, rpy_to_matrix
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.normal(0, 0.5, 18); v = rng.normal(0, 1.0, 18)
        q_mj, v_mj = tio.convert_to_mujoco(q, v)
        w, x, y, z = q_mj[3:7]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        assert abs(np.linalg.norm(q_mj[3:7]) - 1) < 1e-12 and np.abs(R - rpy_to_matrix(q[3:6][::-1])).max() < 1e-12
        assert np.allclose(local_angular_to_euler_derivative(q[3:6], v_mj[3:6]), v[3:6], atol=1e-12)   # convert_from_mujoco's map back
        assert np.array_equal(q_mj[:3], q[:3]) and np.array_equal(q_mj[7:], q[6:]) and np.array_equal(v_mj[6:], v[6:])


def test_loader_rejects_other_files(tmp_path):
    import pytest
    p = os.path.join(tmp_path, "x.npz")
    np.savez(p, time=np.zeros(3), state=np.zeros((3, 44)))
    with pytest.raises(ValueError, match="missing"):
        tio.load_trajectory(p)
