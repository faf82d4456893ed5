"""bench.py's output contract on the GPU: one JSON line with the driver's fields, the roofline object and (default
run) the cpu_baseline object.  Small batches and few steps: this checks the line, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
               "vs_baseline", "dtype", "data", "config", "roofline"}
ROOFLINE_KEYS = {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def run_bench(*args, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def check_common(d, steps, warmup):
    assert DRIVER_KEYS <= set(d), DRIVER_KEYS - set(d)
    assert ROOFLINE_KEYS <= set(d["roofline"]), ROOFLINE_KEYS - set(d["roofline"])
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0 < d["roofline"]["frac"] < 1
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    assert "workload" in d["config"] and "model" not in d["config"]


def test_headline_line_with_cpu_baseline_and_parity():
    d = run_bench("--steps", "4", "--warmup", "2", "--batch", "64", "--headline-only")
    check_common(d, 4, 2)
    assert "wholebody" not in d and "rollouts" not in d
    assert d["metric"].startswith("MPC solves/sec") and d["unit"] == "solves/s" and d["dtype"] == "f32"
    assert abs(d["value"] - 64 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["parity"]["rel_l2_X"] < 2e-5 and cb["parity"]["rel_l2_U"] < 2e-5
    assert d["failed_problems"] == 0 and d["cold_start"]["sqp_iterations"] == 15 and d["cold_start"]["failed_problems"] == 0


@pytest.mark.parametrize("mode", [("--policy", "64"), ("--database", "100000"), ("--torques", "256"), ("--rollouts", "16")])
def test_extra_modes_print_the_same_contract(mode):
    d = run_bench(*mode, "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    check_common(d, 2, 1)
    assert d["cpu_baseline"] is None if "cpu_baseline" in d else True


def test_default_line_carries_every_gpu_configuration():
    """the default line (what the driver records): the headline of configs[1] plus one sibling object per other GPU
    configuration of BASELINE.json -- here at reduced leg sizes, the structure is what is checked"""
    d = run_bench("--steps", "4", "--warmup", "2", "--wb-batch", "64", "--rollout-batch", "96")
    check_common(d, 4, 2)
    assert d["config"]["global_batch"] == 1024 and d["large_batch"]["batch"] == 8192
    for key, cfg in (("wholebody", "configs[2]"), ("mixed_precision", "configs[4]")):
        leg = d[key]
        assert leg["workload"].startswith(cfg) and leg["solves_per_s"] > 0 and leg["failed_problems"] == 0
        assert ROOFLINE_KEYS <= set(leg["roofline"]) and 0 < leg["roofline"]["frac"] < 1
        assert leg["parity"]["rel_l2_X"] < 1e-5 and leg["parity"]["rel_l2_U"] < 1e-5, (key, leg["parity"])
    assert d["wholebody"]["cpu_baseline"]["kind"] == "port" and d["wholebody"]["cpu_baseline"]["value"] > 0
    wr = d["wholebody"]["device_rollouts"]
    assert wr["rollouts"] == 64 and wr["replans"] in (50, 51) and wr["rollouts_per_s"] > 0 and wr["solver_failures"] == 0 and wr["finite"]
    assert "bf16" in d["mixed_precision"]["dtype"]
    r = d["rollouts"]
    assert r["workload"].startswith("configs[3]") and r["rollouts_per_s"] > 0 and r["failed_rollouts"] == 0
    assert r["valid_rollouts"] + r["zero_weight_rollouts"] == 96 and r["rollouts_run_per_attempt"][0] == 96
    assert set(r["unsafe_state_rollouts"]) == {"roll", "pitch", "height", "velocity_tracking"}
    assert {"invalid", "solver", "collision"} <= set(r["first_attempt"])


def test_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` as the driver may invoke it, with no launcher around it: the parent starts the ranks before it
    touches the GPU and relays rank 0's line.  On the one-GPU box the ranks share the device (NMPC_BENCH_BACKEND=gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "64",
                          "--no-cpu-baseline", "--no-cold-start"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(env, NMPC_BENCH_BACKEND="gloo"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["scaling"] == "weak"
    assert d["allgather"]["ranks"] == 2 and d["allgather"]["gathered_shape"] == [128, 51] and d["allgather"]["backend"] == "gloo"
