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


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
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
    d = run_bench("--steps", "4", "--warmup", "2", "--batch", "64")
    check_common(d, 4, 2)
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
