"""Wall-clock timing decorator with the reference's timing keys
(mpc_controller/utils/profiling.py:6-32: `optimize`, `init_solver`, `solve`,
`warm_start_solver`, `update_solver`)."""
from __future__ import annotations

import functools
import time
from typing import Dict, List

import numpy as np


def time_fn(key: str):
    """Append the call's duration in ms to `self.timings[key]` when `self.compute_timings`."""
    def wrap(fn):
        @functools.wraps(fn)
        def timed(self, *args, **kwargs):
            if not getattr(self, "compute_timings", False):
                return fn(self, *args, **kwargs)
            t0 = time.perf_counter()
            out = fn(self, *args, **kwargs)
            self.timings[key].append((time.perf_counter() - t0) * 1e3)
            return out
        return timed
    return wrap


def summarize_timings(timings: Dict[str, List[float]]) -> Dict[str, Dict[str, float]]:
    """mean/std/max over all samples but the first, plus the first (profiling.py:23-32)."""
    out = {}
    for key, samples in timings.items():
        rest = np.asarray(samples[1:] if len(samples) > 1 else samples, dtype=float)
        out[key] = dict(mean=float(rest.mean()), std=float(rest.std()), max=float(rest.max()),
                        first=float(samples[0]))
    return out


def print_timings(timings: Dict[str, List[float]]) -> None:
    for key, s in summarize_timings(timings).items():
        print("---", key, "---")
        print(f"mean: {s['mean']:.2f} ms")
        print(f"std: {s['std']:.2f} ms")
        print(f"max: {s['max']:.2f} ms")
        print(f"first: {s['first']:.2f} ms")
