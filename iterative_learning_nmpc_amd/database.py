"""Device-resident training database -- host mirror of the reference's `Database`
(DAgger/utils/database.py:9-315; Behavior_Cloning/utils/database.py is the same file) over the C-ABI of
include/nmpc_dataset.h.

Same surface and semantics with the rows in HBM: `append` (ring buffer of `limit` rows, statistics
refreshed after every append), `calc_input_mean_std`, `get_database_mean_std`, `set_goal_type`,
`set_normalize_input`, `__len__`, `save_as_npz` / `load_from_npz` (the reference's file schema).  Instead of
one `(x, y)` item per `__getitem__` call, `batch(idx)` assembles a whole normalised batch on the device --
idx typically comes from `policy.weighted_sample` -- ready for `DevicePolicy.train_step`.

Rows are stored in fp32 (what the rollout kernels produce and the network consumes); the statistics are
float64 like numpy's.  Indices are PHYSICAL ring positions, as in the reference (`self.states[index]`)."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib

FIELDS = ("states", "vc_goals", "cc_goals", "actions")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _check(rc, what):
    if rc:
        raise _lib.NmpcError(f"{what}: {_lib.load().nmpc_dataset_last_error().decode()}")


def column_stats(table: torch.Tensor, rows: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """np.mean(table[:rows], axis=0), np.std(table[:rows], axis=0) in float64 on the device."""
    lib = _lib.load()
    assert table.is_cuda and table.dtype == torch.float32 and table.is_contiguous() and table.dim() == 2
    rows = table.shape[0] if rows is None else int(rows)
    cols = table.shape[1]
    out = torch.empty(2, cols, dtype=torch.float64, device=table.device)
    scratch = torch.empty(lib.nmpc_column_stats_scratch(cols), dtype=torch.float64, device=table.device)
    _check(lib.nmpc_column_stats(_ptr(table), rows, cols, _ptr(out[0]), _ptr(out[1]), _ptr(scratch), _stream(table.device)),
           "nmpc_column_stats")
    return out[0], out[1]


class DeviceDatabase:
    def __init__(self, limit: int, n_state: int = 44, n_action: int = 12, n_vc_goal: int = 3, n_cc_goal: int = 8,
                 norm_input: bool = True, goal_type: str = "vc", device="cuda:0"):
        assert goal_type in ("vc", "cc"), "Goal type can only be vc or cc"
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceDatabase needs a HIP device: there is no CPU path")
        _lib.load()
        self.device = torch.device(device)
        self.limit, self.length, self.start = int(limit), 0, 0
        self.widths = {"states": n_state, "vc_goals": n_vc_goal, "cc_goals": n_cc_goal, "actions": n_action}
        self.tables = {f: torch.zeros(self.limit, w, dtype=torch.float32, device=self.device) for f, w in self.widths.items()}
        self.has = {"vc_goals": False, "cc_goals": False}
        self.norm_input, self.goal_type = bool(norm_input), goal_type
        self.states_mean = self.states_std = None          # float64 device vectors
        self.cc_goals_mean = self.cc_goals_std = None
        self.vc_goals_mean, self.vc_goals_std = 0.0, 1.0   # database.py:240-243: 'vc' goals are not normalised

    def __len__(self):
        return self.length

    def set_normalize_input(self, value: bool):
        self.norm_input = bool(value)

    def set_goal_type(self, value: str):
        assert value in ("vc", "cc"), "Goal type can only be vc or cc"
        self.goal_type = value

    # ------------------------------------------------------------------ aggregation
    def append(self, states, actions, vc_goals=None, cc_goals=None):
        """database.py:105-154.  Arguments: device (or host) arrays [n, width]."""
        if vc_goals is None and cc_goals is None:
            raise ValueError("both vc_goals and cc_goals cant be empty!")
        lib = _lib.load()
        given = {"states": states, "actions": actions, "vc_goals": vc_goals, "cc_goals": cc_goals}
        n = None
        for f, a in list(given.items()):
            if a is None:
                continue
            a = torch.as_tensor(a, dtype=torch.float32, device=self.device).contiguous()
            if a.dim() != 2 or a.shape[1] != self.widths[f] or (n is not None and a.shape[0] != n):
                raise ValueError(f"{f}: expected [{n if n is not None else 'n'}, {self.widths[f]}], got {tuple(a.shape)}")
            n = a.shape[0]
            given[f] = a
        first_slot = (self.start + self.length) % self.limit
        for f, a in given.items():
            if a is None:
                continue
            _check(lib.nmpc_ring_append(_ptr(a), a.shape[1], n, _ptr(self.tables[f]), self.limit, first_slot,
                                        _stream(self.device)), "nmpc_ring_append")
            if f in self.has:
                self.has[f] = True
        grow = min(n, self.limit - self.length)             # room first, then the start moves (:124-131)
        self.length += grow
        self.start = (self.start + n - grow) % self.limit
        self.calc_input_mean_std()

    def calc_input_mean_std(self):
        """database.py:208-255: statistics over the physical rows [0, length)."""
        if self.length == 0:
            return
        self.states_mean, self.states_std = column_stats(self.tables["states"], self.length)
        if self.has["cc_goals"]:
            self.cc_goals_mean, self.cc_goals_std = column_stats(self.tables["cc_goals"], self.length)

    def get_database_mean_std(self):
        """[states_mean, states_std, goal_mean, goal_std] (numpy float64) or None -- database.py:257-271."""
        if not self.norm_input:
            return None
        out = [self.states_mean.cpu().numpy(), self.states_std.cpu().numpy()]
        if self.goal_type == "vc":
            return out + [self.vc_goals_mean, self.vc_goals_std]
        return out + [self.cc_goals_mean.cpu().numpy(), self.cc_goals_std.cpu().numpy()]

    # ------------------------------------------------------------------ batches
    def batch(self, idx: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x[i] = hstack(state_norm, goal)[idx[i]], y[i] = actions[idx[i]] as fp32 (database.py:54-84 and
        the `.float()` of train_locosafedagger.py:95) for int32 device indices."""
        lib = _lib.load()
        assert idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous() and idx.dim() == 1
        if self.length == 0:
            raise IndexError("the database is empty")
        goal_field = "vc_goals" if self.goal_type == "vc" else "cc_goals"
        if not self.has[goal_field]:
            raise ValueError(f"no {goal_field} were appended")
        n_state, n_goal, n_action = self.widths["states"], self.widths[goal_field], self.widths["actions"]
        x = torch.empty(idx.numel(), n_state + n_goal, dtype=torch.float32, device=self.device)
        y = torch.empty(idx.numel(), n_action, dtype=torch.float32, device=self.device)
        s_mean = s_std = g_mean = g_std = None
        if self.norm_input:
            s_mean, s_std = self.states_mean, self.states_std
            if self.goal_type == "cc":
                g_mean, g_std = self.cc_goals_mean, self.cc_goals_std
        _check(lib.nmpc_assemble_batch(_ptr(self.tables["states"]), n_state, _ptr(s_mean), _ptr(s_std), 1,
                                       _ptr(self.tables[goal_field]), n_goal, _ptr(g_mean), _ptr(g_std),
                                       _ptr(self.tables["actions"]), n_action, self.length, _ptr(idx), idx.numel(),
                                       _ptr(x), _ptr(y), _stream(self.device)), "nmpc_assemble_batch")
        return x, y

    # ------------------------------------------------------------------ files (npz first)
    def save_as_npz(self, filename: str):
        """database.py:273-282: keys states, vc_goals, cc_goals, actions -- float64 arrays [length, width]."""
        np.savez(filename, **{f: self.tables[f][:self.length].cpu().numpy().astype(np.float64) for f in FIELDS})

    def load_from_npz(self, filename: str):
        """database.py:284-315: replaces the contents, then recomputes the statistics."""
        data = np.load(filename)
        for f in FIELDS:
            if f not in data:
                raise ValueError(f"Missing field '{f}' in NPZ file.")
        n = len(data["states"])
        if n > self.limit:
            raise ValueError(f"{n} rows do not fit limit={self.limit}")
        for f in FIELDS:
            a = np.asarray(data[f], dtype=np.float32)
            if a.shape != (n, self.widths[f]):
                raise ValueError(f"{f}: expected {(n, self.widths[f])}, got {a.shape}")
            self.tables[f][:n] = torch.from_numpy(a).to(self.device)
        self.has = {"vc_goals": True, "cc_goals": True}
        self.length, self.start = n, 0
        self.calc_input_mean_std()
