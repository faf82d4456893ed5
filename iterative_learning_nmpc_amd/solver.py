"""Batched NMPC solver on MI355X: Python host mirror of the reference's solver surface.

`BatchedNmpcSolver` is the batched drop-in for the solve path of
`QuadrupedAcadosSolver` (mpc_controller/utils/solver.py:15-429): the reference's
    set_max_iter / set_nlp_tol / set_qp_tol      solver.py:75-79, mpc.py:464-473
    update_cost / set_cost_weights                solver.py:100-141
    warm_start_solver(i_node)                     solver.py:290-342
    init(...) + solve()                           solver.py:355-429
keep their names and meaning; arrays gain a leading batch axis and live on the GPU as torch
tensors (containers only -- all arithmetic is in libnmpc_hip.so).  Per-problem failures are
reported in `status` (acados numbering) instead of exceptions (mpc.py:562-569).
"""
from __future__ import annotations

import ctypes
from collections import defaultdict
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from .profiling import time_fn
from .workloads import MODEL_DIMS, MP_NAMES


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream_ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class BatchedNmpcSolver:
    """B independent NMPC problems of one model, solved together (one problem per wavefront)."""

    def __init__(self, model_id: int, n_nodes: int, batch_max: int, device="cuda:0",
                 compute_timings: bool = False, precision: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedNmpcSolver needs a HIP device; there is no CPU path")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.model_id, self.n_nodes, self.batch_max = int(model_id), int(n_nodes), int(batch_max)
        d = MODEL_DIMS[self.model_id]
        self.nx, self.nu, self.np, self.ng = d["nx"], d["nu"], d["np"], d["ng"]
        self.ny, self.ny_e = d["ny"], d["ny_e"]      # cost residuals of a stage / of the terminal node
        self.compute_timings = compute_timings
        self.timings = defaultdict(list)
        self.last_node = 0
        # precision 0: fp32 throughout; 1: bf16 barrier product (mixed precision), see include/nmpc.h
        self.precision = int(precision)
        dims = _lib.NmpcDims(self.model_id, self.n_nodes, self.batch_max, self.precision)
        self._h = ctypes.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.lib.nmpc_create(ctypes.byref(dims), dev_index, ctypes.byref(self._h)), None, "nmpc_create")
        self._opts = dict(max_iter=1, max_qp_iter=6, nlp_tol=0.0, qp_tol=1e-2, line_search=0)
        self._push_opts()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.nmpc_destroy(h)
            self._h = None

    # -- configuration (names as in solver.py:75-79,100-141) ------------------------------------
    def _push_opts(self):
        o = self._opts
        _lib.check(self.lib.nmpc_set_opts(self._h, o["max_iter"], o["max_qp_iter"], o["nlp_tol"],
                                          o["qp_tol"], o["line_search"]), self._h, "nmpc_set_opts")

    def set_max_iter(self, n: int):
        self._opts["max_iter"] = int(n); self._push_opts()

    def set_max_qp_iter(self, n: int):
        self._opts["max_qp_iter"] = int(n); self._push_opts()

    def set_nlp_tol(self, tol: float):
        self._opts["nlp_tol"] = float(tol); self._push_opts()

    def set_qp_tol(self, tol: float):
        self._opts["qp_tol"] = float(tol); self._push_opts()

    def set_line_search(self, on: bool):
        self._opts["line_search"] = int(bool(on)); self._push_opts()

    # contact patterns of a trot: diagonal pairs (feet 0+3, 1+2), four-foot stance, flight -- as 4-bit stance flags
    COMMON_CONTACT_PATTERNS = frozenset({0b1001, 0b0110, 0b1111, 0b0000})

    def set_contact_patterns(self, all_patterns: bool = None, gait_sequence=None):
        """Choose the kernel by the gait: `gait_sequence[4, nodes]` (0/1 stance flags, contact_planner.py)
        whose patterns all belong to a trot keeps the default kernel; any other pattern selects the
        kernel with a static stage body for every contact pattern (see include/nmpc.h)."""
        if all_patterns is None:
            g = np.asarray(gait_sequence).astype(np.int64)
            pats = set((g[0] | (g[1] << 1) | (g[2] << 2) | (g[3] << 3)).tolist())
            all_patterns = not pats <= self.COMMON_CONTACT_PATTERNS
        _lib.check(self.lib.nmpc_set_contact_patterns(self._h, int(bool(all_patterns))), self._h, "nmpc_set_contact_patterns")
        return bool(all_patterns)

    def set_skip(self, flags: Optional[torch.Tensor], mask: int = 0):
        """Leave problems out of the following solves: flags int32 [batch_max] on the device (kept alive by the caller
        and by this object), a problem with flags[b] & mask != 0 is skipped -- X, U, status untouched, no time spent
        (terminated rollouts).  None detaches.  The flags are read when the kernels run."""
        if flags is not None:
            self._chk(flags, (self.batch_max,), "flags", torch.int32)
        self._skip_flags = flags
        _lib.check(self.lib.nmpc_set_skip(self._h, _ptr(flags), int(mask)), self._h, "nmpc_set_skip")

    def set_ipm(self, mu0=10.0, sigma=0.2, s_min=1.0, gamma=0.995, tau_min=0.1, merit_rho=1e3):
        _lib.check(self.lib.nmpc_set_ipm(self._h, mu0, sigma, s_min, gamma, tau_min, merit_rho),
                   self._h, "nmpc_set_ipm")

    def set_model_params(self, mp):
        mp = np.ascontiguousarray(mp, dtype=np.float32)
        assert mp.shape == (len(MP_NAMES),)
        _lib.check(self.lib.nmpc_set_model_params(
            self._h, mp.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(mp)), self._h, "nmpc_set_model_params")

    def set_cost_weights(self, W, W_e, reg_eps: float = 1e-6, reg_eps_e: float = 1e-5):
        W = np.ascontiguousarray(W, dtype=np.float32)
        W_e = np.ascontiguousarray(W_e, dtype=np.float32)
        assert W.shape == (self.ny,) and W_e.shape == (self.ny_e,)
        fp = ctypes.POINTER(ctypes.c_float)
        _lib.check(self.lib.nmpc_set_weights(self._h, W.ctypes.data_as(fp), W_e.ctypes.data_as(fp),
                                             reg_eps, reg_eps_e), self._h, "nmpc_set_weights")

    update_cost = set_cost_weights

    # -- tensors --------------------------------------------------------------------------------
    def _chk(self, t: torch.Tensor, shape, name: str, dtype=torch.float32) -> torch.Tensor:
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a torch tensor on {self.device}")
        if t.device.type != "cuda" or t.dtype != dtype or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: need contiguous {dtype} {tuple(shape)} on the GPU, got "
                             f"{t.dtype} {tuple(t.shape)} on {t.device}")
        return t

    def to_device(self, a, dtype=torch.float32) -> torch.Tensor:
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device).contiguous()

    # -- solve path -----------------------------------------------------------------------------
    @time_fn("warm_start_solver")
    def warm_start_solver(self, X: torch.Tensor, U: torch.Tensor, shift: int) -> None:
        """Shift the previous solution left by `shift` nodes in place (solver.py:304-322)."""
        B = X.shape[0]
        self._chk(X, (B, self.n_nodes + 1, self.nx), "X")
        self._chk(U, (B, self.n_nodes, self.nu), "U")
        _lib.check(self.lib.nmpc_shift_warm_start(self._h, B, int(shift), _ptr(X), _ptr(U),
                                                  _stream_ptr(self.device)), self._h, "nmpc_shift_warm_start")

    @time_fn("solve")
    def solve(self, x0, yref, yref_e, params, X, U, status=None, stats=None, shift: int = 0
              ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """Solve B problems in place on (X, U).  Returns (X, U, status[B] int32, stats[B,4]).

        shift > 0 folds `warm_start_solver(X, U, shift)` into the solve (same result, no extra
        launch: the first SQP iteration reads X, U through the shift's index map)."""
        B, N = x0.shape[0], self.n_nodes
        self._chk(x0, (B, self.nx), "x0")
        per_stage = yref.dim() == 3
        self._chk(yref, (B, N, self.ny) if per_stage else (B, self.ny), "yref")
        self._chk(yref_e, (B, self.ny_e), "yref_e")
        if self.np > 0:
            self._chk(params, (B, N + 1, self.np), "params")
        self._chk(X, (B, N + 1, self.nx), "X")
        self._chk(U, (B, N, self.nu), "U")
        if status is None:
            status = torch.empty(B, dtype=torch.int32, device=self.device)
        if stats is None:
            stats = torch.empty(B, 4, dtype=torch.float32, device=self.device)
        self._chk(status, (B,), "status", torch.int32)
        self._chk(stats, (B, 4), "stats")
        _lib.check(self.lib.nmpc_shift_solve_batch(
            self._h, B, int(shift), _ptr(x0), _ptr(yref), int(per_stage), _ptr(yref_e),
            _ptr(params) if self.np > 0 else None, _ptr(X), _ptr(U), _ptr(status), _ptr(stats),
            _stream_ptr(self.device)), self._h, "nmpc_shift_solve_batch")
        return X, U, status, stats

    def riccati(self, Q, R, q, r, A, Bm, d, dx0):
        """LQ core on explicit dense stage data (test / building-block entry)."""
        Bsz, N, nx = A.shape[0], A.shape[1], A.shape[2]
        nu = Bm.shape[3]
        assert N == self.n_nodes
        dX = torch.empty(Bsz, N + 1, nx, dtype=torch.float32, device=self.device)
        dU = torch.empty(Bsz, N, nu, dtype=torch.float32, device=self.device)
        status = torch.empty(Bsz, dtype=torch.int32, device=self.device)
        for t in (Q, R, q, r, A, Bm, d, dx0):
            assert t.is_contiguous() and t.dtype == torch.float32 and t.device.type == "cuda"
        _lib.check(self.lib.nmpc_riccati_batch(
            self._h, Bsz, nx, nu, _ptr(Q), _ptr(R), _ptr(q), _ptr(r), _ptr(A), _ptr(Bm), _ptr(d),
            _ptr(dx0), _ptr(dX), _ptr(dU), _ptr(status), _stream_ptr(self.device)), self._h, "nmpc_riccati_batch")
        return dX, dU, status

    def debug_tile(self, b: int, k: int, which: int) -> np.ndarray:
        """16x16 stage tile (row, col) of problem b: 0 A~, 1 B~, 2 K~, 3 Acl~ (test hook)."""
        out = np.zeros(256, dtype=np.float32)
        _lib.check(self.lib.nmpc_debug_read_tile(self._h, b, k, which,
                                                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))),
                   self._h, "nmpc_debug_read_tile")
        return out.reshape(16, 16).copy()   # logical (row, col), zero padded

    def debug_workspace(self, b: int, offset: int, count: int) -> np.ndarray:
        """raw floats of problem b's workspace (test hook of the whole-body kernels)"""
        out = np.zeros(count, dtype=np.float32)
        _lib.check(self.lib.nmpc_debug_read_workspace(self._h, b, offset, count,
                                                      out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))),
                   self._h, "nmpc_debug_read_workspace")
        return out

    def debug_wb_layout(self) -> dict:
        out = (ctypes.c_size_t * 8)()
        _lib.check(self.lib.nmpc_debug_wb_layout(self.n_nodes, out), self._h, "nmpc_debug_wb_layout")
        return dict(zip(("rec", "js", "qt", "kt", "arr", "stride", "NS", "REC"), (int(v) for v in out)))

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.nmpc_workspace_bytes(self._h))


def tracking_error(S: torch.Tensor, S_nom: torch.Tensor, threshold: float = 4.0,
                   ood_weight: float = 5.0, with_weights: bool = True):
    """err[b,t] = ||S[b,t,1:] - S_nom[t,1:]||_2 and the OOD sampling weights
    (data_collection_force_perturbation.py:138-156; test_train_policy.py:127-134)."""
    lib = _lib.load()
    B, T, ns = S.shape
    assert S_nom.shape == (T, ns)
    for t in (S, S_nom):
        assert t.is_contiguous() and t.dtype == torch.float32 and t.device.type == "cuda"
    err = torch.empty(B, T, dtype=torch.float32, device=S.device)
    w = torch.empty(B, T, dtype=torch.float32, device=S.device) if with_weights else None
    _lib.check(lib.nmpc_tracking_error(None, B, T, ns, _ptr(S), _ptr(S_nom), _ptr(err), _ptr(w),
                                       threshold, ood_weight, _stream_ptr(S.device)), None, "nmpc_tracking_error")
    return (err, w) if with_weights else err
