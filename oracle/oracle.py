"""ctypes front-end of the CPU oracle (oracle/nmpc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The shipped package never imports this module.
Parity status of the solve: UNPINNED (see the header of nmpc_oracle.c).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MP_NAMES = ("dt", "mass", "Ixx", "Iyy", "Izz", "gz", "mu", "umax",
            "p_gain", "hipx", "hipy", "lhip", "l1", "l2", "res0", "res1")
MP_DEFAULTS = dict(dt=0.02, mass=15.0, Ixx=0.11, Iyy=0.27, Izz=0.33, gz=-9.81, mu=0.8, umax=0.0,
                   p_gain=50.0, hipx=0.19, hipy=0.047, lhip=0.095, l1=0.213, l2=0.213, res0=0.0, res1=0.0)
OPT_NAMES = ("max_sqp_iter", "n_ipm", "nlp_tol", "reg", "reg_e", "mu0", "sigma", "s_min",
             "gamma", "line_search", "rho", "yref_per_stage", "tau_min",
             "ipm_warm", "ws_s_floor", "ws_lam_floor", "ws_shift", "ws_have")
OPT_DEFAULTS = dict(max_sqp_iter=1, n_ipm=6, nlp_tol=0.0, reg=1e-6, reg_e=1e-5, mu0=10.0,
                    sigma=0.2, s_min=1.0, gamma=0.995, line_search=0, rho=1e3, yref_per_stage=0, tau_min=0.1,
                    ipm_warm=0, ws_s_floor=1e-2, ws_lam_floor=1e-2, ws_shift=0, ws_have=0)


def build(force: bool = False) -> None:
    """Compile both precisions with the committed Makefile (gcc only)."""
    targets = [os.path.join(_HERE, f"liboracle_{p}.so") for p in ("f64", "f32")]
    src = os.path.join(_HERE, "nmpc_oracle.c")
    stale = force or any(
        (not os.path.exists(t)) or os.path.getmtime(t) < os.path.getmtime(src) for t in targets)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)


class Oracle:
    """One precision of the oracle: Oracle('f64') or Oracle('f32')."""

    def __init__(self, precision: str = "f64"):
        assert precision in ("f64", "f32")
        build()
        self.dtype = np.float64 if precision == "f64" else np.float32
        self.lib = ctypes.CDLL(os.path.join(_HERE, f"liboracle_{precision}.so"))
        assert self.lib.oracle_real_size() == np.dtype(self.dtype).itemsize

    # -- helpers ---------------------------------------------------------
    def _a(self, x, shape=None):
        a = np.ascontiguousarray(np.asarray(x, dtype=self.dtype))
        if shape is not None:
            assert a.shape == tuple(shape), (a.shape, shape)
        return a

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None

    def dims(self, model_id: int):
        v = [ctypes.c_int() for _ in range(4)]
        rc = self.lib.oracle_dims(model_id, *[ctypes.byref(x) for x in v])
        if rc:
            raise ValueError(f"unknown model {model_id}")
        return tuple(x.value for x in v)  # nx, nu, np, ng

    def mp(self, **kw):
        d = dict(MP_DEFAULTS)
        d.update(kw)
        return self._a([d[k] for k in MP_NAMES])

    def _mp(self, mp):
        """model parameter vector padded to the oracle's length (older callers pass the first 8)"""
        mp = np.asarray(mp, dtype=self.dtype).ravel()
        full = self.mp()
        full[:mp.size] = mp
        return full

    def output_dims(self, model_id: int):
        ny, nye = ctypes.c_int(), ctypes.c_int()
        if self.lib.oracle_output_dims(model_id, ctypes.byref(ny), ctypes.byref(nye)):
            raise ValueError(f"unknown model {model_id}")
        return ny.value, nye.value

    def opt(self, **kw):
        d = dict(OPT_DEFAULTS)
        d.update(kw)
        return self._a([d[k] for k in OPT_NAMES])

    # -- API ---------------------------------------------------------------
    def dynamics(self, model_id, mp, x, u, p=None, jac=True):
        nx, nu, np_, _ = self.dims(model_id)
        x, u = self._a(x, (nx,)), self._a(u, (nu,))
        p = self._a(p if p is not None else np.zeros(max(np_, 1)))
        xn = np.zeros(nx, self.dtype)
        A = np.zeros((nx, nx), self.dtype) if jac else None
        B = np.zeros((nx, nu), self.dtype) if jac else None
        self.lib.oracle_dynamics(model_id, self._p(self._mp(mp)), self._p(x), self._p(u), self._p(p),
                                 self._p(xn), self._p(A), self._p(B))
        return (xn, A, B) if jac else xn

    def constraints(self, model_id, mp, p=None):
        nx, nu, np_, ng = self.dims(model_id)
        p = self._a(p if p is not None else np.zeros(max(np_, 1)))
        G = np.zeros((ng, nu), self.dtype)
        h = np.zeros(ng, self.dtype)
        act = np.zeros(ng, np.int32)
        self.lib.oracle_constraints(model_id, self._p(self._mp(mp)), self._p(p), self._p(G),
                                    self._p(h), self._p(act))
        return G, h, act

    def riccati(self, Q, R, q, r, A, B, d, dx0):
        N, nx, nu = A.shape[0], A.shape[1], B.shape[2]
        Q, R, q, r = self._a(Q, (N + 1, nx, nx)), self._a(R, (N, nu, nu)), self._a(q, (N + 1, nx)), self._a(r, (N, nu))
        A, B, d, dx0 = self._a(A), self._a(B, (N, nx, nu)), self._a(d, (N, nx)), self._a(dx0, (nx,))
        dX = np.zeros((N + 1, nx), self.dtype)
        dU = np.zeros((N, nu), self.dtype)
        K = np.zeros((N, nu, nx), self.dtype)
        kff = np.zeros((N, nu), self.dtype)
        P = np.zeros((N + 1, nx, nx), self.dtype)
        st = self.lib.oracle_riccati(nx, nu, N, *[self._p(a) for a in (Q, R, q, r, A, B, d, dx0, dX, dU, K, kff, P)])
        return dict(status=st, dX=dX, dU=dU, K=K, kff=kff, P=P)

    def solve_batch(self, model_id, N, mp, opt, W, We, x0, yref, yref_e, params, X, U, nthreads=0, ipm_state=None):
        """Solves in place on copies; returns (X, U, status, stats).
        ipm_state: None, or a dict {"S", "L"} of [B, N, ng] arrays (slacks, multipliers of the interior point), read if
        opt ws_have is set (warm start across calls, opt ipm_warm) and overwritten with the final state."""
        nx, nu, np_, _ = self.dims(model_id)
        x0 = self._a(x0)
        B = x0.shape[0]
        X = self._a(X, (B, N + 1, nx)).copy()
        U = self._a(U, (B, N, nu)).copy()
        params = self._a(params if np_ > 0 else np.zeros((B, N + 1, 1)))
        ny, nye = self.output_dims(model_id)
        yref, yref_e = self._a(yref), self._a(yref_e, (B, nye))
        opt = self._a(opt)
        per_stage = opt[OPT_NAMES.index("yref_per_stage")] != 0
        assert yref.shape == ((B, N, ny) if per_stage else (B, ny)), yref.shape
        assert np.size(W) == ny and np.size(We) == nye, (np.size(W), np.size(We))
        mp = self._mp(mp)
        if np_ > 0:
            assert params.shape == (B, N + 1, np_), params.shape
        status = np.zeros(B, np.int32)
        stats = np.zeros((B, 4), self.dtype)
        S = L = None
        if ipm_state is not None:
            ng = self.dims(model_id)[3]
            S = self._a(ipm_state.get("S", np.zeros((B, N, ng))), (B, N, ng)).copy()
            L = self._a(ipm_state.get("L", np.zeros((B, N, ng))), (B, N, ng)).copy()
        rc = self.lib.oracle_solve_batch_ws(model_id, N, B, *[self._p(self._a(a)) for a in (mp, opt, W, We)],
                                            self._p(x0), self._p(yref), self._p(yref_e), self._p(params),
                                            self._p(X), self._p(U), self._p(status), self._p(stats),
                                            int(nthreads), self._p(S), self._p(L))
        if ipm_state is not None:
            ipm_state["S"], ipm_state["L"] = S, L
        assert rc == 0
        return X, U, status, stats

    # -- whole-body model (model 2) test hooks ------------------------------------
    def wb_residuals(self, mp, x, u, p, yref=None, jac=True):
        """residuals (reference subtracted) and their dense state Jacobian; u=None: terminal node"""
        ny, nye = self.output_dims(2)
        n = nye if u is None else ny
        x, p = self._a(x, (42,)), self._a(p, (20,))
        u = None if u is None else self._a(u, (30,))
        yref = self._a(np.zeros(n) if yref is None else yref, (n,))
        res = np.zeros(n, self.dtype)
        J = np.zeros((n, 42), self.dtype) if jac else None
        self.lib.oracle_wb_residuals(self._p(self._mp(mp)), self._p(x), self._p(u), self._p(p), self._p(yref),
                                     self._p(res), self._p(J))
        return (res, J) if jac else res

    def wb_feet(self, mp, x):
        x = self._a(x, (42,))
        pos, vel = np.zeros((4, 3), self.dtype), np.zeros((4, 3), self.dtype)
        self.lib.oracle_wb_feet(self._p(self._mp(mp)), self._p(x), self._p(pos), self._p(vel))
        return pos, vel

    def shift_warm_start(self, X, U, shift, nu_keep=None):
        """nu_keep: leading inputs whose exposed tail keeps the previous values -- the whole-body model's 18
        accelerations (solver.py:316-322 zeroes the forces only); default by the input width"""
        B, N1, nx = X.shape
        nu = U.shape[2]
        if nu_keep is None:
            nu_keep = 18 if nu == 30 else 0
        X, U = self._a(X).copy(), self._a(U).copy()
        self.lib.oracle_shift_warm_start(nx, nu, int(nu_keep), N1 - 1, B, int(shift), self._p(X), self._p(U))
        return X, U

    def tracking_error(self, S, Snom):
        S, Snom = self._a(S), self._a(Snom)
        B, T, ns = S.shape
        assert Snom.shape == (T, ns)
        err = np.zeros((B, T), self.dtype)
        self.lib.oracle_tracking_error(B, T, ns, self._p(S), self._p(Snom), self._p(err))
        return err

    def num_threads(self):
        return self.lib.oracle_num_threads()
