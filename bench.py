#!/usr/bin/env python3
"""Throughput harness of the batched NMPC solve path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (WORLD_SIZE unset): this process starts N ranks of itself through
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` BEFORE anything touches
the GPU, lets rank 0 print the JSON line and exits with the ranks' worst return code.  Launched by torchrun itself
(the driver's form) the ranks are used as they come.

A step = one pass of the hot path over one batch: warm-start shift by one node + one NMPC solve
(1 SQP iteration x 6 interior-point Riccati sweeps, the reference's steady-state policy,
mpc_controller/config/quadruped/mpc_opt.py:25-27) for B = 1024 centroidal problems per GPU
(BASELINE configs[1]: nx = nu = 12, N = 50, fp32), inputs resident in HBM.  Independent problems
shard across GPUs with no data-path collective (weak scaling); for N > 1 every step carries the one exchange of the
path, the all-gather of the tracking errors.
Rank 0 prints ONE JSON line.  At N = 1 the line also carries one sibling object per other GPU configuration of
BASELINE.json, each measured in the same run and each with its own roofline / cpu_baseline / parity:
  "wholebody"        configs[2]: B = 8192 whole-body problems (nx 42, nu 30, N 30)
  "rollouts"         configs[3], per-GPU slice: 8192 pushed 2 s rollouts with discard-and-redo of failed ones
  "mixed_precision"  configs[4]: the whole-body solve with the bf16 Gauss-Newton contraction
(--headline-only leaves them out).  See DESIGN.md section 6 for the roofline accounting.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector = fp32-input MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
MIXED_PRECISION_SHIPPED = 3   # nmpc_dims.precision of the recommended configs[4] variant: three-way split bf16


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=("centroidal", "wholebody"), default="centroidal",
                    help="centroidal: BASELINE configs[1], the headline (B = 1024, nx = nu = 12, N = 50); "
                         "wholebody: configs[2] (B = 8192, nx = 42, nu = 30, N = 30)")
    ap.add_argument("--batch", type=int, default=0, help="problems per GPU (default 1024 centroidal, 8192 whole-body)")
    ap.add_argument("--ipm", type=int, default=6)
    ap.add_argument("--sqp", type=int, default=1)
    ap.add_argument("--precision", type=int, default=0, choices=(0, 1, 2, 3),
                    help="0: fp32 (headline); BASELINE configs[4], mixed precision (a different config, not the headline): 1 = the "
                         "Gauss-Newton contraction on the bf16 matrix pipe, 2 = split bf16, 3 = three-way split bf16 (whole-body only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold-start", action="store_true", help="leave out the 15-iteration cold-start and large-batch variants (profiling runs)")
    ap.add_argument("--headline-only", action="store_true",
                    help="leave out the sibling legs (wholebody / rollouts / mixed_precision) of the default line")
    ap.add_argument("--wb-batch", type=int, default=8192, help="batch of the wholebody and mixed_precision legs of the default line")
    ap.add_argument("--rollout-batch", type=int, default=8192, help="rollouts of the rollouts leg of the default line")
    ap.add_argument("--rollouts", type=int, default=0,
                    help="extra mode (not the headline metric): B rollouts per GPU of 2 s (50 replans) fully "
                         "on the device, tracking error vs the nominal rollout, all-gather over the ranks")
    ap.add_argument("--policy", type=int, default=0,
                    help="extra mode (not the headline metric): training steps of the policy network on a batch of this size")
    ap.add_argument("--torques", type=int, default=0,
                    help="extra mode (not the headline metric): inverse dynamics + PD for this many robots")
    ap.add_argument("--database", type=int, default=0,
                    help="extra mode (not the headline metric): mean/std + batch assembly over a state table of this many rows")
    return ap.parse_args(argv)


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (torch.distributed.run, one process
    per GPU) and wait.  The parent has not imported torch and never touches the GPU; the children inherit stdout, so
    rank 0's JSON line is the parent's output.  Returns the launcher's return code (non-zero if any rank failed)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse_args()
    if _a.gpus > 1:
        sys.exit(spawn_ranks(_a.gpus))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from iterative_learning_nmpc_amd import workloads as wl  # noqa: E402
from iterative_learning_nmpc_amd.solver import BatchedNmpcSolver  # noqa: E402


def algorithmic_work(nx, nu, ng, np_, N, n_sweeps):
    """FLOP and HBM bytes one solve needs algorithmically (SURVEY.md 8d formulas)."""
    nz, ny = nx + nu, nx + nu
    f_riccati = N * (7 / 3 * nx ** 3 + 4 * nx * nx * nu + 2 * nx * nu * nu + nu ** 3 / 3)
    f_fwd = N * 2 * (nu * nx + nx * nz)
    f_barrier = N * 2 * ng * nu * (nu + 1) if n_sweeps > 1 or ng else 0    # G'DG and G'v
    f_lin = N * 2 * nx * nz * 3                                            # analytic A,B (c_model ~ 3)
    flops = n_sweeps * (f_riccati + f_fwd + f_barrier) + f_lin
    nbytes = 4 * (nx + (N + 1) * nx + N * nu + (N + 1) * np_ + N * ny + nx + (N + 1) * nx + N * nu)
    return flops, nbytes


def algorithmic_work_wholebody(N, n_sweeps):
    """SURVEY 8(d) for configs[2] (nx 42, nu 30): dense Riccati and forward terms per interior-point sweep; the
    Gauss-Newton contraction counted as what is dense in it (22 residual rows x 43 homogeneous columns, per node --
    the survey's 2 ny nz^2 with ny = 60 would be 18.7 MFLOP); barrier terms are closed-form 3x3 blocks."""
    nx, nu, np_, ny, ny_e = 42, 30, 20, 90, 66
    nz = nx + nu
    f_riccati = N * (7 / 3 * nx ** 3 + 4 * nx * nx * nu + 2 * nx * nu * nu + nu ** 3 / 3)
    f_fwd = N * 2 * (nu * nx + nx * nz)
    f_gn = (N + 1) * 2 * 22 * 43 * 43
    f_lin = (N + 1) * 6000
    flops = n_sweeps * (f_riccati + f_fwd) + f_gn + f_lin
    nbytes = 4 * (nx + (N + 1) * nx + N * nu + (N + 1) * np_ + N * ny + ny_e + (N + 1) * nx + N * nu)
    return flops, nbytes


def host_cores() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(w, n_ipm, sample, gpu_first=None, seconds=10.0):
    """The CPU oracle (fp64 restatement; the reference's acados solver is not installable here)
    on all host cores, same workload, bounded sample.  gpu_first = (X, U) of the device's first solve of
    the same inputs: the oracle, while it is at hand, also checks them (BASELINE's second metric, the
    relative L2 error of the trajectories)."""
    from oracle.oracle import Oracle
    o = Oracle("f64")
    sl = slice(0, sample)
    args = (w.model_id, w.N, w.mp, o.opt(max_sqp_iter=1, n_ipm=n_ipm, yref_per_stage=1, reg=w.meta["reg"], reg_e=w.meta["reg_e"]),
            w.W, w.W_e, w.x0[sl], w.yref[sl], w.yref_e[sl], w.params[sl], w.X[sl], w.U[sl])
    threads = host_cores()
    Xo, Uo = o.solve_batch(*args, nthreads=threads)[:2]     # warm (page in, thread pool)
    parity = None
    if gpu_first is not None:
        Xg, Ug = (np.asarray(v[sl], np.float64) for v in gpu_first)
        parity = {"rel_l2_X": float(np.linalg.norm(Xg - Xo) / np.linalg.norm(Xo)),
                  "rel_l2_U": float(np.linalg.norm(Ug - Uo) / np.linalg.norm(Uo)),
                  "against": f"fp64 oracle, first solve of the {sample} problems (parity unpinned: DESIGN.md 2)"}
    reps, t0 = 0, time.perf_counter()
    while True:
        o.solve_batch(*args, nthreads=threads)
        reps += 1
        if time.perf_counter() - t0 > seconds or reps >= 50:
            break
    dt = time.perf_counter() - t0
    n1 = min(sample, 128 if w.model_id != wl.MODEL_WHOLEBODY else 16)
    args1 = args[:6] + tuple(a[:n1] for a in args[6:])
    t1 = time.perf_counter()
    o.solve_batch(*args1, nthreads=1)
    single = (time.perf_counter() - t1) / n1
    return dict(value=sample * reps / dt, unit="solves/s", cores=threads, kind="port",
                sample=f"{sample} of the {w.B} problems x {reps} repeats, fp64 oracle (OpenMP over problems); "
                       f"single-thread {single * 1e3:.2f} ms/solve", parity=parity)


def kernel_sources_sha() -> str:
    """Fingerprint of the HIP sources the library is built from: ties a PMC traffic record to a build."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "iterative_learning_nmpc_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".inc")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def traffic_record(key):
    """PMC-measured HBM bytes per launch (tools/profile.sh writes profiles/traffic.json with the fingerprint of the kernel
    sources it profiled): used only if that record belongs to THIS build of the kernels."""
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tp):
        return None, None
    try:
        rec = json.load(open(tp)).get(key)
        if rec and rec.get("kernel_sources_sha") == kernel_sources_sha():
            return rec["bytes"], (f"{rec['source']} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of {rec['kernel']}, "
                                  f"kernel sources {rec['kernel_sources_sha']})")
        if rec:
            return None, (f"stale: {rec.get('source')} was measured on kernel sources {rec.get('kernel_sources_sha')}, "
                          f"this build is {kernel_sources_sha()}; traffic withheld")
    except Exception:
        pass
    return None, None


class SolveSetup:
    """One batch of one workload resident on the device, with its solver: what a timed step needs."""

    def __init__(self, wholebody, B, N, precision, sqp, ipm, dev, seed, w=None):
        self.wbm, self.B, self.N, self.precision, self.sqp, self.ipm, self.dev = wholebody, B, N, precision, sqp, ipm, dev
        self.w = w if w is not None else (wl.wholebody_trot if wholebody else wl.centroidal_trot)(B=B, N=N, seed=seed)
        w = self.w
        self.s = s = BatchedNmpcSolver(w.model_id, N, B, dev, precision=precision)
        s.set_model_params(w.mp)
        s.set_cost_weights(w.W, w.W_e, w.meta["reg"], w.meta["reg_e"])
        s.set_max_iter(sqp)
        s.set_max_qp_iter(ipm)
        self.t = {k: s.to_device(getattr(w, k)) for k in ("x0", "yref", "yref_e", "params", "X", "U")}
        self.X0, self.U0 = self.t["X"].clone(), self.t["U"].clone()
        self.status = torch.empty(B, dtype=torch.int32, device=dev)
        self.stats = torch.empty(B, 4, dtype=torch.float32, device=dev)

    def reset_guess(self):
        self.t["X"].copy_(self.X0); self.t["U"].copy_(self.U0)

    def solve(self, shift):
        t = self.t
        self.s.solve(t["x0"], t["yref"], t["yref_e"], t["params"], t["X"], t["U"], self.status, self.stats, shift=shift)

    def first_solve(self):
        """the solve of the untouched inputs (kept for the parity figure), then the initial guess again"""
        self.reset_guess()
        self.solve(0)
        out = (self.t["X"].cpu().numpy().reshape(self.B, self.N + 1, -1), self.t["U"].cpu().numpy().reshape(self.B, self.N, -1))
        self.reset_guess()
        return out

    def failed(self):
        return int((self.status == 1).sum().item() + (self.status == 4).sum().item())

    def roofline(self, kernel_ms):
        n_sweeps = self.ipm if self.ipm > 0 else 1
        flops, nbytes = (algorithmic_work_wholebody(self.N, n_sweeps) if self.wbm else
                         algorithmic_work(self.s.nx, self.s.nu, self.s.ng if self.ipm > 0 else 0, self.s.np, self.N, n_sweeps))
        flops *= self.sqp
        tf = flops * self.B / (kernel_ms * 1e-3) / 1e12
        gbs = nbytes * self.B / (kernel_ms * 1e-3) / 1e9
        traffic, source = traffic_record(f"{'wb_' if self.wbm else ''}B{self.B}_ipm{self.ipm}_sqp{self.sqp}_p{self.precision}")
        return {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_source": source,
                # measured HBM-side bytes over the launch as a fraction of the HBM peak (the counters include Infinity-Cache hits)
                "traffic_frac_of_hbm": (traffic / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if traffic else None,
                "traffic_over_algorithmic": (traffic / (nbytes * self.B)) if traffic else None,
                "kernel": ("nmpc_wb_qp_kernel (+ nmpc_wb_linearize_kernel)" if self.wbm else
                           "nmpc_qp_kernel<Centroidal> (+ nmpc_linearize_kernel, 5 % of the solve call)"), "kernel_ms": kernel_ms,
                "flops_per_solve": flops, "bytes_per_solve": nbytes,
                "hbm_algorithmic_GBs": gbs, "hbm_frac": gbs / PEAK_HBM_GBS,
                "binds": ("backward sweeps: instruction issue of one wave per SIMD along the serial stage recursion (154 MFMAs + the "
                          "elimination's dependent chain per stage); forward sweeps: the stage's own dependent chain (58 broadcasts, FMA chains, "
                          "one crossbar move) at loaded memory latency -- 3.4 TB/s of counter traffic over the launch, see "
                          "traffic_over_algorithmic; DESIGN.md 5b (18)" if self.wbm else
                          "instruction issue of one wave per SIMD along the serial stage recursion (MFMA + VALU + LDS of a wave "
                          "do not overlap; profiles/): neither HBM nor the MFMA peak")}


def timed_steps(setup, steps, warmup, dist=None, exchange=None, ramp=0.25):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; returns (wall seconds MAX over ranks,
    mean solve-call milliseconds from HIP events on the launch stream)."""
    def step(timed):
        setup.solve(1)
        if exchange:
            exchange(timed)
    # a fresh box starts at idle clocks: keep the device busy for a quarter of a second before the W warm-up
    # steps, so that the K timed steps measure the steady state whatever W is (setup, not part of W or K)
    # (the solve alone, no exchange: a time-bounded loop runs a different number of passes on every rank, and a collective
    #  inside it would be called a different number of times -- a deadlock that shows once the ranks' speeds differ)
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < ramp:
        for _ in range(20 if not setup.wbm else 2):
            setup.solve(1)
        torch.cuda.synchronize()
    for _ in range(warmup):
        step(False)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()                         # same stream the kernels are launched on
        setup.solve(1)
        ev[i][1].record()
        if exchange:
            exchange(True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=setup.dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))


def wholebody_legs(a, dev, want_cpu):
    """configs[2] and configs[4] at N = 1, on one workload: B whole-body problems (nx 42, nu 30, N 30), a step = warm-start
    shift + solve (1 SQP x 6 IPM).  Returns the two sibling objects of the default line."""
    B, N = a.wb_batch, 30
    out = {}
    w = wl.wholebody_trot(B=B, N=N, seed=0)
    for key, precision in (("wholebody", 0), ("mixed_precision", MIXED_PRECISION_SHIPPED)):
        su = SolveSetup(True, B, N, precision, 1, 6, dev, 0, w=w)
        first = su.first_solve()
        steps, warmup = 8, 2
        elapsed, kernel_ms = timed_steps(su, steps, warmup, ramp=0.1)
        leg = {"workload": f"configs[{2 if precision == 0 else 4}]: batch={B} whole-body 18-DoF quadruped NMPC nx=42 nu=30 N=30, friction pyramid + "
                           "stance constraints, 1 SQP x 6 IPM Riccati sweeps per solve, warm-start shift + solve per step" +
                           ("" if precision == 0 else f"; J^T W J of the Gauss-Newton Hessian on v_mfma_f32_16x16x16_bf16 "
                            f"(precision {precision}: {'bf16' if precision == 1 else 'split bf16 hi+lo' if precision == 2 else 'three-way split bf16 hi+mid+lo'}"
                            " Jacobian), fp32 Riccati"),
               "solves_per_s": B * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup,
               "dtype": "f32" if precision == 0 else "f32 (bf16 J'WJ contraction)", "failed_problems": su.failed(),
               "roofline": su.roofline(kernel_ms)}
        if want_cpu:
            cb = cpu_baseline(w, 6, min(B, 64), first, seconds=4.0 if precision == 0 else 0.0)
            leg["parity"] = cb.pop("parity")
            if precision == 0:
                leg["cpu_baseline"] = cb
        out[key] = leg
        del su
    out["wholebody"]["single_problem_latency"] = single_solve_latency(dev)
    out["wholebody"]["device_rollouts"] = wholebody_rollouts(B, dev)
    return out


def wholebody_rollouts(B, dev):
    """B pushed whole-body rollouts of 2 s (the reference's own problem through `LocomotionMPC.open_loop_device`: 50 replans, the
    first one 15 SQP iterations, prepare -> solve -> advance on the device, no host round trip inside a rollout)."""
    import warnings
    from iterative_learning_nmpc_amd import wholebody as wbk
    from iterative_learning_nmpc_amd.mpc_wholebody import LocomotionMPC
    rng = np.random.default_rng(0)
    q0 = np.zeros((B, 18)); q0[:, 2] = 0.30; q0[:, 6:] = wbk.Q_HOME + rng.normal(0, 0.03, (B, 12))
    v0 = np.zeros((B, 18))
    force = rng.uniform(-1, 1, (B, 3)); force /= np.linalg.norm(force, axis=1, keepdims=True); force *= rng.uniform(50, 70, (B, 1))
    force[0] = 0.0
    push = dict(start=0.2, duration=0.3, force=force)
    times = []
    for it in range(2):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mpc = LocomotionMPC(print_info=False, device=dev, batch=B, n_nodes=30, force_reference="gravity_share")
        mpc.set_command(np.array([0.2, 0.0, 0.0]), 0.0)
        mpc.solver._device_solver()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        S = mpc.open_loop_device(q0, v0, 2.0, push=push, record_sim_steps=False)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        flags = mpc.failed
        del mpc
    el = times[-1]
    return {"rollouts": B, "replans": int(S.shape[1]), "seconds": el, "rollouts_per_s": B / el, "solver_failures": int((flags & 1).ne(0).sum().item()),
            "terminated_early": int((flags >> 8).ne(0).sum().item()), "finite": bool(torch.isfinite(S).all().item())}


def single_solve_latency(dev):
    """B = 1 through the reference-shaped facade, the reference's own horizon (N = 25, mpc_opt.py:11-13): wall time of
    `LocomotionMPC.optimize(q, v)` -- host assembly of the views, six uploads, the solve, four downloads, one
    synchronisation -- for the first solve (15 SQP iterations) and for steady-state replans, next to the reference's budget
    of 40 ms per solve (replanning at 25 Hz, mpc_opt.py:15)."""
    import warnings
    from iterative_learning_nmpc_amd import wholebody as wbk
    from iterative_learning_nmpc_amd.mpc_wholebody import LocomotionMPC
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mpc = LocomotionMPC(print_info=False, device=dev, force_reference="gravity_share")
    mpc.set_command(np.array([0.3, 0.0, 0.0]), 0.0)
    q = np.zeros(18); q[2] = 0.30; q[6:] = wbk.Q_HOME
    v = np.zeros(18)
    mpc.solver._device_solver()                      # handle creation is set-up, not a solve
    mpc.set_convergence_on_first_iter()
    t0 = time.perf_counter()
    q_sol, v_sol, *_ = mpc.optimize(q, v)
    first_ms = (time.perf_counter() - t0) * 1e3
    mpc.first_solve = False
    mpc.sim_step, mpc.current_opt_node = mpc.replanning_steps, 1
    mpc.set_convergence_on_first_iter()
    times = []
    for i in range(30):
        t0 = time.perf_counter()
        q_sol, v_sol, *_ = mpc.optimize(q_sol[1].copy(), v_sol[1].copy())
        times.append((time.perf_counter() - t0) * 1e3)
        mpc.current_opt_node += 1
    times = np.array(times[5:])
    return {"horizon": mpc.config_opt.n_nodes, "first_solve_ms": first_ms, "steady_ms_mean": float(times.mean()), "steady_ms_max": float(times.max()),
            "reference_budget_ms": 40.0, "path": "LocomotionMPC.optimize -> QuadrupedAcadosSolver.init / solve (views -> pinned buffers -> device -> views)"}


def rollout_leg(B, steps, warmup, world, rank, dev, dist, max_attempts=8, fill_batch=2048):
    """BASELINE configs[3] per-GPU slice: B perturbed rollouts + the shared nominal one, each 2 s = 50
    replans (first one a 15-iteration cold start); rollouts that end with a solver failure or an unsafe base state are
    discarded and redone with a new push, as the reference re-rolls an early-terminated rollout
    (DAgger/example/data_collection_pretrain_omini_vc_policy_1direction_perturbed.py:217-247); tracking errors [B, 50]
    of the valid rollouts against the nominal one, one all-gather of errors and validity per learning iteration,
    OOD weights on every rank.  A redo pass over n < fill_batch rollouts rolls fill_batch // n candidates of each (independent
    pushes, the first that runs to the end is kept -- the distribution of the reference's one-by-one redo): every pass costs fifty
    replans of latency however few rollouts it holds, and 2048 problems are what the chip holds at once."""
    from iterative_learning_nmpc_amd.mpc import BatchedLocomotionMPC, sample_pushes
    from iterative_learning_nmpc_amd.parallel import all_gather_tracking_errors, learning_update, ood_threshold
    from iterative_learning_nmpc_amd.solver import tracking_error
    T = 2.0
    x0 = np.zeros((B, 12)); x0[:, 2] = 0.3
    mpc = BatchedLocomotionMPC(B, n_nodes=50, device=dev, footsteps=True)     # Raibert touch-downs + stance anchoring on the device
    times, info = [], None
    for it in range(warmup + steps):
        mpc.reset()
        mpc.set_command(np.array([0.3, 0.0, 0.0]), 0.0)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        # pushes of bc_experimental.yaml:32-35 (50-70 N, random direction); rollout 0 of every rank: the nominal one
        S, _, info = mpc.open_loop_device_valid(x0, T, lambda n, attempt: sample_pushes(n, seed=(1000 * rank + it, attempt), start=0.2, duration=0.3),
                                                nominal=(0,), max_attempts=max_attempts, fill_batch=fill_batch)
        err = tracking_error(S, S[0].contiguous(), with_weights=False)
        valid = (mpc.failed & mpc.invalid_mask).eq(0)
        err_all = all_gather_tracking_errors(err, world * B)
        valid_all = all_gather_tracking_errors(valid.float().unsqueeze(1), world * B).squeeze(1) > 0.5
        ood, weights = learning_update(err_all, threshold=ood_threshold(S.shape[2]), valid=valid_all)      # 19-slot rows: 2.59 (parallel.py)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        if it >= warmup:
            times.append(time.perf_counter() - t0)
    el = float(np.mean(times))
    if dist:
        tt = torch.tensor([el], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    n_replans = S.shape[1]
    flags = mpc.failed
    n_rolled = int(sum(info["attempt_sizes"]))
    fl = algorithmic_work(12, 12, 16, 16, 50, 6)[0] * n_rolled * (n_replans + 14)   # first replan: 15 SQP iterations
    return {
        "workload": f"configs[3] slice: {B} rollouts/GPU x {n_replans} replans, push 50-70 N x 0.3 s, discard-and-redo of failed rollouts "
                    f"(redo passes filled to {fill_batch} candidates), "
                    "tracking error vs nominal + all-gather of [B,50] errors and validity",
        "rollouts_per_s": world * B / el, "ms_per_step": el * 1e3, "steps": steps, "warmup": warmup,
        "solves_per_s": world * n_rolled * n_replans / el,
        "attempts": len(info["attempt_sizes"]), "rollouts_run_per_attempt": info["attempt_sizes"],
        "first_attempt": info["first_attempt"],                       # flag counts of the un-redone batch
        "valid_rollouts": int(valid.sum().item()), "failed_rollouts": int((flags & 1).ne(0).sum().item()),
        "unsafe_state_rollouts": {"roll": int((flags & 2).ne(0).sum().item()), "pitch": int((flags & 4).ne(0).sum().item()),
                                  "height": int((flags & 8).ne(0).sum().item()), "velocity_tracking": int((flags & 16).ne(0).sum().item())},
        "ood_fraction": float(ood[valid_all].float().mean().item()) if bool(valid_all.any()) else None,
        "zero_weight_rollouts": int((weights.sum(dim=1) == 0).sum().item()),
        # the solves dominate a rollout: same per-solve FLOP count as the headline
        "roofline": {"bound": "mfma", "achieved": fl / el / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": fl / el / 1e12 / PEAK_FP32_TFLOPS, "traffic": None,
                     "kernel": "nmpc_qp_kernel<Centroidal> inside nmpc_rollout_batch", "kernel_ms": None},
    }


def rollout_mode(a, world, rank, dev, dist):
    leg = rollout_leg(a.rollouts, a.steps, a.warmup, world, rank, dev, dist)
    if rank == 0:
        roof = leg.pop("roofline")
        print(json.dumps({
            "metric": "ILC rollouts/sec (2 s centroidal rollouts, 50 replans, device-resident)",
            "value": leg.pop("rollouts_per_s"), "unit": "rollouts/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": leg.pop("ms_per_step"), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "config": leg, "roofline": roof, "cpu_baseline": None}), flush=True)


def policy_mode(a, world, rank, dev, dist):
    """Learning update: training steps of the reference's policy network
    (47 -> 3 x 512 -> 12, BatchNorm, L1 loss, Adam; cfgs/iter_locosafedagger.yaml:52-58) on a batch of
    a.policy samples per GPU; every rank trains its own replica on its own shard (the reference trains
    on one device -- no gradient exchange is part of this path)."""
    from iterative_learning_nmpc_amd.policy import DevicePolicy
    B, n_in, n_out, L, H = a.policy, 47, 12, 3, 512
    pol = DevicePolicy(n_in, n_out, L, H, True, batch_max=B, device=dev, seed=rank)
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    X = torch.randn(B, n_in, generator=g).to(dev); Y = torch.randn(B, n_out, generator=g).to(dev)
    for _ in range(a.warmup):
        pol.train_step(X, Y, 1e-3)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        loss = pol.train_step(X, Y, 1e-3)
    e1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([el], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    ms = e0.elapsed_time(e1) / a.steps
    macs = n_in * H + (L - 1) * H * H + H * n_out
    flops = 2.0 * B * (3 * macs - n_in * H)            # forward, weight gradient, data gradient (not for the input layer)
    nbytes = 4.0 * (B * (n_in + 2 * n_out) + 4 * (macs + (L * 3 + 1) * H))   # batch in/out + theta, grad, m, v once each
    cpu = None
    if rank == 0 and not a.no_cpu_baseline:
        from oracle.policy_oracle import PolicyOracle      # the checker, timed as the reported CPU baseline
        o = PolicyOracle(n_in, n_out, L, H, True, np.float32)
        th, rm, rv = (t.cpu().numpy() for t in pol.get_parameters())
        o.theta[:] = th; o.running_mean[:] = rm; o.running_var[:] = rv
        Xh, Yh = X.cpu().numpy(), Y.cpu().numpy()
        o.train_step(Xh, Yh, 1e-3)
        n_rep, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 10.0:
            o.train_step(Xh, Yh, 1e-3); n_rep += 1
        cpu = {"value": n_rep * B / (time.perf_counter() - t1), "unit": "samples/s", "cores": host_cores(), "kind": "port",
               "sample": f"{n_rep} training steps of the same batch, fp32 numpy oracle (BLAS threads as granted)"}
    if rank == 0:
        tf = flops / (ms * 1e-3) / 1e12
        print(json.dumps({
            "metric": "policy training samples/sec (47-3x512-12 MLP, BatchNorm, L1, Adam)",
            "value": world * B * a.steps / el, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"learning update: batch {B}/GPU, one Adam step per step", "final_loss": float(loss.item())},
            "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / PEAK_FP32_TFLOPS, "traffic": None, "kernel": "gemm_kernel (11 of the 25 launches of a step)",
                         "kernel_ms": ms, "flops_per_step": flops, "bytes_per_step": nbytes},
            "cpu_baseline": cpu}), flush=True)


def database_mode(a, world, rank, dev, dist):
    """Dataset side of the learning update: a step = `Database.calc_input_mean_std` over a state
    table of a.database rows x 44 columns per GPU (database.py:208-255; the reference recomputes it after every
    append, cfgs/iter_locosafedagger.yaml:61 sizes the table at 1e7 rows) + one normalised batch of 1024
    (`__getitem__` x 1024).  HBM-bound: two passes over the table."""
    from iterative_learning_nmpc_amd.database import DeviceDatabase
    rows, n_state, batch = a.database, 44, 1024
    db = DeviceDatabase(rows, n_state=n_state, n_action=12, device=dev)
    g = torch.Generator(device=dev).manual_seed(5 + rank)
    chunk = 1 << 20
    for r0 in range(0, rows, chunk):                      # fill the tables in place (append = the same kernels, timed below)
        n = min(chunk, rows - r0)
        db.tables["states"][r0:r0 + n] = torch.randn(n, n_state, generator=g, device=dev) * 2.0 + 1.0
    db.length, db.has["vc_goals"] = rows, True
    idx = torch.randint(0, rows, (batch,), generator=g, device=dev, dtype=torch.int32)

    def step():
        db.calc_input_mean_std()
        return db.batch(idx)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        x, y = step()
    e1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([el], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    ms = e0.elapsed_time(e1) / a.steps
    nbytes = 2.0 * rows * n_state * 4                      # the table is read once per pass; everything else is KB
    cpu = None
    if rank == 0 and not a.no_cpu_baseline:
        from oracle.database_oracle import DatabaseOracle  # the checker, timed as the reported CPU baseline
        sample = min(rows, 2_000_000)
        o = DatabaseOracle(sample)
        o.rows["states"] = db.tables["states"][:sample].cpu().numpy().astype(np.float64); o.length = sample
        o.calc_input_mean_std()
        n_rep, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 10.0:
            o.calc_input_mean_std(); n_rep += 1
        cpu = {"value": n_rep * sample / (time.perf_counter() - t1), "unit": "rows/s", "cores": 1, "kind": "port",
               "sample": f"{n_rep} x mean/std over the first {sample} rows, float64 numpy oracle"}
    if rank == 0:
        gbs = nbytes / (ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "database rows/sec through mean/std + batch assembly (44-column state table)",
            "value": world * rows * a.steps / el, "unit": "rows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"database statistics: {rows} rows x {n_state} fp32 per GPU, batch {batch}"},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                         "traffic": None, "kernel": "colstat_partial_kernel (2 passes)", "kernel_ms": ms,
                         "bytes_per_step": nbytes},
            "cpu_baseline": cpu}), flush=True)


def torque_mode(a, world, rank, dev, dist):
    """Torque layer: a step = id_torques (dynamics.py:136-163) + PD law (mpc.py:592-599) for
    a.torques robots per GPU on a declared 18-DoF quadruped tree (the reference's URDF is not in the image)."""
    from iterative_learning_nmpc_amd.torque import BatchedTorqueLayer
    tree = wl.quadruped_tree()
    L = BatchedTorqueLayer(device=dev, **tree)
    n, nu = len(tree["parent"]), tree["n_actuated"]
    B = a.torques
    g = torch.Generator(device=dev).manual_seed(9 + rank)
    q, v, acc, qp, vp = (torch.rand(B, n, generator=g, device=dev) * 2 - 1 for _ in range(5))
    f = torch.rand(B, 4, 3, generator=g, device=dev) * 60.0

    def step():
        return L.compute_pd_torques(q, v, L.id_torques(q, v, acc, f), qp, vp, 44.0, 5.0)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        tau = step()
    e1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([el], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    ms = e0.elapsed_time(e1) / a.steps
    nbytes = 4.0 * B * (5 * n + 12 + 3 * nu)           # q, v, a, q_plan, v_plan, f in; tau_ff out + in, tau out
    cpu = None
    if rank == 0 and not a.no_cpu_baseline:
        from oracle import torque_oracle as to             # the checker, timed as the reported CPU baseline
        m = to.TreeModel.from_arrays(tree)
        n_s = min(B, 256)
        qh, vh, ah, fh = (t[:n_s].cpu().numpy().astype(np.float64) for t in (q, v, acc, f))
        n_rep, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 10.0:
            to.id_torques_batch(m, qh, vh, ah, fh); n_rep += 1
        cpu = {"value": n_rep * n_s / (time.perf_counter() - t1), "unit": "robots/s", "cores": 1, "kind": "port",
               "sample": f"{n_rep} x id_torques of the first {n_s} robots, float64 numpy oracle (pure-Python recursion)"}
    if rank == 0:
        gbs = nbytes / (ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "torque layer robots/sec (inverse dynamics with contact forces + PD, 18-DoF tree)",
            "value": world * B * a.steps / el, "unit": "robots/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"torque layer: {B} robots/GPU, declared quadruped tree", "tau_abs_mean": float(tau.abs().mean().item())},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                         "traffic": None, "kernel": "id_torques_kernel (serial recursion per robot: latency-bound)", "kernel_ms": ms,
                         "bytes_per_step": nbytes},
            "cpu_baseline": cpu}), flush=True)


def main():
    a = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if torch.cuda.device_count() == 0:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path (the CPU oracle is only the reported baseline)")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # NMPC_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (the
        # ranks then share the devices round robin and the timing reduction runs on CPU tensors); default RCCL
        if os.environ.get("NMPC_BENCH_BACKEND", "nccl") == "gloo":
            local = local % torch.cuda.device_count()
            dist.init_process_group("gloo")
        else:
            if torch.cuda.device_count() < world:
                raise SystemExit(f"--gpus {world} needs {world} GPUs on this node, found {torch.cuda.device_count()} "
                                 "(NMPC_BENCH_BACKEND=gloo rehearses the multi-rank path on fewer)")
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: run `python bench.py --gpus {a.gpus}` (it starts its own ranks) "
                         "or launch with torch.distributed.run --nproc-per-node equal to --gpus")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    if a.rollouts or a.policy or a.database or a.torques:
        (rollout_mode if a.rollouts else policy_mode if a.policy else database_mode if a.database else torque_mode)(a, world, rank, dev, dist)
        if dist:
            dist.destroy_process_group()
        return

    wbm = a.workload == "wholebody"
    B, N = (a.batch or (8192 if wbm else 1024)), (30 if wbm else 50)
    if wbm and a.steps == 200 and a.warmup == 20:      # defaults sized for the 0.5 ms centroidal step
        a.steps, a.warmup = 20, 3
    su = SolveSetup(wbm, B, N, a.precision, a.sqp, a.ipm, dev, seed=1000 * rank)
    s, w, t = su.s, su.w, su.t

    # N > 1: the one exchange step of the path rides in every timed step -- tracking error of the
    # solved trajectories against problem 0 of the rank ("nominal"), then ONE all-gather of the [B, N+1] errors
    # over the ranks (RCCL over xGMI).  At N = 1 there is no exchange and the step is the solve alone.
    from iterative_learning_nmpc_amd.solver import tracking_error
    from iterative_learning_nmpc_amd.parallel import all_gather_tracking_errors
    ag_ev = []

    def exchange(timed):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        err = tracking_error(t["X"], t["X"][0].contiguous(), with_weights=False)
        e0.record()
        err_all = all_gather_tracking_errors(err, world * B)
        e1.record()
        if timed:
            ag_ev.append((e0, e1, err_all.shape))

    first = su.first_solve()      # the first solve of the untouched inputs, kept for the parity figure of the cpu_baseline leg
    elapsed, kernel_ms = timed_steps(su, a.steps, a.warmup, dist, exchange if world > 1 else None)
    bad = su.failed()
    default_run = (not wbm) and a.sqp == 1 and a.precision == 0 and not a.no_cold_start

    # the cold-start variant of the same workload -- the reference's first solve runs 15 SQP
    # iterations (mpc.py:464-473) from the standing initial guess; reported next to the headline, not as it
    cold = None
    if a.sqp == 1 and a.precision == 0 and not a.no_cold_start:
        s.set_max_iter(15)
        n_cold = max(2, min(10, a.steps))
        for i in range(n_cold + 1):
            if i == 1:
                torch.cuda.synchronize(); tc = time.perf_counter()
            su.reset_guess()
            su.solve(0)
        torch.cuda.synchronize()
        ms_cold = (time.perf_counter() - tc) / n_cold * 1e3
        cold = {"sqp_iterations": 15, "ms_per_solve_call": ms_cold, "solves_per_s_per_gpu": B / (ms_cold * 1e-3),
                "failed_problems": su.failed()}
        s.set_max_iter(a.sqp)

    # the same step at eight times the batch (the per-GPU size of configs[3]): beyond one problem per SIMD the QP kernel
    # runs its two-waves-per-SIMD variant; inputs are the batch above, repeated.  Reported next to the
    # headline, not as it.
    large = None
    if world == 1 and default_run and B == 1024:
        BL = 8 * B
        wL = wl.Workload(w.model_id, N, w.mp, w.W, w.W_e, *(np.concatenate([getattr(w, k)] * 8) for k in ("x0", "yref", "yref_e", "params", "X", "U")),
                         meta=w.meta)
        sl = SolveSetup(False, BL, N, 0, a.sqp, a.ipm, dev, 0, w=wL)
        n_large = max(2, min(20, a.steps))
        el_l, ms_l = timed_steps(sl, n_large, 3, ramp=0.0)
        roof_l = sl.roofline(ms_l)
        large = {"batch": BL, "ms_per_step": el_l / n_large * 1e3, "solves_per_s_per_gpu": BL * n_large / el_l, "failed_problems": sl.failed(),
                 "kernel_ms": ms_l, "roofline_frac": roof_l["frac"], "traffic": roof_l["traffic"], "traffic_source": roof_l["traffic_source"],
                 "traffic_frac_of_hbm": roof_l["traffic_frac_of_hbm"], "traffic_over_algorithmic": roof_l["traffic_over_algorithmic"]}
        del sl, wL

    out = None
    if rank == 0:
        out = {
            "metric": "MPC solves/sec (horizon-30 whole-body, batch)" if wbm else "MPC solves/sec (horizon-50 centroidal, batch)",
            "value": world * B * a.steps / elapsed, "unit": "solves/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if a.precision == 0 else "f32 (bf16 J'WJ contraction)", "data": "synthetic",
            "config": {"workload": (f"configs[{2 if a.precision == 0 else 4}]: batch={B}/GPU whole-body 18-DoF quadruped NMPC nx=42 nu=30 N=30 "
                                    f"{'fp32' if a.precision == 0 else f'mixed precision {a.precision} (bf16 Jacobian, MFMA J^T W J, fp32 Riccati)'}, friction-pyramid + "
                                    f"stance constraints, {a.sqp} SQP x {a.ipm} IPM Riccati sweeps per solve, warm-start shift + solve per step") if wbm else
                                   (f"configs[{1 if a.precision == 0 else 4}]: batch={B}/GPU centroidal quadruped NMPC nx=12 nu=12 N=50 "
                                    f"{'fp32' if a.precision == 0 else 'mixed precision (bf16 MFMA barrier product, fp32 Riccati)'}, "
                                    f"{a.sqp} SQP x {a.ipm} IPM Riccati sweeps per solve, warm-start shift + solve per step"),
                       "global_batch": world * B, "horizon": N,
                       "parallelism": f"dp{world} (independent problems; " + ("no collective)" if world == 1 else
                                      f"one all-gather of the [{B},{N + 1}] tracking errors per step, backend {dist.get_backend()})")},
            "roofline": su.roofline(kernel_ms),
            "failed_problems": bad, "cold_start": cold, "large_batch": large,
        }
        if world > 1 and ag_ev:
            out["allgather"] = {"ms": float(np.mean([e0.elapsed_time(e1) for e0, e1, _ in ag_ev])), "backend": dist.get_backend(),
                                "ranks": world, "gathered_shape": list(ag_ev[0][2]), "per_step": 1}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, a.ipm, min(B, 256 if wbm else 1024), first if (a.sqp == 1 and a.precision == 0) else None)
    del su
    # the other GPU configurations of BASELINE.json, measured in the same run (N = 1, default line only)
    if world == 1 and default_run and not a.headline_only and not a.batch:
        legs = wholebody_legs(a, dev, not a.no_cpu_baseline)
        out.update(legs)
        out["rollouts"] = rollout_leg(a.rollout_batch, 2, 1, 1, 0, dev, None)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
