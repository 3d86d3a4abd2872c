#!/usr/bin/env python3
"""Headline benchmark: iLQR iterations/s as (batch x horizon) steps/s, quadrotor n_x=12 n_u=4 N=50, batch 4096 per GPU
(BASELINE.json configs[2]; configs[3] = the same per GPU over 8 GPUs, weak scaling).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one iLQR iteration of the whole batch exactly as the reference performs it
(quattro_ilqr_tf.py:428-451): nominal rollout + cost (simulate), linearisation of all N steps, the Riccati-like
backward sweep, and the 6-alpha line search with accept/commit.  Every step starts from the same synthetic nominal
(x0 = x_ref + U(-1,1)*[.5,.5,.01,0,0,0,.2,.2,.5,0,0,0], u = hover + 0.1 N(0,1), seed 1234 + rank), so the
work per step is fixed.  Inputs are resident in HBM before the timed region.  With N > 1 ranks each rank owns its
own 4096 trajectories (no data-path collective) and the run ends with the one exchange the north star names: an RCCL
all-gather of the (K, k) gain stacks, inside the timed region.

Rank 0 prints ONE JSON line: throughput, the roofline of the dominant kernel (the sweep; HIP-event durations measured
inside the timed region) and, at N = 1, the CPU baseline (the oracle's reference-style fp64 finite-difference iLQR,
`oracle/ilqr.py`, on a bounded sample, fanned out over the host cores like the reference's own data collection).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HORIZON, BATCH_PER_GPU, NX, NU = 50, 4096, 12, 4
LADDER_WINDOWS = (40, 30, 20, 10)        # + W = 1 (hybrid_config5): the reference's published bars, BASELINE.md section 1
HBM_PEAK_GBS = 8000.0                                   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SWEEP_BYTES_PER_STEP = 4 * (2 * NX * NX + 2 * NX * NU + NU * NU + NX + NU) + 4 * (NU * NX + NU)     # 1664 + 208 = 1872
SWEEP_BYTES_PER_TRAJ = HORIZON * SWEEP_BYTES_PER_STEP + 4 * (NX + NX * NX)                        # + terminal V_x, V_xx


def synthetic_batch(B, rank):
    rng = np.random.default_rng(1234 + rank)
    x_ref = np.zeros(NX)
    x_ref[2] = 0.5
    spread = np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    x0 = x_ref + rng.uniform(-1.0, 1.0, (B, NX)) * spread
    u0 = 2.4525 + 0.1 * rng.standard_normal((B, HORIZON, NU))
    return x0, u0


# --------------------------------------------------------------------------------------------- CPU baseline
def host_cores():
    """Cores this process can really use: min(affinity mask, cgroup CPU quota).  The GPU box shows 64 CPUs in the affinity
    mask under a 16-core cgroup quota; a Pool sized by the mask is 4x oversubscribed and its per-core figure is wrong."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:                                                   # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:                                               # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                q = float(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                per = float(fh.read())
            if q > 0:
                quota = q / per
        except Exception:
            quota = None
    used = aff if quota is None else max(1, min(aff, int(quota + 1e-9)))
    return {"cores_affinity": aff, "cores_quota": quota, "cores_used": max(1, min(used, 64))}


_DEADLINE = None


def _cpu_init(deadline):
    global _DEADLINE
    _DEADLINE = deadline
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    from oracle import ilqr, models, transformer  # noqa: F401  (import cost stays outside the timed map)


def _run_iterations(one_iteration, iters):
    """Up to `iters` iterations of one trajectory; stops early (between iterations) once the shared wall budget is spent."""
    done = 0
    for _ in range(iters):
        if _DEADLINE is not None and time.time() >= _DEADLINE:
            break
        one_iteration()
        done += 1
    return done


def _cpu_worker(args):
    """One trajectory, `iters` iterations of the oracle's FD iLQR (fp64, eps=1e-5, inv(Q_uu+1e-6 I), 6-alpha search)."""
    seed, iters = args
    from oracle import ilqr as o_ilqr
    from oracle import models as o_models
    spec = o_models.quadrotor_spec()
    rng = np.random.default_rng(seed)
    x0 = spec.x_ref + rng.uniform(-1.0, 1.0, NX) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    st = {"u": [2.4525 + 0.1 * rng.standard_normal(NU) for _ in range(HORIZON)]}

    def one():                      # tol < 0: never "converged", so every call is exactly one iteration like the GPU leg
        st["u"], _, _ = o_ilqr.optimize(spec.f, spec.L, spec.Lf, x0, st["u"], HORIZON, max_iter=1, tol=-1.0, keep_logs=True)
    return _run_iterations(one, iters)


def _pool_run(worker, tasks, cores, budget_s):
    import multiprocessing as mp
    ctx = mp.get_context("fork")                       # forked BEFORE this process touches the GPU
    t0 = time.time()
    with ctx.Pool(cores, initializer=_cpu_init, initargs=(None,)) as warm:     # page the imports in once
        warm.map(int, range(cores))
    t0 = time.time()
    with ctx.Pool(cores, initializer=_cpu_init, initargs=(t0 + budget_s,)) as pool:
        t0 = time.time()
        done = pool.map(worker, tasks, chunksize=1)
        wall = time.time() - t0
    return done, wall


def cpu_baseline(traj_per_core=8, iters=10, budget_s=24.0):
    """S = 8 x cores trajectories x 10 iterations of the oracle's reference-style iLQR, fanned out with
    multiprocessing.Pool(cores) like the reference's data collection (training_data_collection.py:298-305) — capped by a
    wall budget so the default bench run stays within minutes: workers stop between iterations once it is spent, and
    only completed iterations are counted."""
    hc = host_cores()
    cores = hc["cores_used"]
    S = traj_per_core * cores
    done, wall = _pool_run(_cpu_worker, [(9000 + i, iters) for i in range(S)], cores, budget_s)
    steps = sum(done) * HORIZON
    # BASELINE configs[0] — cart-pole N = 30, ONE trajectory, one optimize() call of the reference
    # algorithm (single core), the latency the reference's users see
    from oracle import ilqr as o_ilqr
    from oracle import models as o_models
    spec = o_models.cartpole_spec()
    x0c = np.array([0.0, 0.0, 0.1, 0.0])                   # cartpole_sim.py:208
    t1 = time.time()
    _, _, logs = o_ilqr.optimize(spec.f, spec.L, spec.Lf, x0c, [np.zeros(1) for _ in range(30)], 30, max_iter=100, tol=1e-1,
                                 keep_logs=True)
    c1 = {"ms": 1e3 * (time.time() - t1), "iterations": len(logs)}
    out = {"value": steps / wall, "unit": "steps/s", "cores": cores, "kind": "port", "config1_cartpole_N30_B1": c1,
           "sample": f"{S} quadrotor N=50 trajectories x {iters} iLQR iterations planned (SURVEY 8d: 8 x cores x 10), "
                     f"{sum(done)} iterations completed within the {budget_s:.0f} s wall budget on "
                     f"{sum(1 for d in done if d > 0)} trajectories; oracle/ilqr.py (fp64 finite differences, reference "
                     f"algorithm), multiprocessing.Pool({cores}), wall {wall:.1f} s",
           "per_core": steps / wall / cores}
    out.update(hc)
    return out


def _cpu_worker_hybrid(args):
    """One trajectory, `iters` hybrid iterations of the oracle (configs[4]): FD tail step + NumPy fp32 transformer."""
    seed, iters, wpath, window = args
    from oracle import ilqr as o_ilqr
    from oracle import models as o_models
    from oracle import transformer as o_tf
    z = np.load(wpath)
    W = {k: z[k] for k in z.files if not k.startswith("norm.")}
    norm = {k[5:]: z[k] for k in z.files if k.startswith("norm.")}
    spec = o_models.quadrotor_spec()
    rng = np.random.default_rng(seed)
    x0 = spec.x_ref + rng.uniform(-1.0, 1.0, NX) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    st = {"u": [2.4525 + 0.1 * rng.standard_normal(NU) for _ in range(HORIZON)]}
    offset = np.zeros(NX); offset[2] = 0.5
    predict = lambda xe, pr: o_tf.predict(W, norm, xe, pr, 4, window, dtype=np.float32)

    def one():
        st["u"], _, _ = o_ilqr.optimize(spec.f, spec.L, spec.Lf, x0, st["u"], HORIZON, x_ref=spec.x_ref, max_iter=1,
                                        tol=-1.0, tf_predict=predict, tf_window=window, state_offset=offset, keep_logs=True)
    return _run_iterations(one, iters)


def cpu_baseline_hybrid(traj_per_core=2, iters=30, budget_s=10.0, window=1):
    """configs[4] on the host: the oracle's hybrid iteration with a NumPy fp32 evaluation of the same random-init
    transformer ('for config 5 also time the CPU fp32 batched transformer').  window = tf_window: the last
    `window` steps by finite differences + sweep, the first N - window from the predictor (the published ladder)."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "quattro-transformer-ilqr_amd"))
    from quattro_ilqr_amd import TransformerILQR
    tf = TransformerILQR.random_init(NX, NU * (1 + NX), prompt_len=window, target_len=HORIZON - window, d_model=128, nhead=4,
                                     num_decoder_layers=3, dim_feedforward=512, max_seq_len=110, device="cpu")
    wpath = os.path.join(tempfile.mkdtemp(), "w.npz")
    np.savez(wpath, **tf._w, **{"norm." + k: v for k, v in tf._norm.items()})
    hc = host_cores()
    cores = hc["cores_used"]
    S = traj_per_core * cores
    done, wall = _pool_run(_cpu_worker_hybrid, [(9000 + i, iters, wpath, window) for i in range(S)], cores, budget_s)
    steps = sum(done) * HORIZON
    out = {"value": steps / wall, "unit": "steps/s", "cores": cores, "kind": "port",
           "ms_per_iteration_one_core": 1e3 * wall * cores / max(1, sum(done)),
           "sample": f"{S} quadrotor N=50 trajectories x {iters} hybrid iterations planned, {sum(done)} completed within the "
                     f"{budget_s:.0f} s wall budget; oracle/ilqr.py + oracle/transformer.py (fp64 finite differences on the "
                     f"{window}-step tail, NumPy fp32 transformer L=101), multiprocessing.Pool({cores}), wall {wall:.1f} s",
           "per_core": steps / wall / cores}
    out.update(hc)
    return out


def cpu_latency_b1(budget_iters=2):
    """The reference's own use case and its only published metric (figures/quadrotor_result.png, figures/cartpole_result.png,
    README.md:29-33): ONE trajectory, wall time per iLQR iteration, for pure iLQR and the iLQR(W) + TF(N - W) ladder — the
    oracle (the reference's algorithm) on ONE host core, a bounded number of iterations per rung.  Scenario: the README's
    quadrotor start (roll = 0.1 rad, quadrotor_sim.py:250), cold start u = 0 (quadrotor_mpc.py:49), N = 50, Euler."""
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    from oracle import ilqr as o_ilqr
    from oracle import models as o_models
    from oracle import transformer as o_tf
    sys.path.insert(0, os.path.join(ROOT, "quattro-transformer-ilqr_amd"))
    from quattro_ilqr_amd import TransformerILQR
    spec = o_models.quadrotor_spec()
    x0 = spec.x_ref.copy()
    x0[6] = 0.1
    offset = np.zeros(NX); offset[2] = 0.5
    out = {}

    def run(tf_predict, window, iters):
        u = [np.zeros(NU) for _ in range(HORIZON)]
        t0 = time.time()
        kw = {} if tf_predict is None else dict(x_ref=spec.x_ref, tf_predict=tf_predict, tf_window=window, state_offset=offset)
        _, _, logs = o_ilqr.optimize(spec.f, spec.L, spec.Lf, x0, u, HORIZON, max_iter=iters, tol=-1.0, keep_logs=True, **kw)
        return 1e3 * (time.time() - t0) / max(1, len(logs)), len(logs)
    ms, it = run(None, HORIZON, budget_iters)
    out["pure"] = {"ms_per_iteration": ms, "iterations_timed": it}
    for w in LADDER_WINDOWS + (1,):
        if w == 1:      # the shipped checkpoint (P = 1, T = 49), as plain arrays
            z = np.load(os.path.join(ROOT, "tests", "golden", "tf_weights_quadrotor.npz"), allow_pickle=False)
            W = {k: z[k].astype(np.float32) for k in z.files if not k.startswith(("norm.", "hp."))}
            norm = {k[5:]: z[k].astype(np.float64) for k in z.files if k.startswith("norm.")}
        else:
            tf = TransformerILQR.random_init(NX, NU * (1 + NX), prompt_len=w, target_len=HORIZON - w, d_model=128, nhead=4,
                                             num_decoder_layers=3, dim_feedforward=512, max_seq_len=110, device="cpu")
            W, norm = tf._w, tf._norm
        predict = (lambda W_, norm_, w_: (lambda xe, pr: o_tf.predict(W_, norm_, xe, pr, 4, w_, dtype=np.float32)))(W, norm, w)
        ms, it = run(predict, w, budget_iters if w > 1 else 2 * budget_iters)
        out[str(w)] = {"tf_window": w, "ms_per_iteration": ms, "iterations_timed": it,
                       "weights": "shipped checkpoint" if w == 1 else "random-init"}
    return {"quadrotor_N50": out, "cores": 1, "kind": "port",
            "sample": f"oracle/ilqr.py (+ oracle/transformer.py, NumPy fp32) on one core, {budget_iters} iterations per rung "
                      "from the README start (roll 0.1 rad, u = 0)"}


def _cpu_worker_cartpole(args):
    """BASELINE configs[1] on the host: one cart-pole trajectory, N = 50, `iters` iterations of the oracle."""
    seed, iters = args
    from oracle import ilqr as o_ilqr
    from oracle import models as o_models
    spec = o_models.cartpole_spec()
    rng = np.random.default_rng(seed)
    x0 = np.array([rng.uniform(-0.5, 0.5), 0.0, rng.uniform(-0.5, 0.5), 0.0])
    st = {"u": [np.zeros(1) for _ in range(HORIZON)]}

    def one():
        st["u"], _, _ = o_ilqr.optimize(spec.f, spec.L, spec.Lf, x0, st["u"], HORIZON, max_iter=1, tol=-1.0, keep_logs=True)
    return _run_iterations(one, iters)


def cpu_baseline_cartpole(traj_per_core=8, iters=10, budget_s=8.0):
    hc = host_cores()
    cores = hc["cores_used"]
    S = traj_per_core * cores
    done, wall = _pool_run(_cpu_worker_cartpole, [(7000 + i, iters) for i in range(S)], cores, budget_s)
    steps = sum(done) * HORIZON
    out = {"value": steps / wall, "unit": "steps/s", "cores": cores, "kind": "port",
           "sample": f"{S} cart-pole N=50 trajectories x {iters} iterations planned, {sum(done)} completed within the "
                     f"{budget_s:.0f} s wall budget; oracle/ilqr.py, multiprocessing.Pool({cores}), wall {wall:.1f} s",
           "per_core": steps / wall / cores}
    out.update(hc)
    return out


# --------------------------------------------------------------------------------------------- GPU leg
TF_FLOPS_PER_TRAJ = 135.64e6      # L = 101, d = 128, ff = 512, 3 layers, full L x L attention counted
TF_EXECUTED_FLOPS_PER_TRAJ = 169.0e6   # what the kernel's MFMAs execute: 128 token slots, causal tiles only
MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16


def synthetic_cartpole(B, rank):
    rng = np.random.default_rng(1234 + rank)
    x0 = np.zeros((B, 4))
    x0[:, 0] = rng.uniform(-0.5, 0.5, B)
    x0[:, 2] = rng.uniform(-0.5, 0.5, B)
    return x0, np.zeros((B, HORIZON, 1))


class Workload:
    """One iLQR iteration of a batch as separate C-ABI calls with a HIP event between them (events are recorded on
    torch's current stream, the stream every kernel is launched on), so per-kernel durations are measured live inside
    the timed region.  The product path (QuattroILQR.iterate) issues the same launches from one C call."""

    def __init__(self, torch, ops, solver, model, x0, u0, tf=None):
        self.torch, self.ops, self.solver, self.model, self.tf = torch, ops, solver, model, tf
        self.x0, self.u0 = x0, u0
        self.hybrid = tf is not None
        self.fused = ops.model_fuses_sweep(model)
        self._set_names()
        self.n_timed = 0
        solver._alloc(x0.shape[0])
        self.scratch = torch.empty((ops.linesearch_scratch_bytes(model, x0.shape[0], solver.horizon),),
                                   dtype=torch.uint8, device=x0.device)
        if self.hybrid:
            xs = np.asarray(model.x_ref, dtype=np.float64) - solver.state_offset
            tf.shifted_mean(xs, out=solver._tf_mean)
        solver.u.copy_(u0)
        solver._x0.copy_(x0)
        solver._ints.copy_(solver._ints_init)
        self.restart = solver.restart_block.clone()

    EVENT_EVERY = 4

    def _set_names(self):
        lin = () if self.fused else ("linearize",)
        self.names = (("simulate",) + lin + (("sweep", "transformer", "linesearch") if self.hybrid
                                             else ("sweep", "linesearch")))
        self.ev = {k: [] for k in self.names}

    def step(self, timed):
        """One iteration.  HIP events bracket the kernels on every EVENT_EVERY-th timed step only: an event between two
        kernels is a few microseconds of bubble, which the other steps of the timed region do not pay."""
        torch, ops, s, md = self.torch, self.ops, self.solver, self.model
        marks = []
        self.n_timed += 1 if timed else 0
        ev_on = timed and (self.n_timed % self.EVENT_EVERY == 1 or self.EVENT_EVERY == 1)

        def mark():
            if ev_on:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append(e)
        # the same problem again: ONE device copy restores [u | x0 | per-solve state] from a template (rounds 1-3: a copy
        # and a fill, 9.3 us of scaffolding per step; x and the cost are recomputed by simulate)
        s.restart_block.copy_(self.restart)
        mark()
        ops.simulate(md, self.x0, s.u, x=s.x, cost=s.cost)
        mark()
        if not self.fused:      # (fused: the sweep's own wave linearises its trajectory; there is no such kernel)
            ops.linearize(md, s.x, s.u, t_start=s.t_start, layout=s.layout, rec=s.rec, VxN=s.VxN, VxxN=s.VxxN)
            mark()
        if self.fused:
            # hybrid: the swept tail goes straight into rows N - W .. N - 1 of the full stacks (in_place), where the predictor
            # reads its prompt and the line search its gains: two launches, no packing / assembly kernels (rounds 1-3: a
            # concatenation and two strided copies per iteration)
            ops.linearize_sweep(md, s.x, s.u, s.t_start, s.reg, K=s.K, k=s.k, status=s.status, active=s.active,
                                scratch=s._sweep_scratch, in_place=self.hybrid)
        else:
            ops.riccati_sweep(s.rec, s.VxN, s.VxxN, md.n, md.m, s.layout, s.reg, K=s.K, k=s.k, status=s.status,
                              active=s.active)
        mark()
        if self.hybrid:
            # x_err = x - x_ref + offset is formed by the kernel (shifted normalisation mean); prediction unpacked into K, k
            self.tf.predict_gains(s.x, None, s.K, s.k, s.active, x_mean=s._tf_mean)
            mark()
        ops.linesearch(md, s.x, s.u, s.K, s.k, s.cost, s.tol, s.alphas, alpha_idx=s.alpha_idx, active=s.active,
                       iters=s.iters, scratch=self.scratch)
        mark()
        if ev_on:
            for i, name in enumerate(self.names):
                self.ev[name].append((marks[i], marks[i + 1]))

    def kernel_ms(self):
        return {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in self.ev.items() if v}


def timed_steps(torch, fn, steps, warmup, barrier):
    for _ in range(warmup):
        fn(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn(True)
    host_issue = time.perf_counter() - t0
    return t0, host_issue


def latest_pmc_traffic(kernel_prefix, pattern="*_pmc_hbm.json"):
    """Per-launch HBM bytes of a kernel from the latest committed counter pass of THIS bench command (rocprofv3 cannot run
    inside this process): scripts/gpu_pmc.sh -> profiles/*_pmc_hbm.json, separate FETCH_SIZE / WRITE_SIZE passes, gfx950
    x2 FETCH_SIZE correction applied."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))[::-1]:
        try:
            with open(path) as fh:
                kern = json.load(fh)["kernels"]
            v = next(v for k, v in kern.items() if k.startswith(kernel_prefix))
            return float(v["hbm_bytes_corrected"]), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def self_launch(n, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: THIS process becomes the launcher.  It touches no
    GPU (no torch import at all), starts one child per rank — the same script and arguments with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, exactly what `python -m torch.distributed.run --nproc-per-node N` would
    give them — relays rank 0's standard output (the one JSON line) and returns the worst exit code.  Nothing is
    re-executed in a process that has initialised the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
    procs = []
    out0_file = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), QT_BENCH_LAUNCHED_BY="bench.py")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this host
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0_file if r == 0 else subprocess.DEVNULL))
    # wait for every rank; a rank that dies takes the job down (its peers would otherwise sit in the rendezvous or in a
    # collective until the backend's timeout): the survivors get a few seconds, then are killed by their exact PIDs
    failed_at = None
    while any(pr.poll() is None for pr in procs):
        if failed_at is None and any(pr.poll() not in (None, 0) for pr in procs):
            failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > 5.0:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        time.sleep(0.05)
    rcs = [pr.wait() for pr in procs]
    out0_file.seek(0)
    out0 = out0_file.read()
    out0_file.close()
    for line in (out0 or "").splitlines():                      # ONE JSON line on stdout; library chatter (gloo) to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    if bad:
        sys.stderr.write(f"bench.py launcher: exit codes of the {n} ranks: {rcs}\n")
    return (bad[0] if 0 < bad[0] < 256 else 1) if bad else 0


def expected_gather_ms(world, bytes_per_rank):
    """The budget of DESIGN section 6 for the one all-gather of [K | k], so that the first real multi-GPU run interprets itself:
    every GPU receives (world - 1) shards; on a fully connected xGMI node a direct all-gather is bound per LINK (one peer's
    shard per link) — `low` at the 153 GB/s the hardware guide quotes per link, `high` at RCCL's achieved bus bandwidth on this
    class of node (~300 GB/s aggregate receive) — MI355X_MICROARCH.md, xGMI section."""
    if world <= 1:
        return None
    recv = (world - 1) * bytes_per_rank
    return {"low": 1e3 * bytes_per_rank / 153e9, "high": 1e3 * recv / 300e9, "bytes_received_per_rank": recv,
            "assumes": "per-link 153 GB/s (low) ... ~300 GB/s achieved aggregate receive bandwidth (high)"}


def dry_rehearsal(args, rank, world, torch, dist):
    """QT_BENCH_REHEARSAL=dry: everything of the N > 1 path that is not a kernel — rendezvous (gloo), barriers, the gather
    of the [K | k] buffers (CPU tensors of the configured shape) through parallel.GainGather, max-over-ranks timing and
    the JSON line — in a container without a GPU.  Every iLQR step is a no-op; the numbers mean nothing."""
    from quattro_ilqr_amd import parallel          # (imports torch only: the HIP library is not loaded by this module)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B, N = args.batch, HORIZON
    gains = torch.full((B * N * NU * (NX + 1),), float(rank), dtype=torch.float32)
    gg = parallel.GainGather(B, N, NU, NX, torch.float32, "cpu")
    gg(gains)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    tg = time.perf_counter()
    K_all, k_all = gg(gains)
    gather_ms = 1e3 * (time.perf_counter() - tg)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ok = all(bool((K_all[r] == r).all()) and bool((k_all[r] == r).all()) for r in range(world))
    per_rank_ms = [1e3 * elapsed / args.steps]
    per_rank_gather = [gather_ms]
    if world > 1:
        mine = torch.tensor([elapsed, gather_ms], dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = torch.stack(every).numpy()
        elapsed = float(every[:, 0].max())                                   # the contract: MAX over ranks
        per_rank_ms = [1e3 * float(v) / args.steps for v in every[:, 0]]
        per_rank_gather = [float(v) for v in every[:, 1]]
    if rank == 0:
        comm = {"gather_ms": max(per_rank_gather), "gather_ms_per_rank": per_rank_gather, "ms_per_step_per_rank": per_rank_ms,
                "gather_bytes_received_per_rank": gg.bytes_received_per_rank, "collectives_per_gather": 1,
                "backend": "gloo (dry rehearsal)", "rccl_ranks": world,
                "expected_ms": expected_gather_ms(world, gains.numel() * 4)}
        print(json.dumps({"metric": "iLQR iterations/sec (batch x horizon steps/s), quadrotor N=50 batch=4096",
                          "value": world * B * N * args.steps / max(elapsed, 1e-9), "unit": "steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "rehearsal": "dry: no GPU work, launcher / collective plumbing only",
                          "gather_ms": comm["gather_ms"], "rccl_ranks": world, "gather_ok": ok,
                          "ms_per_step_per_rank": per_rank_ms, "comm": comm,
                          "launched_by": os.environ.get("QT_BENCH_LAUNCHED_BY", "external launcher"),
                          "config": {"workload": "dry rehearsal", "batch_per_gpu": B, "global_batch": world * B}}))
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("dry rehearsal: gathered gains differ from what the ranks contributed")


def latest_sq_issue(kernel_prefix="sweep_tile16", pattern="*_pmc_sq_pure.json"):
    """How busy the sweep keeps its SIMDs' issue ports, from the latest committed SQ counter pass of this bench command
    (scripts/gpu_pmc_sq.sh): VALU port = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (per wave) x resident waves per SIMD; matrix
    pipe = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).  -> (valu_frac, mfma_frac, source) or Nones."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))[::-1]:
        try:
            with open(path) as fh:
                kern = json.load(fh)["kernels"]
            c = next(v for k, v in kern.items() if k.startswith(kernel_prefix))["counters"]
            valu = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] * (c["SQ_WAVES"] / 1024.0)
            mfma = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
            return float(valu), float(mfma), os.path.relpath(path, ROOT)
        except (StopIteration, KeyError, ValueError, OSError, ZeroDivisionError):
            continue
    return None, None, None


def latest_mfma_busy(kernel_prefix, pattern="*_pmc_sq_hybrid.json"):
    """Fraction of the kernel's duration its SIMDs' matrix pipes were busy, from the latest committed SQ counter pass:
    SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the 1024 SIMDs) / (1024 x GRBM_GUI_ACTIVE / 8 XCDs)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))[::-1]:
        try:
            with open(path) as fh:
                kern = json.load(fh)["kernels"]
            c = next(v for k, v in kern.items() if k.startswith(kernel_prefix))["counters"]
            return c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clock-settle-ms", type=float, default=60.0,
                    help="untimed run of the workload before the warm-up steps (GPU clock ramp); 0 disables")
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="trajectories per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline workload only (profiling runs)")
    ap.add_argument("--no-fused-sweep", action="store_true",
                    help="A/B: linearise in a kernel of its own (TILE16C records) instead of inside the sweep")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--workload", choices=("pure", "hybrid", "cartpole"), default="pure",
                    help="pure = BASELINE configs[2]/[3] (the metric's config); hybrid = configs[4], transformer-predicted "
                         "gains; cartpole = configs[1].  The default run reports the other two under `extras`.")
    args = ap.parse_args()

    if args.cpu_baseline_only:
        out = cpu_baseline()
        out["hybrid_config5"] = cpu_baseline_hybrid()
        out["config2_cartpole_N50_B1024"] = cpu_baseline_cartpole()
        # the published ladder (figures/quadrotor_result.png): iLQR(W) + TF(N - W)
        out["hybrid_windows"] = {str(w): cpu_baseline_hybrid(traj_per_core=1, iters=4, budget_s=4.0, window=w) for w in LADDER_WINDOWS}
        out["latency_B1"] = cpu_latency_b1()
        print(json.dumps(out))
        return
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0 or args.batch < 1:
        ap.error("--gpus, --steps and --batch must be >= 1 and --warmup >= 0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))           # launcher: no GPU call in this process, ever
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if os.environ.get("QT_BENCH_TEST_FAIL_RANK") == str(rank):     # test hook (tests/test_parallel_cpu.py): a rank that dies
        raise SystemExit(f"rank {rank}: failing on request (QT_BENCH_TEST_FAIL_RANK)")

    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        # In a child process of its own, before this process touches the GPU.  Measured: when the worker pool is
        # forked from THIS process, every later step of the GPU leg idles ~0.7 ms between the sweep and the line search
        # (host calls stay ~15 us, kernel durations in rocprofv3 are unchanged) — 2.3e8 instead of 1.08e9 steps/s.
        import subprocess
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit("cpu baseline leg failed:\n" + r.stderr[-2000:])
        cpu = json.loads(r.stdout.strip().splitlines()[-1])

    import torch
    import torch.distributed as dist

    # QT_BENCH_REHEARSAL: rehearse the N > 1 code path without N GPUs.  Never set by the driver; numbers mean nothing.
    #   "1"   : the real workload, ranks share the cards that exist, gloo instead of RCCL (which refuses two ranks on one
    #           device) — run on the one-GPU box
    #   "dry" : no GPU at all (the build container): launcher, rendezvous, barriers, the gain gather on CPU tensors of the
    #           configured shape and the JSON line; every iLQR step is a no-op
    rehearsal = os.environ.get("QT_BENCH_REHEARSAL", "")
    if rehearsal == "dry":
        return dry_rehearsal(args, rank, world, torch, dist)
    rehearsal = rehearsal == "1"
    from quattro_ilqr_amd import QuattroILQR, TransformerILQR, cartpole_model, ops, parallel, quadrotor_model
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    N = HORIZON
    offset = np.zeros(NX)
    offset[2] = 0.5                                       # quadrotor_mpc.py:64-66

    def make_workload(kind, B):
        """-> Workload.  kind: pure | hybrid | cartpole | rk4 (the pure quadrotor iteration with the RK4 integrator)."""
        if kind == "cartpole":
            md = cartpole_model(dt=0.01, integrator="euler")
            x0_h, u0_h = synthetic_cartpole(B, rank)
            sv = QuattroILQR(md, N, device=dev, tol=1e-1)
            tfm = None
        else:
            md = quadrotor_model(dt=0.01, integrator="rk4" if kind == "rk4" else "euler")
            x0_h, u0_h = synthetic_batch(B, rank)
            tfm = None
            window = int(kind.split(":")[1]) if ":" in kind else 1
            if kind.split(":")[0] in ("hybrid", "hybrid_fp16"):
                # BASELINE configs[4]: gains for t < N-1 from the transformer (architecture of the shipped quadrotor
                # checkpoint: 3 layers, d=128, 4 heads, ff=512, prompt 1, target 49, L=101; random-init weights), last
                # step from the sweep
                tfm = TransformerILQR.random_init(NX, NU * (1 + NX), prompt_len=window, target_len=N - window, d_model=128,
                                                  nhead=4, num_decoder_layers=3, dim_feedforward=512, max_seq_len=110,
                                                  device=dev, precision="fp16" if kind.startswith("hybrid_fp16") else "bf16")
            sv = QuattroILQR(md, N, device=dev, tf=tfm, state_offset=offset if tfm is not None else None)
        x0 = torch.as_tensor(x0_h, dtype=torch.float32, device=dev)
        u0 = torch.as_tensor(u0_h, dtype=torch.float32, device=dev)
        wl_ = Workload(torch, ops, sv, md, x0, u0, tf=tfm)
        if args.no_fused_sweep:
            sv.ensure_records()
            wl_.fused = False
            wl_._set_names()
        return wl_

    def roofline_of(kind, wl, kern_ms, B):
        if kind.startswith("hybrid"):
            tf_s = kern_ms["transformer"] * 1e-3
            fl = TF_FLOPS_PER_TRAJ * B
            traffic, src = latest_pmc_traffic("tf_stream_kernel", "*_pmc_hbm_hybrid.json") if B == BATCH_PER_GPU else (None, None)
            busy, busy_src = latest_mfma_busy("tf_stream_kernel") if B == BATCH_PER_GPU else (None, None)
            return {"traffic_source": src, "kernel": "tf_stream_kernel<4, 512> (quattro_tf_gains_bf16)", "bound": "mfma",
                    "executed_flops_per_launch": TF_EXECUTED_FLOPS_PER_TRAJ * B,
                    "executed_frac": TF_EXECUTED_FLOPS_PER_TRAJ * B / tf_s / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                    "mfma_busy": busy, "mfma_busy_source": busy_src,
                    "achieved": fl / tf_s / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": fl / tf_s / 1e12 / MFMA_BF16_PEAK_TFLOPS, "algorithmic_flops_per_launch": fl,
                    "avg_launch_ms": kern_ms["transformer"], "traffic": traffic,
                    "note": "traffic = HBM bytes per launch from the PMC passes (the 1.2 MB weight stream is read 4096 times "
                            "from L2, not from HBM).  algorithmic flops = SURVEY 8(d): 135.64 MFLOP per trajectory with the FULL L x L attention and "
                            "all L = 101 rows of every layer counted.  The kernel skips the attention tiles above the "
                            "diagonal but pads the sequence to 128 token slots, so it EXECUTES more (169 MFLOP per "
                            "trajectory of MFMA work); frac is algorithmic flops / time / peak, as SURVEY 8(d) defines it"}
        n, m = wl.model.n, wl.model.m
        per_step = 4 * (2 * n * n + 2 * n * m + m * m + n + m) + 4 * (m * n + m)
        per_traj = N * per_step + 4 * (n + n * n)
        sweep_s = kern_ms["sweep"] * 1e-3
        achieved = B * per_traj / sweep_s / 1e9
        roof = {"kernel": "quattro_linearize_sweep_f32 (sweep with the linearisation fused in)" if wl.fused
                else "quattro_riccati_sweep_f32", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": B * per_traj,
                "avg_launch_ms": kern_ms["sweep"], "traffic": None}
        if kind == "pure" and B == BATCH_PER_GPU:
            traffic, src = latest_pmc_traffic("sweep_tile16")
            roof["traffic"], roof["traffic_source"] = traffic, src
            if traffic is not None:
                roof["traffic_frac"] = traffic / sweep_s / 1e9 / HBM_PEAK_GBS
            # What the kernel IS bound by: (1) the dependency chain — a lone wave's time for one sweep (measured here, B = 2: one
            # workgroup on an empty chip) x the residency rounds of the batch (B waves on 1024 SIMDs x 4 waves); (2) instruction
            # issue with every SIMD carrying 4 such waves — VALU-port and matrix-pipe busy fractions from the SQ counter pass
            xs2, us2 = wl.solver.x[:2].clone(), wl.solver.u[:2].clone()
            K2 = torch.empty((2, N, m, n), dtype=torch.float32, device=dev)
            k2 = torch.empty((2, N, m), dtype=torch.float32, device=dev)
            for _ in range(5):
                ops.linearize_sweep(wl.model, xs2, us2, 0, K=K2, k=k2)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                ops.linearize_sweep(wl.model, xs2, us2, 0, K=K2, k=k2)
            e1.record()
            torch.cuda.synchronize(dev)
            lone_ms = e0.elapsed_time(e1) / 40
            rounds = -(-B // (1024 * 4))
            valu, mfma, sq_src = latest_sq_issue()
            roof.update({"lone_wave_sweep_ms": lone_ms, "residency_rounds": rounds, "chain_floor_ms": lone_ms * rounds,
                         "chain_frac": lone_ms * rounds / kern_ms["sweep"], "issue_frac": valu, "mfma_pipe_busy": mfma,
                         "issue_source": sq_src})
            roof["note"] = ("frac = SURVEY 8(d) algorithmic bytes (1872 B/step + terminal) / measured launch time / 8 TB/s; "
                            "traffic_frac = PMC-measured HBM bytes / the same time / 8 TB/s.  The kernel moves far FEWER "
                            "bytes than the accounting figure (constants of the problem are not streamed per step), so "
                            "HBM is not its roof.  The figures it IS bound by: chain_frac = chain_floor_ms / avg_launch_ms (a lone "
                            "wave's 50-step dependency chain x residency rounds: the floor no amount of parallelism removes) and "
                            "issue_frac (VALU port busy with 4 resident waves per SIMD) + mfma_pipe_busy (an exact-fp32 16x16x4 "
                            "MFMA holds the matrix pipe 32 cycles): DESIGN.md section 4.1")
        return roof

    def run(kind, B, steps, warmup, gather, settle_ms=None):
        """-> (workload, elapsed seconds [max over ranks], host issue seconds, comm dict or None)."""
        wl = make_workload(kind, B)
        settle_ms = args.clock_settle_ms if settle_ms is None else settle_ms
        # Clock settle (untimed, before the W warm-up steps, disclosed as `clock_settle_ms` in the output): the same steps
        # for a fixed wall time.  A freshly started process finds the GPU in its idle power state and the first tens of
        # milliseconds of work run through the DVFS ramp (measured: 0.151-0.153 ms per step with --steps 20 --warmup 5
        # alone against 0.140 ms from the ~200th step on, the MFMA-heavy hybrid kernel 773 vs 695 us); the metric is a
        # sustained rate.  The un-settled figure (the driver's --warmup only) is reported beside it (`*_no_settle`).
        t_s = time.perf_counter()
        while settle_ms > 0 and 1e3 * (time.perf_counter() - t_s) < settle_ms:
            for _ in range(10):
                wl.step(False)
            torch.cuda.synchronize()
        for _ in range(warmup):
            wl.step(False)
        gg = None
        if gather:
            # the one exchange of the sharded path: every rank's [K | k] buffer, as the sweep wrote it, into a preallocated
            # (world, flat) receive buffer — ONE collective, no repacking copy (parallel.GainGather); warmed once here
            sv = wl.solver
            gg = parallel.GainGather(B, sv.horizon, wl.model.m, wl.model.n, torch.float32, dev)
            gg(sv.gains_flat)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        barrier()
        t0 = time.perf_counter()
        ev[0].record()
        for _ in range(steps):
            wl.step(True)
        host_issue = time.perf_counter() - t0            # host time to enqueue the steps (GPU-bound when << elapsed)
        ev[1].record()
        if gather:
            gg(wl.solver.gains_flat)
        ev[2].record()
        barrier()
        elapsed = time.perf_counter() - t0
        comm = None
        if world > 1:
            mine = torch.tensor([elapsed * 1e3, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])], dtype=torch.float64,
                                device=dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            every = torch.stack(every).cpu().numpy()
            elapsed = float(every[:, 0].max()) * 1e-3    # the contract: MAX over ranks
            comm = {"gather_ms": float(every[:, 2].max()), "gather_ms_per_rank": [float(v) for v in every[:, 2]],
                    "compute_ms_per_rank": [float(v) for v in every[:, 1]],
                    "ms_per_step_per_rank": [float(v) / steps for v in every[:, 0]],
                    "gather_bytes_received_per_rank": gg.bytes_received_per_rank if gg is not None else 0,
                    "collectives_per_gather": 1,
                    "expected_ms": expected_gather_ms(world, wl.solver.gains_flat.numel() * 4) if gg is not None else None,
                    "backend": "gloo (rehearsal)" if rehearsal else "nccl (RCCL)", "rccl_ranks": world,
                    "note": "gather_ms = HIP events on the compute stream from the end of this rank's last iLQR step to the "
                            "end of the all-gather of [K | k] (includes waiting for the slowest rank); it is inside the "
                            "timed region"}
        return wl, elapsed, host_issue, comm

    kind = args.workload
    B = args.batch
    # first the driver's command as it stands (W warm-up steps in a fresh process, no settle phase), then the settled rate
    no_settle = None
    if args.clock_settle_ms > 0:
        _, el_ns, _, comm_ns = run(kind, B, args.steps, args.warmup, world > 1, settle_ms=0.0)
        no_settle = {"value_no_settle": world * B * HORIZON * args.steps / el_ns,
                     "ms_per_step_no_settle": 1e3 * el_ns / args.steps}
        if comm_ns is not None:
            no_settle["gather_ms_no_settle"] = comm_ns["gather_ms"]
    wl, elapsed, host_issue, comm = run(kind, B, args.steps, args.warmup, world > 1)
    solver = wl.solver
    kern_ms = wl.kernel_ms()
    accepted = float((solver.alpha_idx >= 0).float().mean().item())
    bad = int((solver.status != 0).sum().item())

    # ------------------------------------------------------------------------------------------ extras (rank 0, N = 1)
    extras = {}
    if rank == 0 and world == 1 and not args.no_extras and kind == "pure":
        def product_iterate(sv_kind, B_, steps=20):
            """The product's own iteration — QuattroILQR.iterate(): pure mode = ONE C call (quattro_ilqr_iterate_f32:
            linearise + sweep + line search), no simulate (an accepted forward pass IS the next nominal) — from the same
            synthetic nominal every time."""
            w2 = make_workload(sv_kind, B_)
            s2 = w2.solver
            x_ref_t = s2._x_ref_t
            if w2.hybrid:
                s2._x_ref_t.copy_(torch.as_tensor(np.asarray(w2.model.x_ref, dtype=np.float32), device=dev))
                s2._offset_t.copy_(torch.as_tensor(offset.astype(np.float32), device=dev))
            ops.simulate(w2.model, w2.x0, w2.u0, x=s2.x, cost=s2.cost)
            x_keep, c_keep = s2.x.clone(), s2.cost.clone()

            def one():
                s2.u.copy_(w2.u0); s2.x.copy_(x_keep); s2.cost.copy_(c_keep); s2.active.fill_(1)
                s2.iterate(x_ref_t)
            for _ in range(3):
                one()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(steps):
                one()
            torch.cuda.synchronize(dev)
            ms = 1e3 * (time.perf_counter() - t1) / steps
            return {"ms_per_iteration": ms, "steps_per_s": B_ * N / (ms * 1e-3), "batch": B_,
                    "note": "includes 4 small reset copies per iteration (u, x, cost, active)"}
        extras["product_iterate_pure"] = product_iterate("pure", B)

        def device_loop_solve(sv_kind, B_, iters=20, reps=5):
            """The device-resident loop: QuattroILQR.solve(max_iter=iters, fixed_iters=True) = ONE C call
            (quattro_ilqr_solve_f32) = one persistent launch where the model has such a kernel: nominal rollout + `iters`
            iterations of every trajectory, no host involvement in between.  Reported per iteration."""
            w2 = make_workload(sv_kind, B_)
            s2 = w2.solver
            s2.solve(w2.x0, w2.u0, max_iter=iters, fixed_iters=True)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(reps):
                s2.solve(w2.x0, w2.u0, max_iter=iters, fixed_iters=True)
            torch.cuda.synchronize(dev)
            ms = 1e3 * (time.perf_counter() - t1) / reps
            return {"ms_per_solve": ms, "iterations": iters, "ms_per_iteration": ms / iters,
                    "steps_per_s": B_ * N * iters / (ms * 1e-3), "batch": B_,
                    "one_persistent_launch": bool(ops.model_has_device_loop(w2.model)),
                    "note": "includes the nominal rollout and two small uploads (x0, u) per solve"}
        extras["device_loop_solve_pure"] = device_loop_solve("pure", B)

        # BASELINE configs[4]: hybrid iteration, B = 4096
        # (the MFMA-heavy kernel settles more slowly than the pure workload: 773 us per launch over the first 25 launches,
        #  693-699 us from the ~100th on, one box — hence 100 timed steps after the settle phase)
        HS, HW = 100, 20
        # (measured twice, the second run reported: a process's first stretch of long launches can contain ONE host call that
        #  blocks ~40 ms inside the runtime — DESIGN section 5, "a one-off host stall" — which the first run absorbs; when it fell
        #  into the 100 timed steps it read as 1.29 instead of 0.78 ms per step)
        run("hybrid", BATCH_PER_GPU, HS, HW, False)
        wh, el_h, hi_h, _ = run("hybrid", BATCH_PER_GPU, HS, HW, False)
        km = wh.kernel_ms()
        extras["hybrid_config5"] = {
            "workload": "quadrotor n_x=12 n_u=4 N=50 B=4096, hybrid iteration (BASELINE configs[4]) = simulate + 1-step "
                        "tail linearize/sweep + bf16-MFMA transformer (L=101, d=128, 3 layers, random-init) + gain-stack "
                        "assembly + 6-alpha line search/commit",
            "value": BATCH_PER_GPU * N * HS / el_h, "unit": "steps/s", "ms_per_step": 1e3 * el_h / HS, "steps": HS,
            "warmup": HW, "kernel_ms": km, "host_issue_ms_per_step": 1e3 * hi_h / HS,
            "roofline": roofline_of("hybrid", wh, km, BATCH_PER_GPU),
            "accepted_fraction": float((wh.solver.alpha_idx >= 0).float().mean().item()),
            "product_iterate": product_iterate("hybrid", BATCH_PER_GPU)}
        if cpu is not None and "hybrid_config5" in cpu:
            extras["hybrid_config5"]["cpu_baseline"] = cpu["hybrid_config5"]
        del wh
        # the same iteration with the predictor's fp16 operand variant (TransformerILQR(precision="fp16"): the arithmetic of
        # the reference's own predict(), 12x closer to its fp32 output than bf16 on the shipped checkpoints)
        wf, el_f, _, _ = run("hybrid_fp16", BATCH_PER_GPU, HS, HW, False)
        kmf = wf.kernel_ms()
        extras["hybrid_config5"]["fp16_operands"] = {"ms_per_step": 1e3 * el_f / HS, "transformer_ms": kmf["transformer"],
                                                     "frac_of_mfma_peak": TF_FLOPS_PER_TRAJ * BATCH_PER_GPU / (kmf["transformer"] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS}
        del wf
        # the reference's published ladder (figures/quadrotor_result.png): iLQR(W) + TF(N - W) per iteration,
        # W = 40, 30, 20, 10 (W = 1 is hybrid_config5 above; pure iLQR is the headline) — random-init predictors of the shipped
        # architecture with prompt_len = W, target_len = N - W (always L = 101 tokens), each beside its CPU counterpart
        ladder = {}
        for w in LADDER_WINDOWS:
            ww, el_w, _, _ = run(f"hybrid:{w}", BATCH_PER_GPU, 30, 10, False)
            kmw = ww.kernel_ms()
            ladder[str(w)] = {"tf_window": w, "ms_per_step": 1e3 * el_w / 30, "value": BATCH_PER_GPU * N * 30 / el_w, "unit": "steps/s",
                              "kernel_ms": kmw}
            if cpu is not None and "hybrid_windows" in cpu:
                ladder[str(w)]["cpu_baseline"] = cpu["hybrid_windows"].get(str(w))
            del ww
        ladder["1"] = {"tf_window": 1, "ms_per_step": extras["hybrid_config5"]["ms_per_step"], "see": "hybrid_config5"}
        ladder["50 (pure iLQR)"] = {"tf_window": 50, "ms_per_step": 1e3 * elapsed / args.steps, "see": "the headline line"}
        ladder["reference_published_ms_per_iteration_one_trajectory"] = {"50 (pure iLQR)": 246.25, "40": 201.37, "30": 182.87,
                                                                         "20": 102.49, "10": 54.76, "1": 9.10,
                                                                         "source": "figures/quadrotor_result.png via BASELINE.md section 1 (hardware not stated)"}
        extras["hybrid_windows"] = ladder
        # BASELINE configs[1]: cart-pole N = 50, B = 1024 (launch-bound: eager and hipGraph replay of the product iteration)
        wc, el_c, hi_c, _ = run("cartpole", 1024, 50, 5, False)
        km = wc.kernel_ms()
        c2 = {"workload": "cart-pole n_x=4 n_u=1 N=50 B=1024 (BASELINE configs[1]), pure iLQR iteration = simulate + "
                          "linearisation and Riccati sweep (one fused launch, one DPP quad of lanes per trajectory) + 6-alpha line "
                          "search/commit",
              "value": 1024 * N * 50 / el_c, "unit": "steps/s", "ms_per_step": 1e3 * el_c / 50, "steps": 50,
              "kernel_us": {k: 1e3 * v for k, v in km.items()}, "host_issue_ms_per_step": 1e3 * hi_c / 50,
              "roofline": roofline_of("cartpole", wc, km, 1024),
              "product_iterate": product_iterate("cartpole", 1024, steps=50),
              "device_loop_solve": device_loop_solve("cartpole", 1024, iters=50, reps=10)}
        if cpu is not None and "config2_cartpole_N50_B1024" in cpu:
            c2["cpu_baseline"] = cpu["config2_cartpole_N50_B1024"]
        extras["config2_cartpole_N50_B1024"] = c2
        del wc

        # the pure iteration with RK4, the default integrator of the reference's MPC classes (quadrotor_mpc.py:12): the
        # linearisation is forward-mode through the four stages (dense [A | B], TILE16R records), no fused sweep
        wr, el_r, hi_r, _ = run("rk4", BATCH_PER_GPU, 30, 5, False)
        extras["quadrotor_rk4_B4096"] = {
            "workload": "quadrotor n_x=12 n_u=4 N=50 B=4096, pure iLQR iteration with the RK4 integrator = simulate + "
                        "linearisation (RK4 forward-mode, TILE16R records) + Riccati sweep + 6-alpha line search/commit",
            "value": BATCH_PER_GPU * N * 30 / el_r, "unit": "steps/s", "ms_per_step": 1e3 * el_r / 30, "steps": 30,
            "kernel_us": {k: 1e3 * v for k, v in wr.kernel_ms().items()}}
        del wr

        # one solve of the headline batch with the REAL exit test (per-trajectory convergence, tol 1e-3, cold start u = 0
        # like the reference), and BASELINE configs[0] as a single-trajectory call of the drop-in
        from quattro_ilqr_amd import BatchedMPC, CartPoleMPC
        model = wl.model
        x0 = wl.x0
        conv = QuattroILQR(model, N, max_iter=100, tol=1e-3, device=dev)
        conv.solve(x0, max_iter=9)          # warm-up long enough to reach a convergence check (first use of torch's
                                            # reduce kernel loads its code object: ~20 ms once per process)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        res = conv.solve(x0)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t1
        its = res["iters"].double()
        extras["converged_solve"] = {"batch": B, "wall_ms": 1e3 * wall, "iterations_mean": float(its.mean().item()),
                                     "iterations_max": int(its.max().item()),
                                     "steps_per_s": float(its.sum().item()) * N / wall,
                                     "flagged": int((res["status"] != 0).sum().item()),
                                     "device_loop": bool(conv.device_loop and ops.model_has_device_loop(model))}
        convh = QuattroILQR(model, N, max_iter=100, tol=1e-3, device=dev, device_loop=False)
        convh.solve(x0, max_iter=9)
        walls_h = []
        for _ in range(3):          # (median of three: a single call can contain the runtime's one-off ~40 ms host stall)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            convh.solve(x0)
            torch.cuda.synchronize(dev)
            walls_h.append(1e3 * (time.perf_counter() - t1))
        extras["converged_solve"]["wall_ms_host_driven_loop"] = float(np.median(walls_h))
        del convh
        # the receding-horizon loop itself — B controllers, warm-started, plant = the device model
        mpc = BatchedMPC(model, N, max_iter=100, tol=1e-3, device=dev)
        mpc.run(x0, 2)
        mpc.u_warm = None
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        runo = mpc.run(x0, 10)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t1
        extras["batched_mpc"] = {"controllers": B, "control_steps": 10, "wall_ms": 1e3 * wall,
                                 "ms_per_control_step": 1e2 * wall,
                                 "ilqr_iterations_per_control_step_mean": float(runo["iters"].double().mean().item()),
                                 "control_steps_per_s": B * 10 / wall,
                                 "device_loop": bool(ops.model_has_device_loop(model))}
        mpc.u_warm = None
        mpc.run(x0, 2, device_loop=False)
        walls_h = []
        for _ in range(3):
            mpc.u_warm = None
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            mpc.run(x0, 10, device_loop=False)
            torch.cuda.synchronize(dev)
            walls_h.append(1e3 * (time.perf_counter() - t1))
        extras["batched_mpc"]["wall_ms_host_driven_loop"] = float(np.median(walls_h))
        # Does the predictor pay on this GPU?  The reference's claim (README.md:29-33: the transformer makes MPC 17.8x faster)
        # rests on its backward pass costing 865 Python-level cost evaluations per step.  Same problems, SHIPPED quadrotor
        # checkpoint (tests/golden/tf_weights_quadrotor.npz: iLQR(1) + TF(49)), real exit test: iterations to converge, wall
        # time and final cost of the hybrid solve beside the pure one — cold-started batch, and a warm-started closed loop.
        tf_ship = TransformerILQR(12, 52, device=dev).load(os.path.join(ROOT, "tests", "golden", "tf_weights_quadrotor.npz"))
        convt = QuattroILQR(model, N, max_iter=100, tol=1e-3, tf=tf_ship, state_offset=offset, device=dev, use_graph=True)
        convt.solve(x0, max_iter=9)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        rest = convt.solve(x0)
        torch.cuda.synchronize(dev)
        wall_t = time.perf_counter() - t1
        its_t = rest["iters"].double()
        cost_t = rest["cost"].clone()
        res_p = conv.solve(x0)
        torch.cuda.synchronize(dev)
        pay = {"weights": "shipped quadrotor checkpoint, iLQR(1) + TF(49), bf16 MFMA",
               "cold_start_B4096": {
                   "pure": {"wall_ms": extras["converged_solve"]["wall_ms"], "iterations_mean": extras["converged_solve"]["iterations_mean"],
                            "iterations_max": extras["converged_solve"]["iterations_max"], "final_cost_mean": float(res_p["cost"].mean().item()),
                            "final_cost_median": float(res_p["cost"].median().item())},
                   "hybrid": {"wall_ms": 1e3 * wall_t, "iterations_mean": float(its_t.mean().item()), "iterations_max": int(its_t.max().item()),
                              "final_cost_mean": float(cost_t.mean().item()), "final_cost_median": float(cost_t.median().item()),
                              "fraction_with_cost_within_1pct_of_pure": float(((cost_t - res_p["cost"]) <= 0.01 * res_p["cost"].abs()).double().mean().item())}}}
        mpct = BatchedMPC(model, N, max_iter=100, tol=1e-3, tf=tf_ship, state_offset=offset, device=dev)
        mpct.run(x0, 2)
        mpct.u_warm = None
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        runt = mpct.run(x0, 10)
        torch.cuda.synchronize(dev)
        wall_m = time.perf_counter() - t1
        xr_t = torch.as_tensor(np.asarray(model.x_ref), dtype=torch.float32, device=dev)
        track = lambda r: float((r["x"][:, -1] - xr_t).norm(dim=1).mean().item())
        pay["closed_loop_B4096_10_steps"] = {
            "pure": {"wall_ms": extras["batched_mpc"]["wall_ms"], "ilqr_iterations_per_control_step_mean": extras["batched_mpc"]["ilqr_iterations_per_control_step_mean"],
                     "mean_distance_to_reference_after_10_steps": track(runo)},
            "hybrid": {"wall_ms": 1e3 * wall_m, "ilqr_iterations_per_control_step_mean": float(runt["iters"].double().mean().item()),
                       "mean_distance_to_reference_after_10_steps": track(runt)}}
        if cpu is not None and "latency_B1" in cpu:
            pay["cpu_one_core_ms_per_iteration"] = {"pure": cpu["latency_B1"]["quadrotor_N50"]["pure"]["ms_per_iteration"],
                                                    "hybrid": cpu["latency_B1"]["quadrotor_N50"]["1"]["ms_per_iteration"]}
        pay["reading"] = ("fewer iterations do not buy time here: a hybrid iteration costs 5-6x a pure one at B = 4096 and ~2x at B = 1 "
                          "(latency_B1), against 32x cheaper on the reference's CPU path; see DESIGN section 4")
        extras["predictor_payoff"] = pay
        del convt, mpct
        # The reference's own use case and only published metric: ONE trajectory through the drop-in classes (QuadrotorMPC /
        # CartPoleMPC -> iLQR_TF.optimize), wall time per iteration / per control step, next to the oracle on one host core
        # and the published bars.  optimize() = one persistent launch + one download (pure), one captured graph per
        # iteration (hybrid); the device-side split comes from the log ring's stamps (the *_time lists).
        def dropin_latency(make, x_start, reps=7):
            mpc = make()
            n_h = len(mpc.ilqr.u)
            cold = [np.zeros_like(np.asarray(mpc.ilqr.u[0], dtype=np.float64)) for _ in range(n_h)]
            walls, its = [], 0
            for r in range(reps + 2):                       # two untimed calls: buffers, graph capture, code objects
                mpc.ilqr.u = [c.copy() for c in cold]
                mpc.ilqr.logs = []
                mpc.ilqr.x0 = x_start
                for lst in mpc.ilqr.get_time():
                    del lst[:]
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                mpc.ilqr.optimize(mpc.x_ref)
                w = time.perf_counter() - t1
                if r >= 2:
                    walls.append(w)
                its = len(mpc.ilqr.logs)
            tt = mpc.ilqr.get_time()
            med = float(np.median(walls))
            res = {"iterations": its, "ms_per_solve": 1e3 * med, "ms_per_iteration": 1e3 * med / max(1, its),
                   "ms_per_solve_min": 1e3 * min(walls),
                   "device_backward_ms_per_iteration": 1e3 * float(np.mean(tt[1])) if tt[1] else None,
                   "device_linesearch_ms_per_iteration": 1e3 * float(np.mean(tt[2])) if tt[2] else None}
            if len(tt) == 4 and tt[3]:
                res["device_inference_ms_per_iteration"] = 1e3 * float(np.mean(tt[3]))
            return res
        from quattro_ilqr_amd import QuadrotorMPC
        xq = np.zeros(12); xq[2] = 0.5; xq[6] = 0.1          # README start: roll 0.1 rad (quadrotor_sim.py:250)
        PUBLISHED_MS = {"pure": 246.25, "40": 201.37, "30": 182.87, "20": 102.49, "10": 54.76, "1": 9.10}
        lat = {"workload": "ONE trajectory through the drop-in classes (iLQR_TF.optimize): quadrotor n_x=12 n_u=4 N=50 Euler from "
                           "the README start (roll 0.1 rad, cold start), max_iter 100, tol 1e-3; median of 7 solves",
               "published_hardware": "Apple M4 Pro (README.md:31-33)", "quadrotor_N50": {}}
        lat["quadrotor_N50"]["pure"] = dropin_latency(lambda: QuadrotorMPC(horizon=N, dt=0.01, integration_method="euler", device=str(dev)), xq)
        for w in LADDER_WINDOWS + (1,):
            if w == 1:
                tfw = TransformerILQR(12, 52, device=str(dev)).load(os.path.join(ROOT, "tests", "golden", "tf_weights_quadrotor.npz"))
            else:
                tfw = TransformerILQR.random_init(NX, NU * (1 + NX), prompt_len=w, target_len=N - w, d_model=128, nhead=4,
                                                  num_decoder_layers=3, dim_feedforward=512, max_seq_len=110, device=str(dev))
            lat["quadrotor_N50"][str(w)] = dropin_latency(
                lambda: QuadrotorMPC(horizon=N, dt=0.01, integration_method="euler", transformer_model=tfw, device=str(dev)), xq)
            lat["quadrotor_N50"][str(w)].update(tf_window=w, weights="shipped checkpoint" if w == 1 else "random-init")
        for key, row in lat["quadrotor_N50"].items():
            row["published_ms_per_iteration"] = PUBLISHED_MS[key]
            if cpu is not None and "latency_B1" in cpu:
                row["cpu_one_core_ms_per_iteration"] = cpu["latency_B1"]["quadrotor_N50"][key]["ms_per_iteration"]
        # BASELINE configs[0]: cart-pole N = 30, one control_step from the simulator's start (cartpole_sim.py:208)
        c1 = dropin_latency(lambda: CartPoleMPC(horizon=30, dt=0.01, integration_method="euler", ilqr_only=True, device=str(dev)),
                            np.array([0.0, 0.0, 0.1, 0.0]))
        c1["published_ms_per_iteration"] = 10.19
        if cpu is not None:
            c1["cpu_baseline"] = dict(cpu["config1_cartpole_N30_B1"], cores=1, kind="port", unit="ms per control step",
                                      sample="oracle/ilqr.py optimize(), one core, one call from the simulator's start")
        lat["config1_cartpole_N30"] = c1
        if cpu is not None and "latency_B1" in cpu:
            lat["cpu_baseline"] = {k: v for k, v in cpu["latency_B1"].items() if k != "quadrotor_N50"}
        extras["latency_B1"] = lat
        extras["config1_cartpole_N30_B1"] = {"ms": c1["ms_per_solve"], "iterations": c1["iterations"], "see": "latency_B1"}
        if cpu is not None:
            extras["config1_cartpole_N30_B1"]["cpu_baseline"] = c1["cpu_baseline"]
        # a user-compiled model (what the reference's callable interface allows, on the device): the planar
        # example, B = 4096, N = 50, 20 fixed iterations through the generic kernels (one C call; ROWMAJOR records from
        # forward-mode duals, pivoting sweep, one lane per line-search candidate)
        try:
            from quattro_ilqr_amd import user_model
            t1 = time.perf_counter()
            um = user_model.example_planar_model()
            build_s = time.perf_counter() - t1
            rng_u = np.random.default_rng(0)
            ux0 = torch.as_tensor(np.asarray(um.x_ref) + rng_u.normal(0, 0.3, (B, 6)) * np.array([1, 1, 0.3, 0.5, 0.5, 0.5]),
                                  dtype=torch.float32, device=dev)
            uu0 = torch.as_tensor(np.full((B, N, 2), 9.81 / 2) + rng_u.normal(0, 0.2, (B, N, 2)), dtype=torch.float32, device=dev)
            us = QuattroILQR(um, N, device=dev)
            us.solve(ux0, uu0, max_iter=20, fixed_iters=True)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            us.solve(ux0, uu0, max_iter=20, fixed_iters=True)
            e1.record()
            torch.cuda.synchronize(dev)
            extras["user_model_planar_B4096"] = {
                "workload": "user-compiled planar two-rotor vehicle n_x=6 n_u=2 N=50 B=4096 RK4, 20 fixed iLQR iterations from one C call",
                "ms_per_step": e0.elapsed_time(e1) / 20, "value": B * N * 20 / (1e-3 * e0.elapsed_time(e1)), "unit": "steps/s",
                "compile_or_cache_s": build_s}
            del us
        except Exception as exc:          # (needs hipcc at run time unless the model library is already cached in-tree)
            extras["user_model_planar_B4096"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}

        # one mini-batch of TransformerILQR.fit on the shipped quadrotor predictor shape — the
        # hand-written step (quattro_tf_train_step_f32 + quattro_tf_adam_f32) beside torch autograd / rocBLAS, same weights
        import torch.nn.functional as F
        from quattro_ilqr_amd import train_hip, training
        shp = (12, 52, 128, 4, 3, 512, 51, 1, 49)
        TB = 256
        prm, buf = training.init_params(*shp[:6], 110, shp[8], seed=0, device=dev)
        trn = train_hip.HipTrainer(*shp, 0.0, buf["pos_encoder.pe"].cpu().numpy(), dev)
        trn.load_state_dict({k: v.detach() for k, v in prm.items()})
        gg = torch.Generator().manual_seed(1)
        tx, tu, ty = (torch.randn(sz, generator=gg).to(dev) for sz in ((TB, shp[6], 12), (TB, shp[7], 52), (TB, shp[8], 52)))
        topt = torch.optim.Adam(list(prm.values()), lr=1e-3)

        def hip_step():
            trn.forward_backward(tx, tu, ty)
            trn.adam_step()

        def torch_step():
            topt.zero_grad(set_to_none=True)
            F.mse_loss(training.forward(prm, buf, tx, tu, shp[3]), ty).backward()
            topt.step()

        tms = {}
        for nm, fn in (("hip", hip_step), ("torch", torch_step)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize(dev)
            tms[nm] = 1e3 * (time.perf_counter() - t1) / 20
        extras["train_step_quadrotor_B256"] = {
            "ms_hip": tms["hip"], "ms_torch_autograd": tms["torch"], "sequences_per_s_hip": TB / (tms["hip"] * 1e-3),
            "note": "forward + MSE + backward + Adam of the 616 k-parameter predictor (L = 101), fp32; hip = "
                    "csrc/tf_train.hip through the C ABI, torch = training.forward under autograd (rocBLAS)"}
        del trn, prm, topt

    if rank == 0:
        total_steps = world * B * N * args.steps
        roof = roofline_of(kind, wl, kern_ms, B)
        if kind == "hybrid":
            workload = ("quadrotor n_x=12 n_u=4 N=50, hybrid iteration (BASELINE configs[4]) = simulate + 1-step tail "
                        "linearize/sweep + bf16-MFMA transformer (L=101, d=128, 3 layers, random-init) + gain-stack "
                        "assembly + 6-alpha line search/commit")
        elif kind == "cartpole":
            workload = "cart-pole n_x=4 n_u=1 N=50 (BASELINE configs[1]), pure iLQR iteration"
        else:
            workload = ("quadrotor n_x=12 n_u=4 N=50, pure iLQR iteration = simulate (nominal rollout + cost) + "
                        "linearisation and Riccati sweep (one fused launch) + 6-alpha line search/commit "
                        "(BASELINE configs[2]; configs[3] when n_gpus=8)")
        out = {
            "metric": "iLQR iterations/sec (batch x horizon steps/s), quadrotor N=50 batch=4096",
            "value": total_steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "clock_settle_ms": args.clock_settle_ms,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "batch_per_gpu": B, "global_batch": world * B, "horizon": N, "n_x": wl.model.n, "n_u": wl.model.m,
                       "integrator": "euler", "dt": 0.01, "parallelism": f"dp{world} (independent trajectory shards"
                       + (", one all-gather of [K | k])" if world > 1 else ")")},
            "iterations_per_s": world * B * args.steps / elapsed,
            "kernel_ms": kern_ms, "host_issue_ms_per_step": 1e3 * host_issue / args.steps,
            "accepted_fraction": accepted, "flagged_trajectories": bad,
            "roofline": roof,
        }
        if no_settle is not None:
            out.update(no_settle)
        if comm is not None:
            out.update({"gather_ms": comm["gather_ms"], "rccl_ranks": comm["rccl_ranks"],
                        "ms_per_step_per_rank": comm["ms_per_step_per_rank"], "comm": comm})
        if cpu is not None:
            sub = {"pure": None, "hybrid": "hybrid_config5", "cartpole": "config2_cartpole_N50_B1024"}[kind]
            base = cpu if sub is None else cpu[sub]
            out["cpu_baseline"] = {k: v for k, v in base.items() if k not in ("hybrid_config5", "config2_cartpole_N50_B1024", "hybrid_windows", "latency_B1")}
            out["speedup_vs_cpu_all_cores"] = out["value"] / base["value"]
        if extras:
            out["extras"] = extras
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
