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
def _cpu_worker(args):
    """One trajectory, `iters` iterations of the oracle's FD iLQR (fp64, eps=1e-5, inv(Q_uu+1e-6 I), 6-alpha search)."""
    seed, iters = args
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    from oracle import ilqr as o_ilqr
    from oracle import models as o_models
    spec = o_models.quadrotor_spec()
    rng = np.random.default_rng(seed)
    x0 = spec.x_ref + rng.uniform(-1.0, 1.0, NX) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    u_seq = [2.4525 + 0.1 * rng.standard_normal(NU) for _ in range(HORIZON)]
    done = 0
    for _ in range(iters):          # tol < 0: never "converged", so exactly `iters` iterations like the GPU leg
        u_seq, _, logs = o_ilqr.optimize(spec.f, spec.L, spec.Lf, x0, u_seq, HORIZON, max_iter=1, tol=-1.0, keep_logs=True)
        done += 1
    return done


def _cpu_warm(_):
    from oracle import ilqr, models  # noqa: F401  (import cost stays outside the timed map)
    return 0


def cpu_baseline(per_core_traj=1, iters=5):
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    S = per_core_traj * cores
    ctx = mp.get_context("fork")                       # forked BEFORE this process touches the GPU
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_warm, range(cores))
        t0 = time.time()
        done = pool.map(_cpu_worker, [(9000 + i, iters) for i in range(S)], chunksize=1)
        wall = time.time() - t0
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
    return {"value": steps / wall, "unit": "steps/s", "cores": cores, "kind": "port", "config1_cartpole_N30_B1": c1,
            "sample": f"{S} quadrotor N=50 trajectories x {iters} iLQR iterations, oracle/ilqr.py (fp64 finite differences, "
                      f"reference algorithm), multiprocessing.Pool({cores}), wall {wall:.1f} s",
            "per_core": steps / wall / cores}


def _cpu_worker_hybrid(args):
    """One trajectory, `iters` hybrid iterations of the oracle (configs[4]): FD tail step + NumPy fp32 transformer."""
    seed, iters, wpath = args
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    from oracle import ilqr as o_ilqr
    from oracle import models as o_models
    from oracle import transformer as o_tf
    z = np.load(wpath)
    W = {k: z[k] for k in z.files if not k.startswith("norm.")}
    norm = {k[5:]: z[k] for k in z.files if k.startswith("norm.")}
    spec = o_models.quadrotor_spec()
    rng = np.random.default_rng(seed)
    x0 = spec.x_ref + rng.uniform(-1.0, 1.0, NX) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    u_seq = [2.4525 + 0.1 * rng.standard_normal(NU) for _ in range(HORIZON)]
    offset = np.zeros(NX); offset[2] = 0.5
    predict = lambda xe, pr: o_tf.predict(W, norm, xe, pr, 4, 1, dtype=np.float32)
    done = 0
    for _ in range(iters):
        u_seq, _, _ = o_ilqr.optimize(spec.f, spec.L, spec.Lf, x0, u_seq, HORIZON, x_ref=spec.x_ref, max_iter=1, tol=-1.0,
                                      tf_predict=predict, tf_window=1, state_offset=offset, keep_logs=True)
        done += 1
    return done


def cpu_baseline_hybrid(per_core_traj=2, iters=30):
    """configs[4] on the host: the oracle's hybrid iteration with a NumPy fp32 evaluation of the same random-init
    transformer ('for config 5 also time the CPU fp32 batched transformer')."""
    import multiprocessing as mp
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "quattro-transformer-ilqr_amd"))
    from quattro_ilqr_amd import TransformerILQR
    tf = TransformerILQR.random_init(NX, NU * (1 + NX), prompt_len=1, target_len=HORIZON - 1, d_model=128, nhead=4,
                                     num_decoder_layers=3, dim_feedforward=512, max_seq_len=110, device="cpu")
    wpath = os.path.join(tempfile.mkdtemp(), "w.npz")
    np.savez(wpath, **tf._w, **{"norm." + k: v for k, v in tf._norm.items()})
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    S = per_core_traj * cores
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_warm, range(cores))
        t0 = time.time()
        done = pool.map(_cpu_worker_hybrid, [(9000 + i, iters, wpath) for i in range(S)], chunksize=1)
        wall = time.time() - t0
    steps = sum(done) * HORIZON
    return {"value": steps / wall, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{S} quadrotor N=50 trajectories x {iters} hybrid iterations, oracle/ilqr.py + oracle/transformer.py "
                      f"(fp64 finite differences on the 1-step tail, NumPy fp32 transformer L=101), "
                      f"multiprocessing.Pool({cores}), wall {wall:.1f} s",
            "per_core": steps / wall / cores}


# --------------------------------------------------------------------------------------------- GPU leg
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="trajectories per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--workload", choices=("pure", "hybrid"), default="pure",
                    help="pure = BASELINE configs[2]/[3] (the metric's config); hybrid = configs[4], transformer-predicted gains")
    args = ap.parse_args()

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_hybrid() if args.workload == "hybrid" else cpu_baseline()))
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        # In a child process of its own, before this process touches the GPU.  Measured: when the 64-worker pool is
        # forked from THIS process, every later step of the GPU leg idles ~0.7 ms between the sweep and the line search
        # (host calls stay ~15 us, kernel durations in rocprofv3 are unchanged) — 2.3e8 instead of 1.08e9 steps/s.
        import subprocess
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", args.workload],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit("cpu baseline leg failed:\n" + r.stderr[-2000:])
        cpu = json.loads(r.stdout.strip().splitlines()[-1])

    import torch
    import torch.distributed as dist
    from quattro_ilqr_amd import QuattroILQR, ops, parallel, quadrotor_model

    # QT_BENCH_REHEARSAL=1: rehearse the N > 1 code path on a box with fewer GPUs than ranks (ranks share the cards,
    # gloo instead of RCCL, which refuses two ranks on one device).  Never set by the driver; numbers mean nothing.
    rehearsal = os.environ.get("QT_BENCH_REHEARSAL") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B, N = args.batch, HORIZON
    model = quadrotor_model(dt=0.01, integrator="euler")
    hybrid = args.workload == "hybrid"
    tf = None
    if hybrid:
        # BASELINE configs[4]: gains for t < N-1 from the transformer (architecture of the shipped quadrotor checkpoint:
        # 3 layers, d=128, 4 heads, ff=512, prompt 1, target 49, L=101; random-init weights), last step from the sweep
        from quattro_ilqr_amd import TransformerILQR
        from quattro_ilqr_amd.solver import _pack_prompt
        tf = TransformerILQR.random_init(NX, NU * (1 + NX), prompt_len=1, target_len=N - 1, d_model=128, nhead=4,
                                         num_decoder_layers=3, dim_feedforward=512, max_seq_len=110, device=dev)
    solver = QuattroILQR(model, N, device=dev, tf=tf)
    x0_h, u0_h = synthetic_batch(B, rank)
    x0 = torch.as_tensor(x0_h, dtype=torch.float32, device=dev)
    u0 = torch.as_tensor(u0_h, dtype=torch.float32, device=dev)
    solver._alloc(B)
    x_ref_t = torch.as_tensor(np.asarray(model.x_ref, dtype=np.float32), device=dev)
    offset_t = torch.zeros(NX, dtype=torch.float32, device=dev)
    offset_t[2] = 0.5                                     # quadrotor_mpc.py:64-66
    x_shift = np.asarray(model.x_ref, dtype=np.float64) - offset_t.double().cpu().numpy()

    names = ("simulate", "linearize", "sweep", "transformer", "assemble", "linesearch") if hybrid else \
            ("simulate", "linearize", "sweep", "linesearch")
    ev = {k: [] for k in names}

    def step(timed):
        marks = []

        def mark():
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record()                               # current stream = the stream every kernel is launched on
                marks.append(e)
        solver.u.copy_(u0)
        solver.active.fill_(1)
        mark()
        ops.simulate(model, x0, solver.u, x=solver.x, cost=solver.cost)
        mark()
        ops.linearize(model, solver.x, solver.u, t_start=solver.t_start, layout=solver.layout, rec=solver.rec,
                      VxN=solver.VxN, VxxN=solver.VxxN)
        mark()
        if not hybrid:
            ops.riccati_sweep(solver.rec, solver.VxN, solver.VxxN, NX, NU, solver.layout, solver.reg, K=solver.K,
                              k=solver.k, status=solver.status, active=solver.active)
            mark()
        else:
            ops.riccati_sweep(solver.rec, solver.VxN, solver.VxxN, NX, NU, solver.layout, solver.reg, K=solver.K_seg,
                              k=solver.k_seg, status=solver.status, active=solver.active)
            mark()
            prompt = _pack_prompt(solver.k_seg, solver.K_seg)
            # x_err = x - x_ref + offset is formed by the kernel (shifted normalisation mean); prediction unpacked into K, k
            tf.predict_gains(solver.x, prompt, solver.K, solver.k, solver.active, x_shift=x_shift)
            mark()
            solver.k[:, N - 1:] = solver.k_seg                                    # the swept tail step (:517-518)
            solver.K[:, N - 1:] = solver.K_seg
            mark()
        ops.linesearch(model, solver.x, solver.u, solver.K, solver.k, solver.cost, solver.tol, solver.alphas,
                       alpha_idx=solver.alpha_idx, active=solver.active, iters=solver.iters)
        mark()
        if timed:
            for i, name in enumerate(names):
                ev[name].append((marks[i], marks[i + 1]))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    gather_buf = None
    if world > 1:                                        # warm the collective too (and keep its receive buffer)
        K_all, k_all = parallel.all_gather_gains(solver.K, solver.k, equal_shards=True)
        gather_buf = K_all._base if K_all._base is not None else None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    host_issue = time.perf_counter() - t0                # host time to enqueue the steps (GPU-bound when << elapsed)
    if world > 1:
        K_all, k_all = parallel.all_gather_gains(solver.K, solver.k, equal_shards=True, out=gather_buf)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # HBM traffic of the sweep from the PMC counters: rocprofv3 cannot run inside this process, so the number is
    # the per-launch value of the latest committed counter pass of THIS command (scripts/gpu_pmc.sh ->
    # profiles/*_pmc_hbm.json: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 FETCH_SIZE correction applied).
    traffic, traffic_src = None, None
    if B == BATCH_PER_GPU and not hybrid:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json")))[-1:]:
            try:
                with open(path) as fh:
                    kern = json.load(fh)["kernels"]
                traffic = float(next(v for k, v in kern.items() if k.startswith("sweep_tile16_kernel"))["hbm_bytes_corrected"])
                traffic_src = os.path.relpath(path, ROOT)
            except Exception:
                traffic = None

    # Outside the timed region: one solve of the same batch with the REAL exit test (per-trajectory
    # convergence, tol 1e-3, cold start u = 0 like the reference), and BASELINE configs[0] as a single-trajectory call
    # of the drop-in (its latency is host round trips, not kernels).
    extras = {}
    if rank == 0 and world == 1 and not hybrid and cpu is not None:
        from quattro_ilqr_amd import CartPoleMPC
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
                                     "flagged": int((res["status"] != 0).sum().item())}
        # the receding-horizon loop itself — B controllers, warm-started, plant = the device model
        from quattro_ilqr_amd import BatchedMPC
        mpc = BatchedMPC(model, N, max_iter=100, tol=1e-3, device=dev)
        mpc.run(x0, 2)
        mpc.u_warm = None
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        run = mpc.run(x0, 10)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t1
        extras["batched_mpc"] = {"controllers": B, "control_steps": 10, "wall_ms": 1e3 * wall,
                                 "ilqr_iterations_per_control_step_mean": float(run["iters"].double().mean().item()),
                                 "control_steps_per_s": B * 10 / wall}
        cp = CartPoleMPC(horizon=30, dt=0.01, integration_method="euler", ilqr_only=True, device=str(dev))
        cp.control_step(np.array([0.0, 0.0, 0.1, 0.0]))
        cp.ilqr.u = [np.zeros(1) for _ in range(30)]
        cp.ilqr.logs = []
        t1 = time.perf_counter()
        cp.control_step(np.array([0.0, 0.0, 0.1, 0.0]))
        extras["config1_cartpole_N30_B1"] = {"ms": 1e3 * (time.perf_counter() - t1), "iterations": len(cp.ilqr.logs)}

    kern_ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}
    accepted = float((solver.alpha_idx >= 0).float().mean().item())
    bad = int((solver.status != 0).sum().item())
    if rank == 0:
        total_steps = world * B * N * args.steps
        sweep_s = kern_ms["sweep"] * 1e-3
        achieved = B * SWEEP_BYTES_PER_TRAJ / sweep_s / 1e9
        if hybrid:
            tf_s = kern_ms["transformer"] * 1e-3
            tf_flops = 135.64e6 * B                      # 135.64 MFLOP per trajectory (L=101, full L x L attention counted)
            roof = {"kernel": "tf_forward_kernel<4> (quattro_tf_forward_bf16)", "bound": "mfma",
                    "achieved": tf_flops / tf_s / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                    "frac": tf_flops / tf_s / 1e12 / 2500.0, "algorithmic_flops_per_launch": tf_flops,
                    "avg_launch_ms": kern_ms["transformer"], "traffic": None}
            workload = ("quadrotor n_x=12 n_u=4 N=50, hybrid iteration (BASELINE configs[4]) = simulate + 1-step tail "
                        "linearize/sweep + bf16-MFMA transformer (L=101, d=128, 3 layers, random-init) + gain-stack "
                        "assembly + 6-alpha line search/commit")
        else:
            roof = {"kernel": "sweep_tile16_kernel<true> (quattro_riccati_sweep_f32, TILE16C records)", "bound": "hbm",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": B * SWEEP_BYTES_PER_TRAJ, "avg_launch_ms": kern_ms["sweep"],
                    "traffic": traffic, "traffic_source": traffic_src,
                    "note": "achieved = SURVEY 8(d) algorithmic bytes (1872 B/step + terminal) / measured launch time; "
                            "the HBM traffic is BELOW the algorithmic bytes because the constants of the problem are "
                            "kept once in a header record (L2-resident) and only 304 of a record's 1664 bytes stream per step"}
            workload = ("quadrotor n_x=12 n_u=4 N=50, pure iLQR iteration = simulate + linearize + Riccati sweep "
                        "+ 6-alpha line search/commit (BASELINE configs[2]; configs[3] when n_gpus=8)")
        out = {
            "metric": "iLQR iterations/sec (batch x horizon steps/s), quadrotor N=50 batch=4096",
            "value": total_steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "batch_per_gpu": B, "global_batch": world * B, "horizon": N, "n_x": NX, "n_u": NU,
                       "integrator": "euler", "dt": 0.01, "parallelism": f"dp{world} (independent trajectory shards"
                       + (", one all-gather of K/k)" if world > 1 else ")")},
            "iterations_per_s": world * B * args.steps / elapsed,
            "kernel_ms": kern_ms, "host_issue_ms_per_step": 1e3 * host_issue / args.steps,
            "accepted_fraction": accepted, "flagged_trajectories": bad,
            "roofline": roof,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu_all_cores"] = out["value"] / cpu["value"]
        if extras:
            out["extras"] = extras
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
