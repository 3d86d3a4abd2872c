"""Diagnostics of the device-resident loops: lone-workgroup iteration latency, iteration statistics of the MPC run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
sys.path.insert(0, ROOT)
from bench import synthetic_batch, synthetic_cartpole

dev = torch.device("cuda:0")
N = 50
kind = sys.argv[1] if len(sys.argv) > 1 else "quadrotor"
md = q.quadrotor_model() if kind == "quadrotor" else (q.quadrotor_model(integrator="rk4") if kind == "rk4" else q.cartpole_model(dt=0.01, integrator="euler"))
syn = synthetic_cartpole if kind == "cartpole" else synthetic_batch


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / reps


for B in (2, 64, 512, 2048, 4096, 8192):
    x0, u0 = syn(B, 0)
    x0 = torch.as_tensor(x0, dtype=torch.float32, device=dev); u0 = torch.as_tensor(u0, dtype=torch.float32, device=dev)
    s = q.QuattroILQR(md, N, device=dev, tol=1e-1 if kind == "cartpole" else 1e-3)
    ms = timed(lambda: s.solve(x0, u0, max_iter=20, fixed_iters=True))
    ms1 = timed(lambda: s.solve(x0, u0, max_iter=1, fixed_iters=True))
    print(f"{kind} B={B:5d}: fixed 20 iterations {ms:7.3f} ms  -> {(ms - ms1) / 19 * 1e3:7.1f} us per iteration (1 iteration: {ms1 * 1e3:6.1f} us)")

B = 4096
x0, _ = syn(B, 0)
x0 = torch.as_tensor(x0, dtype=torch.float32, device=dev)
for loop in (True, False):
    mpc = q.BatchedMPC(md, N, max_iter=100, tol=1e-1 if kind == "cartpole" else 1e-3, device=dev)
    mpc.run(x0, 2, device_loop=loop); mpc.u_warm = None
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = mpc.run(x0, 10, device_loop=loop)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t)
    it = out["iters"].cpu().numpy()
    pair = np.maximum(it[0::2], it[1::2])
    print(f"mpc device_loop={loop}: {ms:.2f} ms; iters per step mean {it.mean():.2f}; per-step max over batch {it.max(axis=0)}; "
          f"sum of per-step max {it.max(axis=0).sum()}; max over trajectories of the sum {it.sum(axis=1).max()}; "
          f"max over workgroup pairs of sum of pair-max {pair.sum(axis=1).max()}; mean sum {it.sum(axis=1).mean():.1f}")
s = q.QuattroILQR(md, N, max_iter=100, device=dev, tol=1e-1 if kind == "cartpole" else 1e-3)
ms = timed(lambda: s.solve(x0))
it = s.iters.cpu().numpy()
print(f"converged solve: {ms:.2f} ms, iters mean {it.mean():.2f} max {it.max()} ; hist {np.bincount(it)[:40]}")
