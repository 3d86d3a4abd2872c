"""Diagnostic: the persistent kernel alone — 20 fixed iterations of 4096 quadrotor trajectories in ONE launch (per iteration), a
converged solve of the same batch, and a lone trajectory's solve (B = 1).  Honours QUATTRO_HIP_LIB (A/B of library builds).
usage: time_device_loop.py [euler|rk4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
from quattro_ilqr_amd import QuattroILQR, quadrotor_model
import bench
dev = "cuda:0"; N = 50
integ = sys.argv[1] if len(sys.argv) > 1 else "euler"
md = quadrotor_model(dt=0.01, integrator=integ)


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); best.append((time.perf_counter() - t0) / reps)
    return 1e3 * float(np.median(best))


out = []
for B in (4096, 1):
    x0h, u0h = bench.synthetic_batch(B, 0)
    x0 = torch.as_tensor(x0h, dtype=torch.float32, device=dev); u0 = torch.as_tensor(u0h, dtype=torch.float32, device=dev)
    s = QuattroILQR(md, N, device=dev)
    for _ in range(30): s.solve(x0, u0, max_iter=20, fixed_iters=True)      # clocks
    fixed = timed(lambda: s.solve(x0, u0, max_iter=20, fixed_iters=True), 10) / 20
    r = s.solve(x0, u0, max_iter=100)
    conv = timed(lambda: s.solve(x0, u0, max_iter=100), 10)
    out.append(f"B={B}: fixed-iteration loop {1e3 * fixed:.1f} us/iteration, converged solve {conv:.3f} ms "
               f"({float(r['iters'].float().mean()):.1f} iterations mean, {int(r['iters'].max())} max)")
print(integ, "|", " | ".join(out))
