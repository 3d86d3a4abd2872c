"""Workload for rocprofv3 --kernel-trace --stats: the device-resident loops (one persistent launch each) next to the
host-driven loop's kernels, at the sizes bench.py's extras report."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from bench import synthetic_batch, synthetic_cartpole
dev = torch.device("cuda:0")
N = 50
for kind in ("quadrotor", "rk4", "cartpole"):
    md = q.quadrotor_model() if kind == "quadrotor" else (q.quadrotor_model(integrator="rk4") if kind == "rk4" else q.cartpole_model(dt=0.01, integrator="euler"))
    B = 1024 if kind == "cartpole" else 4096
    x0, u0 = (synthetic_cartpole if kind == "cartpole" else synthetic_batch)(B, 0)
    x0 = torch.as_tensor(x0, dtype=torch.float32, device=dev); u0 = torch.as_tensor(u0, dtype=torch.float32, device=dev)
    tol = 1e-1 if kind == "cartpole" else 1e-3
    s = q.QuattroILQR(md, N, device=dev, tol=tol)
    for _ in range(4):
        s.solve(x0, u0, max_iter=20, fixed_iters=True)       # 20 fixed iterations, one launch
    for _ in range(4):
        s.solve(x0)                                          # converged solve from the cold start
    mpc = q.BatchedMPC(md, N, max_iter=100, tol=tol, device=dev)
    for _ in range(3):
        mpc.u_warm = None
        mpc.run(x0, 10)
    torch.cuda.synchronize()
    print(kind, "done")
