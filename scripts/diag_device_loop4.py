import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from quattro_ilqr_amd import ops
from bench import synthetic_batch
dev = torch.device("cuda:0")
N, B, steps = 50, 4096, 10
md = q.quadrotor_model()
x0, _ = synthetic_batch(B, 0)
x0 = torch.as_tensor(x0, dtype=torch.float32, device=dev)
sv = q.QuattroILQR(md, N, max_iter=100, tol=1e-3, device=dev)
sv._alloc(B)
ws = ops.workspace(md, B, N, dev)
traj_x = torch.empty((B, steps + 1, 12), dtype=torch.float32, device=dev)
traj_u = torch.empty((B, steps, 4), dtype=torch.float32, device=dev)
traj_it = torch.empty((B, steps), dtype=torch.int32, device=dev)
for rep in range(4):
    sv.u.zero_(); x_cur = x0.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter()
    e0.record()
    ops.mpc_run(md, x_cur, sv.x, sv.u, sv.K, sv.k, sv.cost, sv.tol, 100, steps, ws, traj_x, traj_u, traj_it, alphas=sv.alphas, reg=sv.reg,
                alpha_idx=sv.alpha_idx, active=sv.active, iters=sv.iters, status=sv.status)
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: host call {1e3*(t1-t):.3f} ms, events {e0.elapsed_time(e1):.3f} ms, wall {1e3*(t2-t):.3f} ms")
# the same solve via solve(): cold 1 step
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    sv.solve(x0)
    torch.cuda.synchronize(); print(f"solve: {1e3*(time.perf_counter()-t):.3f} ms")
