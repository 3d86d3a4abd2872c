"""Experiment: does running the two halves of the batch as two independent solvers on two HIP streams
(sweep of one half beside the line search of the other) beat one solver on the whole batch?  Pure iteration, quadrotor
N = 50, B = 4096, fixed iteration count, same synthetic nominal as bench.py."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import torch
import bench
from quattro_ilqr_amd import QuattroILQR, ops, quadrotor_model
dev = torch.device("cuda:0"); B, N, IT = 4096, 50, 60
md = quadrotor_model()
x0h, u0h = bench.synthetic_batch(B, 0)
x0 = torch.as_tensor(x0h, dtype=torch.float32, device=dev); u0 = torch.as_tensor(u0h, dtype=torch.float32, device=dev)

def prep(sl):
    s = QuattroILQR(md, N, device=dev)
    s._alloc(sl.stop - sl.start)
    s.u.copy_(u0[sl]); ops.simulate(md, x0[sl].contiguous(), s.u, x=s.x, cost=s.cost); s.active.fill_(1)
    return s

def run_one():
    s = prep(slice(0, B))
    for _ in range(5): s.active.fill_(1); s.iterate()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(IT): s.active.fill_(1); s.iterate()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / IT * 1e3

def run_two(stagger):
    a, b = prep(slice(0, B // 2)), prep(slice(B // 2, B))
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    def it():
        with torch.cuda.stream(sa): a.active.fill_(1); a.iterate()
        with torch.cuda.stream(sb): b.active.fill_(1); b.iterate()
    if stagger:                       # half an iteration of head start for stream a: its line search meets b's sweep
        with torch.cuda.stream(sa):
            ops.linearize_sweep(md, a.x, a.u, 0, a.reg, K=a.K, k=a.k, status=a.status, active=a.active)
    for _ in range(5): it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(IT): it()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / IT * 1e3

for rep in range(2):
    print(f"one solver, B = {B}: {run_one():.4f} ms/iteration | two half-batch solvers on two streams: {run_two(False):.4f} ms "
          f"| the same, staggered by one sweep: {run_two(True):.4f} ms")
