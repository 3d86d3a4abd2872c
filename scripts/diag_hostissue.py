"""Diagnostic: host time per op call in the bench loop, with and without a forked multiprocessing pool beforehand."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
def work(i):
    import numpy as np
    a = np.random.default_rng(i).standard_normal((200, 200)); return float(np.linalg.norm(a @ a))
if mode in ("pool", "pool_oracle"):
    import multiprocessing as mp
    import bench
    ctx = mp.get_context("fork")
    with ctx.Pool(64) as pool:
        if mode == "pool_oracle":
            pool.map(bench._cpu_warm, range(64)); pool.map(bench._cpu_worker, [(9000 + i, 1) for i in range(64)], chunksize=1)
        else:
            pool.map(work, range(256))
import torch
from quattro_ilqr_amd import QuattroILQR, ops, quadrotor_model
import bench
dev = torch.device("cuda", 0); B, N = 4096, 50
md = quadrotor_model()
solver = QuattroILQR(md, N, device=dev); solver._alloc(B); solver.ensure_records()
x0h, u0h = bench.synthetic_batch(B, 0)
x0 = torch.as_tensor(x0h, dtype=torch.float32, device=dev); u0 = torch.as_tensor(u0h, dtype=torch.float32, device=dev)
names = ["copy", "fill", "simulate", "linearize", "sweep", "linesearch"]
acc = {k: 0.0 for k in names}
def step(record):
    t = [time.perf_counter()]
    solver.u.copy_(u0); t.append(time.perf_counter())
    solver.active.fill_(1); t.append(time.perf_counter())
    ops.simulate(md, x0, solver.u, x=solver.x, cost=solver.cost); t.append(time.perf_counter())
    ops.linearize(md, solver.x, solver.u, layout=solver.layout, rec=solver.rec, VxN=solver.VxN, VxxN=solver.VxxN); t.append(time.perf_counter())
    ops.riccati_sweep(solver.rec, solver.VxN, solver.VxxN, 12, 4, solver.layout, solver.reg, K=solver.K, k=solver.k, status=solver.status, active=solver.active); t.append(time.perf_counter())
    ops.linesearch(md, solver.x, solver.u, solver.K, solver.k, solver.cost, solver.tol, solver.alphas, alpha_idx=solver.alpha_idx, active=solver.active, iters=solver.iters); t.append(time.perf_counter())
    if record:
        for i, k in enumerate(names): acc[k] += t[i + 1] - t[i]
for _ in range(5): step(False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step(True)
issue = time.perf_counter() - t0
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(mode, f"issue {issue/50*1e3:.3f} ms/step, total {tot/50*1e3:.3f} ms/step |", " ".join(f"{k} {v/50*1e6:.0f}us" for k, v in acc.items()))
