"""Per-call durations (HIP events) of the device-resident solve at batch sizes where occasional slow runs were seen."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from bench import synthetic_batch, synthetic_cartpole

dev = torch.device("cuda:0")
N = 50
kind = sys.argv[1] if len(sys.argv) > 1 else "quadrotor"
sizes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2048, 4096, 8192]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
md = q.quadrotor_model() if kind == "quadrotor" else (q.quadrotor_model(integrator="rk4") if kind == "rk4" else q.cartpole_model(dt=0.01, integrator="euler"))
syn = synthetic_cartpole if kind == "cartpole" else synthetic_batch
for B in sizes:
    x0, u0 = syn(B, 0)
    x0 = torch.as_tensor(x0, dtype=torch.float32, device=dev); u0 = torch.as_tensor(u0, dtype=torch.float32, device=dev)
    s = q.QuattroILQR(md, N, device=dev, tol=1e-1 if kind == "cartpole" else 1e-3)
    s.solve(x0, u0, max_iter=iters, fixed_iters=True); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
    import time
    host = []
    ev[0].record()
    for r in range(12):
        t = time.perf_counter()
        s.solve(x0, u0, max_iter=iters, fixed_iters=True)
        ev[r + 1].record()
        host.append(1e3 * (time.perf_counter() - t))
    torch.cuda.synchronize()
    print(f"{kind} B={B}: device " + " ".join(f"{ev[r].elapsed_time(ev[r + 1]):.2f}" for r in range(12)) +
          " | host call " + " ".join(f"{h:.2f}" for h in host), flush=True)
