import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from bench import synthetic_batch
dev = torch.device("cuda:0")
md = q.quadrotor_model()
x0, _ = synthetic_batch(4096, 0)
x0 = torch.as_tensor(x0, dtype=torch.float32, device=dev)
for dl in (True, False, False, True):
    sv = q.QuattroILQR(md, 50, max_iter=100, tol=1e-3, device=dev, device_loop=dl)
    sv.solve(x0, max_iter=9); torch.cuda.synchronize()
    ts = []
    for r in range(6):
        t = time.perf_counter(); out = sv.solve(x0); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    print("device_loop", dl, [round(v, 2) for v in ts], "iters max", int(out["iters"].max()))
