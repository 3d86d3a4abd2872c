"""Single-trajectory latency of the RK4 quadrotor (the default integrator of the reference's QuadrotorMPC): persistent kernel (fused
RK4 sweep) against enqueued iterations (records path), per iteration, B = 1 and a few small batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from quattro_ilqr_amd import ops
dev = "cuda:0"
for integ in ("euler", "rk4"):
    md = q.quadrotor_model(integrator=integ)
    for N in (30, 50):
        for B in (1, 64, 512):
            x0 = np.tile(np.asarray(md.x_ref), (B, 1)); x0[:, 6] = 0.1
            x0 = torch.as_tensor(x0, dtype=torch.float32, device=dev)
            res = {}
            for name, kw in (("persistent", dict(device_loop=True)), ("enqueued", dict(device_loop=False, check_every=1000))):
                sv = q.QuattroILQR(md, N, max_iter=10, tol=1e-3, device=dev, **kw)
                sv.solve(x0, max_iter=10, fixed_iters=True)
                torch.cuda.synchronize()
                ts = []
                for _ in range(7):
                    t = time.perf_counter(); sv.solve(x0, max_iter=10, fixed_iters=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
                res[name] = 1e6 * float(np.median(ts)) / 10
            print(f"{integ} N={N} B={B}: us per iteration persistent {res['persistent']:.1f} enqueued {res['enqueued']:.1f}")
