"""Device-resident loop against the host-driven loop across batch sizes beyond one residency round."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from bench import synthetic_batch
dev = torch.device("cuda:0")
N = 50
md = q.quadrotor_model()
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / reps
for B in (3072, 4096, 4352, 5120, 6144, 8192, 16384):
    x0, u0 = synthetic_batch(B, 0)
    x0 = torch.as_tensor(x0, dtype=torch.float32, device=dev); u0 = torch.as_tensor(u0, dtype=torch.float32, device=dev)
    out = []
    for loop in (True, False):
        s = q.QuattroILQR(md, N, device=dev, device_loop=loop)
        for it in (5, 20):
            out.append(timed(lambda: s.solve(x0, u0, max_iter=it, fixed_iters=True)))
    print(f"B={B:6d}: device loop 5 it {out[0]:7.3f} ms, 20 it {out[1]:7.3f} ms ({(out[1]-out[0])/15*1e3:6.1f} us/it) | host loop 5 it {out[2]:7.3f}, 20 it {out[3]:7.3f} ({(out[3]-out[2])/15*1e3:6.1f} us/it)")
