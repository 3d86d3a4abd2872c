import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from bench import synthetic_batch
dev = torch.device("cuda:0")
N = 50
md = q.quadrotor_model()
def run(B, steps, reps=3):
    x0, _ = synthetic_batch(4096, 0)
    x0 = torch.as_tensor(x0[:B], dtype=torch.float32, device=dev)
    mpc = q.BatchedMPC(md, N, max_iter=100, tol=1e-3, device=dev)
    mpc.run(x0, steps); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        mpc.u_warm = None
        torch.cuda.synchronize(); t = time.perf_counter()
        out = mpc.run(x0, steps)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    it = out["iters"].cpu().numpy()
    pair = np.maximum(it[0::2], it[1::2]) if B > 1 else it
    return min(ts), it, pair
for B in (2, 64, 4096):
    for steps in (1, 2, 5, 10):
        ms, it, pair = run(B, steps)
        print(f"B={B:5d} steps={steps:2d}: {ms:7.3f} ms; mean iters/traj {it.sum(axis=1).mean():6.1f}; slowest pair {pair.sum(axis=1).max():3d} iterations -> {1e3*ms/pair.sum(axis=1).max():6.1f} us per iteration of the slowest pair")
