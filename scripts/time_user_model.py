"""Iteration time of a user-compiled model through the generic kernels, with per-kernel times (run under rocprofv3 for those)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from quattro_ilqr_amd import user_model

dev = torch.device("cuda:0")
N = 50
um = user_model.example_planar_model()
for B in (1, 64, 1024, 4096, 16384):
    rng = np.random.default_rng(0)
    x0 = torch.as_tensor(np.asarray(um.x_ref) + rng.normal(0, 0.3, (B, 6)) * np.array([1, 1, 0.3, 0.5, 0.5, 0.5]), dtype=torch.float32, device=dev)
    u0 = torch.as_tensor(np.full((B, N, 2), 9.81 / 2) + rng.normal(0, 0.2, (B, N, 2)), dtype=torch.float32, device=dev)
    out = []
    for loop in (True, False):                      # the persistent kernel of the model's library / enqueued iterations
        s = q.QuattroILQR(um, N, device=dev, device_loop="always" if loop else False, check_every=1000)
        s.solve(x0, u0, max_iter=20, fixed_iters=True); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); s.solve(x0, u0, max_iter=20, fixed_iters=True); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"planar user model B={B}: device-resident loop {out[0]:.1f} us per iteration, host-driven {out[1]:.1f}")
