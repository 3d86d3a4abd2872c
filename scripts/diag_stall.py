"""Diagnostic: where does the host stall when solve() is fed host (numpy) inputs back to back?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
from quattro_ilqr_amd import QuattroILQR, ops, quadrotor_model
dev = "cuda:0"; B, N = 4096, 50
md = quadrotor_model()
rng = np.random.default_rng(0)
x0n = np.asarray(md.x_ref) + 0.02 * rng.standard_normal((B, 12))
sv = QuattroILQR(md, N, device=dev)
sv.solve(x0n, max_iter=3, fixed_iters=True)
torch.cuda.synchronize()
pin = torch.empty((B, 12), dtype=torch.float32, pin_memory=True)
dst = torch.empty((B, 12), dtype=torch.float32, device=dev)
def burst(k):
    for _ in range(k):
        sv.iterate(None)
for mode in ("none", "pageable", "pinned_async", "event_sync", "tiny_tensor"):
    torch.cuda.synchronize()
    ts = []
    for rep in range(4):
        t0 = time.perf_counter()
        burst(40)
        t1 = time.perf_counter()
        if mode == "pageable":
            dst.copy_(torch.as_tensor(x0n, dtype=torch.float32))
        elif mode == "pinned_async":
            dst.copy_(pin, non_blocking=True)
        elif mode == "event_sync":
            e = torch.cuda.Event(); e.record(); e.synchronize()
        elif mode == "tiny_tensor":
            torch.tensor([1.0, 2.0], device=dev)
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(mode, " | ".join(f"issue {a*1e3:.2f} op {b*1e3:.2f}" for a, b in ts), f"| total {sum(a+b for a,b in ts)*1e3 + 0:.1f} ms, drained at +{(t3-t2)*1e3:.1f}")
