"""Diagnostic: per-iteration wall time of a real-exit-test solve (active set shrinking)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
from quattro_ilqr_amd import QuattroILQR, ops, quadrotor_model
import bench
dev = "cuda:0"; B, N = 4096, 50
md = quadrotor_model()
x0h, _ = bench.synthetic_batch(B, 0)
x0 = torch.as_tensor(x0h, dtype=torch.float32, device=dev)
s = QuattroILQR(md, N, max_iter=100, tol=1e-3, device=dev)
s.solve(x0, max_iter=2)
torch.cuda.synchronize(); t0 = time.perf_counter(); r = s.solve(x0); torch.cuda.synchronize()
print("solve() wall %.2f ms, iters max %d" % (1e3 * (time.perf_counter() - t0), int(r["iters"].max())))
# manual loop with a sync per iteration
s._alloc(B); s.u.zero_(); ops.simulate(md, x0, s.u, x=s.x, cost=s.cost)
s.active.fill_(1); s.iters.zero_(); s.alpha_idx.fill_(-1); s.status.zero_()
for it in range(32):
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s.iterate(None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t1
    na = int(s.active.sum().item())
    print(f"it {it}: {dt*1e6:.0f} us, active after {na}, max|x| {float(s.x.abs().max()):.3g}")
    if na == 0: break

for ce in (1000, 8, 4, 1):
    sv = QuattroILQR(md, N, max_iter=29, tol=1e-3, device=dev, check_every=ce)
    sv.solve(x0, max_iter=2)
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = sv.solve(x0); torch.cuda.synchronize()
    print(f"check_every={ce}: solve() wall {1e3 * (time.perf_counter() - t0):.2f} ms")
# what does one convergence check cost?
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); n = int(s.active.sum().item()); t1 = time.perf_counter()
    a = s.active.sum(); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter(); v = a.item(); t4 = time.perf_counter()
    print(f"sum().item() {1e6*(t1-t0):.0f} us | sum issue {1e6*(t2-t1):.0f} us, sync {1e6*(t3-t2):.0f} us, item {1e6*(t4-t3):.0f} us")

for mi in (29, 33, 40, 100):
    sv = QuattroILQR(md, N, max_iter=mi, tol=1e-3, device=dev)
    sv.solve(x0, max_iter=2)
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = sv.solve(x0); torch.cuda.synchronize()
    print(f"max_iter={mi}: solve() wall {1e3 * (time.perf_counter() - t0):.2f} ms")
sv = QuattroILQR(md, N, max_iter=100, tol=1e-3, device=dev); sv.solve(x0, max_iter=2)
sv._alloc(B); sv.u.zero_(); ops.simulate(md, x0, sv.u, x=sv.x, cost=sv.cost)
sv.active.fill_(0)
for it in range(4):
    torch.cuda.synchronize(); t1 = time.perf_counter(); sv.iterate(None); torch.cuda.synchronize()
    print(f"all-inactive iterate: {1e6*(time.perf_counter()-t1):.0f} us")
