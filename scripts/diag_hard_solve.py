"""Diagnostic: per-kernel time inside a hard multi-iteration solve (large random initial errors on all 12 states)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
from quattro_ilqr_amd import QuattroILQR, ops, quadrotor_model
dev = "cuda:0"; B, N = 4096, 50
md = quadrotor_model()
rng = np.random.default_rng(0)
x0 = torch.as_tensor(np.asarray(md.x_ref) + float(os.environ.get("PERT", "0.02")) * rng.standard_normal((B, 12)), dtype=torch.float32, device=dev)
s = QuattroILQR(md, N, device=dev); s._alloc(B); s.ensure_records(); s.u.zero_()
ops.simulate(md, x0, s.u, x=s.x, cost=s.cost)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for it in range(int(os.environ.get("ITS", "12"))):
    s.active.fill_(1)
    e0 = ev(); ops.linearize(md, s.x, s.u, layout=s.layout, rec=s.rec, VxN=s.VxN, VxxN=s.VxxN)
    e1 = ev(); ops.riccati_sweep(s.rec, s.VxN, s.VxxN, 12, 4, s.layout, s.reg, K=s.K, k=s.k, status=s.status, active=s.active)
    e2 = ev(); ops.linesearch(md, s.x, s.u, s.K, s.k, s.cost, 1e-3, ops.ALPHAS, alpha_idx=s.alpha_idx, active=s.active, iters=s.iters)
    e3 = ev(); torch.cuda.synchronize()
    ai = s.alpha_idx.cpu().numpy()
    print(f"it {it}: linearize {e0.elapsed_time(e1)*1e3:.0f} sweep {e1.elapsed_time(e2)*1e3:.0f} linesearch {e2.elapsed_time(e3)*1e3:.0f} us | "
          f"alpha idx hist {np.bincount(ai + 1, minlength=7)} | max|x| {float(s.x.abs().max()):.3g} finite cost {bool(torch.isfinite(s.cost).all())} status!=0 {int((s.status != 0).sum())}")

# ---- slope/intercept of solve() wall time against the iteration count, eager and graph
import time
x0n = x0.double().cpu().numpy()
for use_graph in (False, True):
    sv = QuattroILQR(md, N, device=dev, use_graph=use_graph)
    sv.solve(x0n, max_iter=3, fixed_iters=True)
    for iters in (1, 10, 20, 40, 80):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): sv.solve(x0n, max_iter=iters, fixed_iters=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        t1 = time.perf_counter()
        for _ in range(3): sv.solve(x0, max_iter=iters, fixed_iters=True)
        torch.cuda.synchronize(); dt2 = (time.perf_counter() - t1) / 3
        print(f"graph={use_graph} iters={iters}: solve {dt*1e3:.2f} ms (numpy x0) {dt2*1e3:.2f} ms (device x0) -> {dt2/iters*1e6:.0f} us/iter")
