import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
from quattro_ilqr_amd import QuattroILQR, ops, quadrotor_model
import bench
dev = "cuda:0"; B, N = 4096, 50
md = quadrotor_model(integrator="rk4")
x0h, u0h = bench.synthetic_batch(B, 0)
x0 = torch.as_tensor(x0h, dtype=torch.float32, device=dev); u0 = torch.as_tensor(u0h, dtype=torch.float32, device=dev)
s = QuattroILQR(md, N, device=dev); s._alloc(B)
s.u.copy_(u0); ops.simulate(md, x0, s.u, x=s.x, cost=s.cost)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print(os.environ.get("QUATTRO_HIP_LIB", "shipped"), "fused RK4 linearize+sweep: %.1f us" % t(lambda: ops.linearize_sweep(md, s.x, s.u, 0, s.reg, K=s.K, k=s.k, status=s.status, scratch=s._sweep_scratch)))
