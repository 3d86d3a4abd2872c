"""Diagnostic: per-kernel times of one quadrotor iteration with the RK4 integrator (the MPC classes' default; the shipped
simulators and the headline use Euler)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
from quattro_ilqr_amd import QuattroILQR, ops, quadrotor_model
import bench
dev = "cuda:0"; B, N = 4096, 50
for integ in ("euler", "rk4"):
    md = quadrotor_model(integrator=integ)
    x0h, u0h = bench.synthetic_batch(B, 0)
    x0 = torch.as_tensor(x0h, dtype=torch.float32, device=dev); u0 = torch.as_tensor(u0h, dtype=torch.float32, device=dev)
    s = QuattroILQR(md, N, device=dev); s._alloc(B); s.ensure_records()
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
    s.u.copy_(u0)
    t_sim = t(lambda: ops.simulate(md, x0, s.u, x=s.x, cost=s.cost))
    t_lin = t(lambda: ops.linearize(md, s.x, s.u, layout=s.layout, rec=s.rec, VxN=s.VxN, VxxN=s.VxxN))
    t_swp = t(lambda: ops.riccati_sweep(s.rec, s.VxN, s.VxxN, 12, 4, s.layout, s.reg, K=s.K, k=s.k, status=s.status))
    t_fused = t(lambda: ops.linearize_sweep(md, s.x, s.u, 0, s.reg, K=s.K, k=s.k, status=s.status, scratch=s._sweep_scratch))
    xs, us, cs = s.x.clone(), s.u.clone(), s.cost.clone()
    def ls():
        s.x.copy_(xs); s.u.copy_(us); s.cost.copy_(cs); s.active.fill_(1)
        ops.linesearch(md, s.x, s.u, s.K, s.k, s.cost, 1e-3, ops.ALPHAS, alpha_idx=s.alpha_idx, active=s.active, iters=s.iters)
    def cp():
        s.x.copy_(xs); s.u.copy_(us); s.cost.copy_(cs); s.active.fill_(1)
    t_ls = t(ls) - t(cp)
    tot = t_sim + t_lin + t_swp + t_ls
    print(f"{integ}: layout {s.layout} simulate {t_sim:.1f} linearize {t_lin:.1f} sweep {t_swp:.1f} linesearch {t_ls:.1f} us -> {tot:.1f} us/iteration, {B*N/tot:.0f} M steps/s"
          f" | fused linearize+sweep {t_fused:.1f} us -> {t_sim + t_fused + t_ls:.1f} us/iteration")
