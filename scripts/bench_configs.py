"""Per-kernel timing of the non-headline BASELINE configs (diagnostic; the bench line is bench.py).
configs[1]: cart-pole n=4 m=1 N=50 B=1024 (pure iteration, generic LDS sweep)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
from quattro_ilqr_amd import QuattroILQR, cartpole_model, ops

def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

dev = "cuda:0"
for B in (1024, 16384):
    N = 50
    md = cartpole_model()
    rng = np.random.default_rng(1234)
    x0 = np.stack([rng.uniform(-0.5, 0.5, B), np.zeros(B), rng.uniform(-0.5, 0.5, B), np.zeros(B)], axis=1)
    s = QuattroILQR(md, N, device=dev); s._alloc(B); s.ensure_records()
    x0t = torch.as_tensor(x0, dtype=torch.float32, device=dev)
    s.u.zero_(); ops.simulate(md, x0t, s.u, x=s.x, cost=s.cost)
    t_sim = timed(lambda: ops.simulate(md, x0t, s.u, x=s.x, cost=s.cost))
    t_lin = timed(lambda: ops.linearize(md, s.x, s.u, layout=s.layout, rec=s.rec, VxN=s.VxN, VxxN=s.VxxN))
    t_swp = timed(lambda: ops.riccati_sweep(s.rec, s.VxN, s.VxxN, 4, 1, s.layout, s.reg, K=s.K, k=s.k, status=s.status))
    xs, us, cs = s.x.clone(), s.u.clone(), s.cost.clone()
    def ls():
        s.x.copy_(xs); s.u.copy_(us); s.cost.copy_(cs); s.active.fill_(1)
        ops.linesearch(md, s.x, s.u, s.K, s.k, s.cost, 1e-1, ops.ALPHAS, alpha_idx=s.alpha_idx, active=s.active, iters=s.iters)
    def cp():
        s.x.copy_(xs); s.u.copy_(us); s.cost.copy_(cs); s.active.fill_(1)
    t_ls = timed(ls) - timed(cp)
    tot = t_sim + t_lin + t_swp + t_ls
    print(f"cart-pole N=50 B={B}: simulate {t_sim:.1f} linearize {t_lin:.1f} sweep {t_swp:.1f} linesearch {t_ls:.1f} us"
          f" -> {tot:.1f} us/iteration = {B*N/tot:.1f} M steps/s (upper bounds: back-to-back launches, host-bound below ~30 us)")

# eager launches vs hipGraph replay of one iteration, whole solves of 20 fixed iterations
import time
from quattro_ilqr_amd import quadrotor_model
import bench
for name, md, B, N in (("cart-pole", cartpole_model(), 1024, 50), ("quadrotor", quadrotor_model(), 256, 50),
                       ("quadrotor", quadrotor_model(), 4096, 50)):
    rng = np.random.default_rng(0)
    x0 = np.asarray(md.x_ref) + 0.2 * rng.standard_normal((B, md.n)) * (1.0 if name == "cart-pole" else 0.1)
    for use_graph in (False, True):
        s = QuattroILQR(md, N, device=dev, use_graph=use_graph)
        s.solve(x0, max_iter=3, fixed_iters=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): s.solve(x0, max_iter=20, fixed_iters=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (5 * 20)
        print(f"{name} B={B} N={N} {'graph' if use_graph else 'eager'}: {dt*1e6:.1f} us per iteration ({B*N/dt/1e6:.1f} M steps/s)")
