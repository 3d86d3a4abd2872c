"""Diagnostic timing of individual ops at the headline shape (not a test)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
from quattro_ilqr_amd import QuattroILQR, ops, quadrotor_model
import bench
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 50
dev = "cuda:0"
md = quadrotor_model()
x0h, u0h = bench.synthetic_batch(B, 0)
x0 = torch.as_tensor(x0h, dtype=torch.float32, device=dev); u0 = torch.as_tensor(u0h, dtype=torch.float32, device=dev)
s = QuattroILQR(md, N, device=dev); s._alloc(B)
s.u.copy_(u0); ops.simulate(md, x0, s.u, x=s.x, cost=s.cost); s.backward()
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print("rollout 6 alphas, costs only      : %.1f us" % t(lambda: ops.rollout(md, s.x, s.u, s.K, s.k, ops.ALPHAS)))
print("rollout 1 alpha,  costs only      : %.1f us" % t(lambda: ops.rollout(md, s.x, s.u, s.K, s.k, (1.0,))))
print("rollout 6 alphas, with trajectories: %.1f us" % t(lambda: ops.rollout(md, s.x, s.u, s.K, s.k, ops.ALPHAS, want_traj=True)))
xs, us, cs = s.x.clone(), s.u.clone(), s.cost.clone()
def ls():
    s.x.copy_(xs); s.u.copy_(us); s.cost.copy_(cs); s.active.fill_(1)
    ops.linesearch(md, s.x, s.u, s.K, s.k, s.cost, 1e-3, ops.ALPHAS, alpha_idx=s.alpha_idx, active=s.active, iters=s.iters)
def cp():
    s.x.copy_(xs); s.u.copy_(us); s.cost.copy_(cs); s.active.fill_(1)
print("linesearch (+4 small copies)       : %.1f us  (copies alone %.1f us)" % (t(ls), t(cp)))
print("simulate                           : %.1f us" % t(lambda: ops.simulate(md, x0, s.u, x=s.x, cost=s.cost)))
print("total_cost                         : %.1f us" % t(lambda: ops.total_cost(md, s.x, s.u)))

# ---- the same ops timed in the bench's sequence (linearize + sweep first: 341 MB of records stream through the caches)
def in_sequence(fn, n=20):
    tot = 0.0
    for i in range(n + 3):
        s.x.copy_(xs); s.u.copy_(us); s.cost.copy_(cs); s.active.fill_(1)
        s.backward()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 3: tot += e0.elapsed_time(e1)
    return tot / n * 1e3
print("after linearize+sweep: rollout costs only %.1f us | with trajectories %.1f us | linesearch %.1f us | simulate %.1f us" % (
    in_sequence(lambda: ops.rollout(md, s.x, s.u, s.K, s.k, ops.ALPHAS)),
    in_sequence(lambda: ops.rollout(md, s.x, s.u, s.K, s.k, ops.ALPHAS, want_traj=True)),
    in_sequence(lambda: ops.linesearch(md, s.x, s.u, s.K, s.k, s.cost, 1e-3, ops.ALPHAS, alpha_idx=s.alpha_idx, active=s.active, iters=s.iters)),
    in_sequence(lambda: ops.simulate(md, x0, s.u, x=s.x, cost=s.cost))))
