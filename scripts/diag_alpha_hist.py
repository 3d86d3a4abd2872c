"""How often does the line search accept alpha = 1 (candidate 0)?  Per iteration of a converged solve, cold and warm starts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from bench import synthetic_batch
dev = torch.device("cuda:0")
N, B = 50, 4096
md = q.quadrotor_model()
x0, u0 = synthetic_batch(B, 0)
for name, ui in (("cold start (u = 0)", None), ("hover + noise nominal", u0)):
    s = q.QuattroILQR(md, N, device=dev, device_loop=False, check_every=1)
    s._alloc(B)
    x0t = torch.as_tensor(x0, dtype=torch.float32, device=dev)
    if ui is None: s.u.zero_()
    else: s.u.copy_(torch.as_tensor(ui, dtype=torch.float32, device=dev))
    q.ops.simulate(md, x0t, s.u, x=s.x, cost=s.cost)
    s.active.fill_(1); s.iters.zero_(); s.alpha_idx.fill_(-1); s.status.zero_()
    hist = np.zeros(8, dtype=np.int64); waves_slow = 0; waves_tot = 0
    for it in range(40):
        act = s.active.clone()
        if int(act.sum()) == 0: break
        s.iterate(None)
        a = s.alpha_idx.cpu().numpy(); m = act.cpu().numpy() != 0
        for v in a[m]: hist[v] += 1
        pair_act = m.reshape(-1, 2).any(axis=1)
        slow = ((a > 0) & m).reshape(-1, 2).any(axis=1)
        waves_slow += int(slow.sum()); waves_tot += int(pair_act.sum())
        if it < 6 or it % 5 == 0:
            print(f"  it {it}: active {int(m.sum())}, accepted alpha idx counts {np.bincount(a[m] + 1, minlength=8)} (index 0 = none)")
    print(f"{name}: alpha-index histogram over all active (trajectory, iteration) pairs: {hist[:6]} none: {hist[-1]}; waves with a non-alpha-1 accept: {waves_slow} of {waves_tot} ({100.0 * waves_slow / max(1, waves_tot):.1f} %)")
