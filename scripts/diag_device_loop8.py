import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from quattro_ilqr_amd import ops, _lib
from bench import synthetic_batch
dev = torch.device("cuda:0")
N, steps, B = 50, 10, 4096
md = q.quadrotor_model()
lib = _lib.load()
lib.quattro_debug_set_solve_stamps.argtypes = [ctypes.c_void_p]
x0a, _ = synthetic_batch(B, 0)
x0 = torch.as_tensor(x0a, dtype=torch.float32, device=dev)
def mk(Bx):
    sv = q.QuattroILQR(md, N, max_iter=100, tol=1e-3, device=dev); sv._alloc(Bx)
    return dict(sv=sv, ws=ops.workspace(md, Bx, N, dev), tx=torch.empty((Bx, steps + 1, 12), device=dev), tu=torch.empty((Bx, steps, 4), device=dev),
                ti=torch.empty((Bx, steps), dtype=torch.int32, device=dev), st=torch.zeros(((Bx + 1) // 2, 2 * (steps + 1)), dtype=torch.int64, device=dev),
                xc=torch.empty((Bx, 12), device=dev))
def launch(d, x0_, MI):
    sv = d["sv"]
    lib.quattro_debug_set_solve_stamps(ctypes.c_void_p(d["st"].data_ptr()))
    ops.mpc_run(md, d["xc"], sv.x, sv.u, sv.K, sv.k, sv.cost, sv.tol, MI, steps, d["ws"], d["tx"], d["tu"], d["ti"], alphas=sv.alphas, reg=sv.reg,
                alpha_idx=sv.alpha_idx, active=sv.active, iters=sv.iters, status=sv.status)
main, pre, post = mk(B), mk(2), mk(2)
for rep in range(5):
    for d, xx in ((main, x0), (pre, x0[:2]), (post, x0[:2])):
        d["sv"].u.zero_(); d["xc"].copy_(xx)
    torch.cuda.synchronize()
    launch(pre, x0[:2], 0); launch(main, x0, 100); launch(post, x0[:2], 0)
    torch.cuda.synchronize()
    a, m, p = (d["st"].cpu().numpy() for d in (pre, main, post))
    print(f"rep {rep}: pre kernel end -> first main workgroup start {(m[:,0].min() - a[:,-2].max())/100:.1f} us; main first start -> last end {(m[:,-2].max() - m[:,0].min())/1e5:.3f} ms; "
          f"last main end -> post kernel start {(p[:,0].min() - m[:,-2].max())/100:.1f} us")
