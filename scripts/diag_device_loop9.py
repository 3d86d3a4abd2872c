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
sv = q.QuattroILQR(md, N, max_iter=100, tol=1e-3, device=dev); sv._alloc(B)
ws = ops.workspace(md, B, N, dev)
tx = torch.empty((B, steps + 1, 12), device=dev); tu = torch.empty((B, steps, 4), device=dev); ti = torch.empty((B, steps), dtype=torch.int32, device=dev)
C = 2 * (steps + 1) + 2
stamps = torch.zeros((B // 2, C), dtype=torch.int64, device=dev)
lib.quattro_debug_set_solve_stamps(ctypes.c_void_p(stamps.data_ptr()))
for rep in range(3):
    sv.u.zero_(); xc = x0.clone(); stamps.zero_()
    torch.cuda.synchronize()
    ops.mpc_run(md, xc, sv.x, sv.u, sv.K, sv.k, sv.cost, sv.tol, 100, steps, ws, tx, tu, ti, alphas=sv.alphas, reg=sv.reg,
                alpha_idx=sv.alpha_idx, active=sv.active, iters=sv.iters, status=sv.status)
    torch.cuda.synchronize()
    st = stamps.cpu().numpy(); t0 = st[:, 0].min()
    last_step = (st[:, 2 * steps] - t0) / 100; w0 = (st[:, 1] - t0) / 100; w1 = (st[:, 2 * (steps + 1)] - t0) / 100
    passes1 = st[:, 2 * (steps + 1) + 1]; passes0 = st[:, 3:2 * (steps + 1):2].sum(axis=1)
    late = np.argsort(w1)[-4:]
    print(f"rep {rep}: last step stamp max {last_step.max():.0f} us; wave0 exit max {w0.max():.0f}; wave1 exit max {w1.max():.0f}; workgroups with wave1 passes != wave0 passes: {(passes1 != passes0).sum()}")
    for w in late:
        print(f"   wg {w}: last step {last_step[w]:.0f} wave0 exit {w0[w]:.0f} wave1 exit {w1[w]:.0f}; passes wave0 {passes0[w]} wave1 {passes1[w]}; iters/step {st[w, 3:2*(steps+1):2]}; traj iters {ti[2*w].cpu().numpy()} {ti[2*w+1].cpu().numpy()}")
lib.quattro_debug_set_solve_stamps(ctypes.c_void_p(0))
