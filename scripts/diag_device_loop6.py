import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from quattro_ilqr_amd import ops, _lib
from bench import synthetic_batch
dev = torch.device("cuda:0")
N, B = 50, 4096
md = q.quadrotor_model()
x0, _ = synthetic_batch(B, 0)
x0 = torch.as_tensor(x0, dtype=torch.float32, device=dev)
sv = q.QuattroILQR(md, N, max_iter=100, tol=1e-3, device=dev)
sv._alloc(B)
ws = ops.workspace(md, B, N, dev)
lib = _lib.load()
lib.quattro_debug_set_solve_stamps.argtypes = [ctypes.c_void_p]
for steps in (1, 2, 5, 10):
    traj_x = torch.empty((B, steps + 1, 12), dtype=torch.float32, device=dev)
    traj_u = torch.empty((B, steps, 4), dtype=torch.float32, device=dev)
    traj_it = torch.empty((B, steps), dtype=torch.int32, device=dev)
    stamps = torch.zeros((B // 2, 2 * (steps + 1)), dtype=torch.int64, device=dev)
    for rep in range(3):
        sv.u.zero_(); x_cur = x0.clone()
        lib.quattro_debug_set_solve_stamps(ctypes.c_void_p(stamps.data_ptr() if rep == 2 else 0))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.mpc_run(md, x_cur, sv.x, sv.u, sv.K, sv.k, sv.cost, sv.tol, 100, steps, ws, traj_x, traj_u, traj_it, alphas=sv.alphas, reg=sv.reg,
                    alpha_idx=sv.alpha_idx, active=sv.active, iters=sv.iters, status=sv.status)
        e1.record()
        torch.cuda.synchronize()
        ev = e0.elapsed_time(e1)
    lib.quattro_debug_set_solve_stamps(ctypes.c_void_p(0))
    st = stamps.cpu().numpy()
    T = (st[:, 0::2] - st[:, 0].min()) / 100.0
    fin = np.sort(T[:, -1])
    print(f"steps={steps}: events {ev:.3f} ms; stamps: last workgroup end {fin[-1]/1e3:.3f} ms, 99% {fin[int(0.99*len(fin))]/1e3:.3f}, median {np.median(fin)/1e3:.3f}; zero stamps {(st[:,2::2]==0).sum()}")
