import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from quattro_ilqr_amd import ops, _lib
from bench import synthetic_batch
dev = torch.device("cuda:0")
N, steps = 50, 10
md = q.quadrotor_model()
lib = _lib.load()
lib.quattro_debug_set_solve_stamps.argtypes = [ctypes.c_void_p]
x0a, _ = synthetic_batch(8192, 0)
for B, MI in ((3840, 100), (4096, 100), (4096, 20)):
    x0 = torch.as_tensor(x0a[:B], dtype=torch.float32, device=dev)
    sv = q.QuattroILQR(md, N, max_iter=100, tol=1e-3, device=dev)
    sv._alloc(B)
    ws = ops.workspace(md, B, N, dev)
    traj_x = torch.empty((B, steps + 1, 12), dtype=torch.float32, device=dev)
    traj_u = torch.empty((B, steps, 4), dtype=torch.float32, device=dev)
    traj_it = torch.empty((B, steps), dtype=torch.int32, device=dev)
    stamps = torch.zeros(((B + 1) // 2, 2 * (steps + 1)), dtype=torch.int64, device=dev)
    lib.quattro_debug_set_solve_stamps(ctypes.c_void_p(stamps.data_ptr()))
    res = []
    for rep in range(6):
        sv.u.zero_(); x_cur = x0.clone(); stamps.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.mpc_run(md, x_cur, sv.x, sv.u, sv.K, sv.k, sv.cost, sv.tol, MI, steps, ws, traj_x, traj_u, traj_it, alphas=sv.alphas, reg=sv.reg,
                    alpha_idx=sv.alpha_idx, active=sv.active, iters=sv.iters, status=sv.status)
        e1.record()
        torch.cuda.synchronize()
        st = stamps.cpu().numpy()
        res.append((e0.elapsed_time(e1), (st[:, -2].max() - st[:, 0].min()) / 1e5, (st[:, 1].max() - st[:, 0].min()) / 1e5))
    lib.quattro_debug_set_solve_stamps(ctypes.c_void_p(0))
    print(f"B={B:5d} max_iter={MI:3d}: (events ms, last step stamp ms, last wave exit ms) " + "  ".join(f"({a:.2f}, {b:.2f}, {c:.2f})" for a, b, c in res))
