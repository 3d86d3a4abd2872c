import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from quattro_ilqr_amd import _lib
from bench import synthetic_batch
dev = torch.device("cuda:0")
N = 50
md = q.quadrotor_model()
lib = _lib.load()
lib.quattro_debug_set_solve_stamps.argtypes = [ctypes.c_void_p]
for B in (2, 4096):
    x0, u0 = synthetic_batch(B, 0)
    x0 = torch.as_tensor(x0, dtype=torch.float32, device=dev); u0 = torch.as_tensor(u0, dtype=torch.float32, device=dev)
    s = q.QuattroILQR(md, N, device=dev)
    s.solve(x0, u0, max_iter=20, fixed_iters=True)
    stamps = torch.zeros(((B + 1) // 2, 4), dtype=torch.int64, device=dev)
    lib.quattro_debug_set_solve_stamps(ctypes.c_void_p(stamps.data_ptr()))
    torch.cuda.synchronize(); t = time.perf_counter()
    s.solve(x0, u0, max_iter=20, fixed_iters=True)
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t)
    lib.quattro_debug_set_solve_stamps(ctypes.c_void_p(0))
    st = stamps.cpu().numpy()
    ticks = st[:, 2].max() - st[:, 0].min()
    print(f"B={B}: wall {ms:.3f} ms, stamp span {ticks} ticks -> {ticks / ms / 1e3:.1f} MHz if the kernel is all of the wall time; iters {st[:,3].max()}")
