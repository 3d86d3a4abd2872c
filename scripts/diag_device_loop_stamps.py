"""Per-workgroup timeline of the persistent MPC kernel (s_memrealtime stamps after every control step)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from quattro_ilqr_amd import _lib
from bench import synthetic_batch
dev = torch.device("cuda:0")
N, B, steps = 50, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 10
md = q.quadrotor_model()
x0, _ = synthetic_batch(B, 0)
x0 = torch.as_tensor(x0, dtype=torch.float32, device=dev)
mpc = q.BatchedMPC(md, N, max_iter=100, tol=1e-3, device=dev)
mpc.run(x0, steps); mpc.u_warm = None
lib = _lib.load()
W = (B + 1) // 2
# row layout (csrc/solve_quad.hip): [start, wave-0 exit, (end of control step s, iterations in it) x steps, wave-1 exit, wave-1 passes]
ROW = 2 * (steps + 1) + 2
stamps = torch.zeros((W, ROW), dtype=torch.int64, device=dev)
lib.quattro_debug_set_solve_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.quattro_debug_set_solve_stamps(ctypes.c_void_p(stamps.data_ptr()), W)      # the kernel stamps workgroups < W only
torch.cuda.synchronize(); t = time.perf_counter()
out = mpc.run(x0, steps)
torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t)
lib.quattro_debug_set_solve_stamps(ctypes.c_void_p(0), 0)
st = stamps.cpu().numpy()
t0 = st[:, 0].min()
T = (st[:, 0:2 * (steps + 1):2] - t0) / 100.0          # us: start, then the end of every control step
I = st[:, 3:2 * (steps + 1):2]
print(f"wave-1 exit after wave-0 exit: max {(st[:, ROW - 2] - st[:, 1]).max() / 100.0:.1f} us; wave-1 loop passes max {st[:, ROW - 1].max()}")
print(f"wall {ms:.2f} ms; workgroup start spread {T[:,0].max():.1f} us; finish min/median/max {T[:,-1].min():.0f} / {np.median(T[:,-1]):.0f} / {T[:,-1].max():.0f} us")
for cs in range(steps):
    d = T[:, cs + 1] - T[:, cs]
    it = I[:, cs]
    per = d / np.maximum(it, 1)
    print(f"step {cs}: finish median {np.median(T[:,cs+1]):7.0f} us max {T[:,cs+1].max():7.0f}; duration median {np.median(d):6.0f} max {d.max():6.0f}; wg iters mean {it.mean():5.2f} max {it.max():3d}; us/iter median {np.median(per):6.1f}")
slow = np.argsort(T[:, -1])[-5:]
for w in slow:
    print("slow wg", w, "iters", I[w], "step durations us", np.round(T[w, 1:] - T[w, :-1]).astype(int))
