"""Diagnostic: which host call stalls when solve() is fed numpy inputs back to back?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
from quattro_ilqr_amd import QuattroILQR, ops, quadrotor_model
import quattro_ilqr_amd.solver as S
dev = "cuda:0"; B, N = 4096, 50
md = quadrotor_model()
x0n = np.asarray(md.x_ref) + 0.02 * np.random.default_rng(0).standard_normal((B, 12))
sv = QuattroILQR(md, N, device=dev)
sv.solve(x0n, max_iter=3, fixed_iters=True)
stamps = []
orig_upload = sv._upload
def timed_upload(dst, src, name):
    t0 = time.perf_counter()
    pin = sv._pin.get(name)
    if sv._pin_done is not None:
        sv._pin_done.synchronize()
    t1 = time.perf_counter()
    pin.copy_(torch.as_tensor(np.asarray(src)).reshape(dst.shape))
    t2 = time.perf_counter()
    dst.copy_(pin, non_blocking=True)
    t3 = time.perf_counter()
    sv._pin_done = torch.cuda.Event(); sv._pin_done.record()
    t4 = time.perf_counter()
    stamps.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
sv._upload = timed_upload
orig_sim = ops.simulate
for iters in (20, 40):
    stamps.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    marks = []
    for _ in range(3):
        ta = time.perf_counter(); sv.solve(x0n, max_iter=iters, fixed_iters=True); marks.append(time.perf_counter() - ta)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print(f"iters={iters}: total {tot*1e3:.1f} ms; per-solve host {['%.1f' % (m*1e3) for m in marks]} ms")
    for s in stamps: print("   upload: evsync %.2f ms, host copy %.2f ms, H2D issue %.2f ms, record %.2f ms" % tuple(1e3 * v for v in s))
