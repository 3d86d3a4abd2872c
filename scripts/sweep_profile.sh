#!/bin/bash
# Diagnostic: per-phase s_memtime shares of one sweep step (-DQT_SWEEP_PROFILE build of sweep_tile16.hip).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DQT_SWEEP_PROFILE -shared quattro-transformer-ilqr_amd/csrc/sweep_tile16.hip -o gpurun_out/libsweepprof.so || exit 1
python3 - <<'PY'
import ctypes, sys, numpy as np, torch
sys.path[:0] = [".", "quattro-transformer-ilqr_amd"]
from quattro_ilqr_amd import _lib, ops, quadrotor_model
import bench
lib = ctypes.CDLL("gpurun_out/libsweepprof.so")
dev = "cuda:0"; md = quadrotor_model(); N = 50
P = ctypes.c_void_p
for B in (256, 1024, 4096, 16384):
    x0h, u0h = bench.synthetic_batch(B, 0)
    x0 = torch.as_tensor(x0h, dtype=torch.float32, device=dev); u = torch.as_tensor(u0h, dtype=torch.float32, device=dev)
    x, _ = ops.simulate(md, x0, u)
    for layout, compact in ((_lib.LAYOUT_TILE16C, 1), (_lib.LAYOUT_TILE16, 0)):
        rec, VxN, VxxN, _ = ops.linearize(md, x, u, layout=layout)
        K = torch.empty((B, N, 4, 12), device=dev); k = torch.empty((B, N, 4), device=dev)
        dbg = torch.zeros(B * 8, dtype=torch.int64, device=dev)
        for _ in range(3):
            rc = lib.quattro_sweep_profile(P(rec.data_ptr()), P(VxN.data_ptr()), P(VxxN.data_ptr()), B, N, ctypes.c_float(1e-6),
                                           P(K.data_ptr()), P(k.data_ptr()), compact, P(dbg.data_ptr()), None)
        torch.cuda.synchronize(); assert rc == 0
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.quattro_sweep_profile(P(rec.data_ptr()), P(VxN.data_ptr()), P(VxxN.data_ptr()), B, N, ctypes.c_float(1e-6),
                                  P(K.data_ptr()), P(k.data_ptr()), compact, P(dbg.data_ptr()), None)
        e1.record(); torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(B, 8).astype(np.float64)
        med = np.median(d, axis=0)
        names = ["P,Q MFMAs (+record wait)", "q_z row sum", "Gauss-Jordan (4 pivots)", "V' MFMA + V_x' row sum", "LDS transpose / symmetrise"]
        print(f"B={B} compact={compact}: kernel {e0.elapsed_time(e1)*1e3:.1f} us; wave lifetime median {med[5]:.0f} ticks; per step:")
        for n_, v in zip(names, med[:5]):
            print(f"     {n_:30s} {v/N:7.1f} ticks/step  {100*v/med[5]:5.1f}%")
PY
