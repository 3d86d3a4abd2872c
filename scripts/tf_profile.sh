#!/bin/bash
# Diagnostic: per-phase cycle shares of the fused transformer kernel (s_memtime stamps, -DQT_TF_PROFILE build).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DQT_TF_PROFILE -shared quattro-transformer-ilqr_amd/csrc/tf_stream.hip -o gpurun_out/libtfprof.so || exit 1
timeout -k 10 200 python3 - <<'PY'
import ctypes, sys, numpy as np, torch
sys.path[:0] = [".", "quattro-transformer-ilqr_amd"]
from quattro_ilqr_amd import TransformerILQR, _lib
lib = ctypes.CDLL("gpurun_out/libtfprof.so")
dev = "cuda:0"
tf = TransformerILQR.random_init(12, 52, prompt_len=1, target_len=49, device=dev)
names = ["prologue", "param step", "QKV steps", "xch barrier", "attention", "O step", "LN1", "FFN steps", "LN2", "head+stores"]
for B in (256, 512, 4096):
    x = torch.randn(B, 51, 12, device=dev); p = torch.randn(B, 1, 52, device=dev)
    s = tf._struct(51)
    pred = torch.empty(B, 49, 52, device=dev)
    dbg = torch.zeros(B * 4 * 16, dtype=torch.int64, device=dev)
    for _ in range(3):
        rc = lib.quattro_tf_stream_profile(ctypes.byref(s), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(p.data_ptr()), B,
                                           ctypes.c_void_p(pred.data_ptr()), ctypes.c_void_p(dbg.data_ptr()), None)
    torch.cuda.synchronize()
    assert rc == 0
    d = dbg.cpu().numpy().reshape(B, 4, 16).astype(np.float64)
    tot = np.median(d[:, :, 11])
    print(f"B={B}: median cycles per workgroup {tot:.0f}  (sum of phases {np.median(d[:, :, :10].sum(axis=2)):.0f})")
    for w in range(4):
        row = "  ".join(f"{np.median(d[:, w, i]):8.0f}" for i in range(10))
        print(f"   wave {w}: {row}   rendezvous {np.median(d[:, w, 10]):8.0f}")
    print("   " + "  ".join(f"{n[:8]:>8s}" for n in names))
    print("   share  : " + "  ".join(f"{100*np.median(d[:, :, i])/tot:7.1f}%" for i in range(10)) + f"   rendezvous {100*np.median(d[:,:,10])/tot:5.1f}%")
PY
