#!/bin/bash
# Diagnostic: per-phase cycle shares of the fused transformer kernel (s_memtime stamps, -DQT_TF_PROFILE build).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DQT_TF_PROFILE -shared quattro-transformer-ilqr_amd/csrc/tf_forward.hip -o gpurun_out/libtfprof.so || exit 1
python3 - <<'PY'
import ctypes, sys, numpy as np, torch
sys.path[:0] = [".", "quattro-transformer-ilqr_amd"]
from quattro_ilqr_amd import TransformerILQR, _lib
lib = ctypes.CDLL("gpurun_out/libtfprof.so")
dev = "cuda:0"
tf = TransformerILQR.random_init(12, 52, prompt_len=1, target_len=49, device=dev)
for B in (256, 4096):
    x = torch.randn(B, 51, 12, device=dev); p = torch.randn(B, 1, 52, device=dev)
    s = tf._struct(51)
    pred = torch.empty(B, 49, 52, device=dev)
    dbg = torch.zeros(B * 64, dtype=torch.int64, device=dev)
    for _ in range(3):
        rc = lib.quattro_tf_forward_profile(ctypes.byref(s), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(p.data_ptr()), B,
                                            ctypes.c_void_p(pred.data_ptr()), ctypes.c_void_p(dbg.data_ptr()), None)
    torch.cuda.synchronize()
    assert rc == 0
    d = dbg.cpu().numpy().reshape(B, 64).astype(np.int64)
    names = ["embed"] + [f"L{l}:{n}" for l in range(3) for n in ("qkv", "attn", "outproj", "ln1", "ffn", "ln2")] + ["head"]
    idx = list(range(0, 20)) + [62]
    t = d[:, idx]
    dt = np.diff(t, axis=1)
    med = np.median(dt, axis=0)
    tot = np.median(t[:, -1] - t[:, 0])
    print(f"B={B}: total median {tot:.0f} ticks (100 MHz s_memtime => {tot/100:.1f} us per workgroup)")
    for n, v in zip(names, med):
        print(f"   {n:12s} {v:9.0f}  {100*v/tot:5.1f}%")
PY
