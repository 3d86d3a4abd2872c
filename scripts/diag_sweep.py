"""Diagnostic: error metrics of the GPU sweep on the golden inputs (not a test)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd"), os.path.join(ROOT, "tests")]
from conftest import load_golden, per_step_rel, per_step_rel_floor, rel_fro
from quattro_ilqr_amd import _lib, ops
from oracle import ilqr as o_ilqr
BLOCKS = ["A", "B", "lx", "lu", "lxx", "luu", "lux"]
dev32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda:0")
for name, n, m in [("sweep_cartpole_N30.npz", 4, 1), ("sweep_cartpole_N50.npz", 4, 1), ("sweep_quadrotor_N30.npz", 12, 4), ("sweep_quadrotor_N50.npz", 12, 4)]:
    g = load_golden(name)
    for layout in [0] + ([1] if n == 12 else []):
        rec, _ = ops.pack_derivs(*[dev32(g[k]) for k in BLOCKS], layout=layout)
        K, k, st = ops.riccati_sweep(rec, dev32(g["VxN"]), dev32(g["VxxN"]), n, m, layout)
        K, k = K.cpu().numpy(), k.cpu().numpy()
        for b in range(K.shape[0]):
            d = {kk: g[kk][b] for kk in BLOCKS + ["VxN", "VxxN"]}
            k32, K32 = o_ilqr.riccati_sweep(d, dtype=np.float32)
            print(f"{name} layout {layout} b{b}: K fro {rel_fro(K[b], g['K'][b]):.2e} step {per_step_rel(K[b], g['K'][b]):.2e} | "
                  f"k fro {rel_fro(k[b], g['k'][b]):.2e} step {per_step_rel(k[b], g['k'][b]):.2e} | "
                  f"k floor1% {per_step_rel_floor(k[b], g['k'][b], 0.01):.2e} 5% {per_step_rel_floor(k[b], g['k'][b], 0.05):.2e} 10% {per_step_rel_floor(k[b], g['k'][b], 0.1):.2e} np32 5% {per_step_rel_floor(k32, g['k'][b], 0.05):.2e} | "
                  f"numpy32: K step {per_step_rel(K32, g['K'][b]):.2e} k step {per_step_rel(k32, g['k'][b]):.2e} status {int(st[b])}")
