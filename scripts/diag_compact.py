"""Diagnostic: expand TILE16C (header + compact) back to full TILE16 records and compare with the direct ones."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
from quattro_ilqr_amd import _lib, ops, quadrotor_model
dev = "cuda:0"
md = quadrotor_model()
rng = np.random.default_rng(23)
B, N = 5, 7
x = torch.as_tensor(np.asarray(md.x_ref) + 0.4 * rng.standard_normal((B, N + 1, 12)), dtype=torch.float32, device=dev)
u = torch.as_tensor(2.4525 + 1.5 * rng.standard_normal((B, N, 4)), dtype=torch.float32, device=dev)
full, VxN, VxxN, _ = ops.linearize(md, x, u, layout=_lib.LAYOUT_TILE16)
comp, _, _, _ = ops.linearize(md, x, u, layout=_lib.LAYOUT_TILE16C)
full = full.reshape(-1, 416).cpu().numpy(); comp = comp.cpu().numpy()
hdr, body = comp[:416], comp[416:].reshape(-1, 76)
DL = [19, 23, 24, 25, 26, 27, 31, 40, 41, 45, 46, 60, 61, 62]
exp = np.tile(hdr, (B * N, 1))
for d, lane in enumerate(DL):
    exp[:, 3 * lane:3 * lane + 3] = body[:, 3 * d:3 * d + 3]
exp[:, 384:400] = body[:, 44:60]
exp[:, 400:416] = body[:, 60:76]
bad = np.argwhere(exp != full)
print("mismatching (record, offset):", len(bad))
offs = sorted(set(bad[:, 1].tolist()))
print("offsets:", offs)
for o in offs[:20]:
    r = bad[bad[:, 1] == o][0][0]
    where = f"F lane {o // 3} s {o % 3} (r={o // 3 // 16}, c={o // 3 % 16})" if o < 192 else f"off {o}"
    print(o, where, "full", full[r, o], "expanded", exp[r, o])
K1 = ops.riccati_sweep(torch.as_tensor(full.reshape(B, N, 416), device=dev), VxN, VxxN, 12, 4, _lib.LAYOUT_TILE16)[0]
K2 = ops.riccati_sweep(torch.as_tensor(comp, device=dev), VxN, VxxN, 12, 4, _lib.LAYOUT_TILE16C)[0]
print("K equal:", torch.equal(K1, K2), "max abs diff", float((K1 - K2).abs().max()))
