"""Find a random RK4 quadrotor case where the fused sweep and the record path differ by more than 2e-5 per step, and measure both
against the fp64 oracle (analytic RK4 linearisation + sweep) on that trajectory."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from quattro_ilqr_amd import ops
from oracle import ilqr as o_ilqr, linearize as o_lin, models as o_models

DEV = torch.device("cuda:0")
t32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=DEV).contiguous()
md = q.quadrotor_model(integrator="rk4")
spec = o_models.quadrotor_spec(integrator=o_models.INTEGRATOR_RK4)
rng = np.random.default_rng(3)
found = 0
for case in range(4000):
    B, N = 64, int(rng.choice([25, 50, 51, 75]))
    x = t32(np.asarray(md.x_ref) + 0.4 * rng.standard_normal((B, N + 1, 12)))
    u = t32(2.4525 + 1.5 * rng.standard_normal((B, N, 4)))
    rec, VxN, VxxN, lay = ops.linearize(md, x, u)
    Kr, kr, sr = ops.riccati_sweep(rec, VxN, VxxN, 12, 4, lay)
    Kf, kf, sf = ops.linearize_sweep(md, x, u)
    num = (Kf.double() - Kr.double()).flatten(2).norm(dim=2); den = Kr.double().flatten(2).norm(dim=2)
    e = num / den
    if float(e.max()) > 2e-5:
        b = int(e.max(dim=1).values.argmax()); t = int(e[b].argmax())
        xs, us = x[b:b + 1].double().cpu().numpy(), u[b:b + 1].double().cpu().numpy()
        d = o_lin.linearize_analytic(spec, xs, us)
        k_o, K_o = o_ilqr.riccati_sweep_batched(d)
        er = np.linalg.norm(Kr[b].double().cpu().numpy() - K_o[0], axis=(1, 2)) / np.linalg.norm(K_o[0], axis=(1, 2))
        ef = np.linalg.norm(Kf[b].double().cpu().numpy() - K_o[0], axis=(1, 2)) / np.linalg.norm(K_o[0], axis=(1, 2))
        print(f"case {case} N={N} b={b} t={t}: fused-vs-records {float(e[b, t]):.2e}; vs fp64 oracle at that step: records {er[t]:.2e}, fused {ef[t]:.2e}; "
              f"worst over t: records {er.max():.2e}, fused {ef.max():.2e}; |K_t| {np.linalg.norm(K_o[0][t]):.3g}, status {int(sr[b])} {int(sf[b])}, "
              f"min u {us.min():.2f}, max|angle| {np.abs(xs[0, :, 6:9]).max():.2f}")
        found += 1
        if found >= 6:
            break
print("searched", case + 1, "batches, found", found)
