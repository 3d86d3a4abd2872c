"""16-lane cart-pole sweep against the quad-lane one (QUATTRO_HIP_LIB=build_ab/libquattro_cpquad.so): bitwise + timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from quattro_ilqr_amd import ops
dev = torch.device("cuda:0")
out = {}
for integ in ("euler", "rk4"):
    md = q.cartpole_model(dt=0.01, integrator=integ)
    for (B, N, ts) in ((1024, 50, 0), (37, 30, 0), (5, 67, 3), (1, 50, 49), (4096, 50, 0)):
        rng = np.random.default_rng(B + N)
        x0 = np.zeros((B, 4)); x0[:, 0] = rng.uniform(-0.5, 0.5, B); x0[:, 2] = rng.uniform(-0.5, 0.5, B)
        u0 = 0.5 * rng.standard_normal((B, N, 1))
        x0 = torch.as_tensor(x0, dtype=torch.float32, device=dev); u0 = torch.as_tensor(u0, dtype=torch.float32, device=dev)
        xs, _ = ops.simulate(md, x0, u0)
        act = torch.ones(B, dtype=torch.int32, device=dev); act[::3] = 0
        K, k, st = ops.linearize_sweep(md, xs, u0, t_start=ts)
        K2 = torch.full_like(K, 7.0); k2 = torch.full_like(k, 7.0)
        ops.linearize_sweep(md, xs, u0, t_start=ts, K=K2, k=k2, active=act)
        assert torch.equal(K2[1::3], K[1::3]) and bool((K2[::3] == 7.0).all())
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            ops.linearize_sweep(md, xs, u0, t_start=ts, K=K, k=k, status=st)
        torch.cuda.synchronize()
        us = 1e6 * (time.perf_counter() - t) / 20
        out[(integ, B, N, ts)] = (K.cpu().numpy(), k.cpu().numpy(), st.cpu().numpy())
        print(f"{integ} B={B} N={N} t_start={ts}: {us:.1f} us per sweep; |K| max {float(K.abs().max()):.3f} status {int(st.abs().sum())}")
np.savez(sys.argv[1], **{"_".join(map(str, key)) + "_" + nm: arr for key, v in out.items() for nm, arr in zip("Kks", v)})
