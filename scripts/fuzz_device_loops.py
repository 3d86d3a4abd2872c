"""Randomised comparison of the device-resident loops with the host-driven ones (bit for bit; RK4 quadrotor: iteration counts under
fixed_iters): random batch sizes, horizons, iteration caps, warm / cold starts, closed loops with disturbances.
usage: fuzz_device_loops.py [seconds] [seed] [max_cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else None       # (tests/test_fuzz_gpu.py runs a fixed-size, fixed-seed slice)
rng = np.random.default_rng(seed)
DEV = "cuda:0"
t_end = time.time() + budget
n_cases, n_fail = 0, 0
summary = {}
while time.time() < t_end and (max_cases is None or n_cases < max_cases):
    kind = rng.choice(["quad", "cart", "cart_rk4", "quad_rk4"])
    B = int(rng.choice([1, 2, 3, 5, 17, 64, 129, 300, 511, 1024])) if rng.random() < 0.8 else int(rng.integers(1, 700))
    N = int(rng.choice([1, 2, 3, 7, 12, 13, 24, 25, 26, 30, 49, 50, 51, 64, 75])) if rng.random() < 0.8 else int(rng.integers(1, 90))
    tw = 0
    if kind.startswith("quad"):
        md = q.quadrotor_model(integrator="rk4" if kind.endswith("rk4") else "euler")
        x0 = np.asarray(md.x_ref) + rng.uniform(-1, 1, (B, 12)) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
        u0 = 2.4525 + 0.1 * rng.standard_normal((B, N, 4))
        tol = 1e-3
    else:
        md = q.cartpole_model(dt=0.01, integrator="rk4" if kind.endswith("rk4") else "euler")
        x0 = np.zeros((B, 4)); x0[:, 0] = rng.uniform(-0.5, 0.5, B); x0[:, 2] = rng.uniform(-0.5, 0.5, B)
        u0 = 0.3 * rng.standard_normal((B, N, 1))
        tol = float(rng.choice([1e-1, 1e-3]))
    mi = int(rng.integers(1, 12))
    kw = dict(max_iter=mi, fixed_iters=bool(rng.random() < 0.3))
    u_init = u0 if rng.random() < 0.7 else None
    exact = kind != "quad_rk4"
    dev = q.QuattroILQR(md, N, max_iter=mi, tol=tol, device=DEV, device_loop=True, tf_window=tw)
    host = q.QuattroILQR(md, N, max_iter=mi, tol=tol, device=DEV, device_loop=False, check_every=1, tf_window=tw)
    od = {k: v.clone() for k, v in dev.solve(x0, u_init, **kw).items()}
    oh = host.solve(x0, u_init, **kw)
    bad = []
    if exact:
        bad = [k for k in ("K", "k", "x", "u", "cost", "iters", "alpha", "status") if not torch.equal(od[k], oh[k])]
    elif kw["fixed_iters"] and not torch.equal(od["iters"], oh["iters"]):
        bad = ["iters"]
    if not bool(torch.isfinite(od["cost"]).all()):
        bad.append("nonfinite cost")
    steps = int(rng.integers(1, 4))
    if exact and rng.random() < 0.5:
        dist = torch.as_tensor(1e-3 * rng.standard_normal((steps, B, md.n)), dtype=torch.float32, device=DEV) if rng.random() < 0.5 else None
        a = q.BatchedMPC(md, N, max_iter=mi, tol=tol, device=DEV, check_every=1, tf_window=tw)
        b = q.BatchedMPC(md, N, max_iter=mi, tol=tol, device=DEV, check_every=1, tf_window=tw)
        oa = a.run(x0.astype(np.float32), steps, disturbance=dist, device_loop=True)
        ob = b.run(x0.astype(np.float32), steps, disturbance=dist, device_loop=False)
        bad += ["mpc_" + k for k in ("x", "u", "iters") if not torch.equal(oa[k], ob[k].to(oa[k].dtype))]
    n_cases += 1
    if bad:
        n_fail += 1
        key = (kind, "N=1" if N == 1 else "N>1", tuple(sorted(set("mpc" if b.startswith("mpc") else "solve" for b in bad))))
        summary[key] = summary.get(key, 0) + 1
        if n_fail <= 40:
            print(f"MISMATCH {kind} B={B} N={N} {kw} cold={u_init is None} tol={tol}: {bad}", flush=True)
    if n_cases % 5000 == 0:
        print(f"{n_cases} cases, {n_fail} mismatches", flush=True)
print(f"done: {n_cases} cases, {n_fail} mismatches (seed {seed}); by kind: {summary}")
sys.exit(1 if n_fail else 0)
