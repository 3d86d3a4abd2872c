"""Randomised consistency checks of the rollout kernels: the fused line search against the candidate rollouts it must agree with
(costs, accepted index = first candidate whose cost does not exceed the nominal's, committed trajectory = that candidate, bit for bit),
for 1..8 step sizes, random batch sizes / horizons / active masks, both models and integrators; simulate vs total_cost; records
pack -> unpack round trips.  usage: fuzz_rollouts.py [seconds] [seed] [max_cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q
from quattro_ilqr_amd import _lib, ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else None       # (tests/test_fuzz_gpu.py runs a fixed-size, fixed-seed slice)
rng = np.random.default_rng(seed)
DEV = torch.device("cuda:0")
t32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=DEV).contiguous()
t_end = time.time() + budget
n_cases, fails = 0, {}


def fail(tag, info):
    fails[tag] = fails.get(tag, 0) + 1
    if sum(fails.values()) <= 25:
        print("MISMATCH", tag, info, flush=True)


while time.time() < t_end and (max_cases is None or n_cases < max_cases):
    kind = rng.choice(["quad", "quad_rk4", "cart", "cart_rk4"])
    B = int(rng.choice([1, 2, 3, 5, 17, 64, 129, 300])) if rng.random() < 0.8 else int(rng.integers(1, 500))
    N = int(rng.choice([1, 2, 3, 7, 12, 25, 30, 50, 51, 75])) if rng.random() < 0.8 else int(rng.integers(1, 100))
    if kind.startswith("quad"):
        md = q.quadrotor_model(integrator="rk4" if kind.endswith("rk4") else "euler")
        x0 = t32(np.asarray(md.x_ref) + rng.uniform(-1, 1, (B, 12)) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0]))
        u = t32(2.4525 + 0.3 * rng.standard_normal((B, N, 4)))
        Ksc, ksc = 0.5, 0.2
    else:
        md = q.cartpole_model(dt=0.01, integrator="rk4" if kind.endswith("rk4") else "euler")
        x0 = t32(np.stack([rng.uniform(-0.5, 0.5, B), np.zeros(B), rng.uniform(-0.5, 0.5, B), np.zeros(B)], axis=1))
        u = t32(0.3 * rng.standard_normal((B, N, 1)))
        Ksc, ksc = 2.0, 0.5
    n, m = md.n, md.m
    info = f"{kind} B={B} N={N}"
    x, cost = ops.simulate(md, x0, u)
    tc = ops.total_cost(md, x, u)
    if not bool(((cost - tc).abs() <= 1e-6 * tc.abs().clamp_min(1e-30)).all()):
        fail("simulate cost vs total_cost", info)
    K = t32(Ksc * rng.standard_normal((B, N, m, n)) * (rng.random() < 0.8)); k = t32(ksc * rng.standard_normal((B, N, m)))
    na = int(rng.integers(1, 9))
    alphas = tuple(sorted(rng.uniform(0.005, 1.0, na).tolist(), reverse=True))
    active = torch.as_tensor((rng.random(B) < 0.7).astype(np.int32), device=DEV) if rng.random() < 0.5 else None
    costs, xn, un = ops.rollout(md, x, u, K, k, alphas=alphas, want_traj=True, active=active)
    xs, us, cs = x.clone(), u.clone(), cost.clone()
    aidx = torch.full((B,), -5, dtype=torch.int32, device=DEV)
    act2 = None if active is None else active.clone()
    iters = torch.zeros(B, dtype=torch.int32, device=DEV)
    ops.linesearch(md, xs, us, K, k, cs, 1e-3, alphas=alphas, alpha_idx=aidx, active=act2, iters=iters)
    live = torch.ones(B, dtype=torch.bool, device=DEV) if active is None else active.bool()
    ok = (costs <= cost[None, :])                                          # (na, B); NaN compares false like the reference
    first = torch.where(ok.any(dim=0), ok.int().argmax(dim=0), torch.full((B,), -1, device=DEV, dtype=torch.int64))
    if bool(live.any()):
        if not torch.equal(aidx[live].long(), first[live]):
            fail("accepted index", info + f" n_alpha={na}")
        else:
            acc = live & (first >= 0)
            if bool(acc.any()):
                sel = first[acc]
                bi = acc.nonzero().flatten()
                if not (torch.equal(xs[bi], xn[sel, bi]) and torch.equal(us[bi], un[sel, bi]) and torch.equal(cs[bi], costs[sel, bi])):
                    fail("committed candidate", info + f" n_alpha={na}")
            rej = live & (first < 0)
            if bool(rej.any()) and not (torch.equal(xs[rej], x[rej]) and torch.equal(us[rej], u[rej]) and torch.equal(cs[rej], cost[rej])):
                fail("rejected trajectory modified", info)
            if not bool((iters[live] == 1).all()):
                fail("iters", info)
    if bool((~live).any()) and not (torch.equal(xs[~live], x[~live]) and torch.equal(us[~live], u[~live]) and bool((aidx[~live] == -5).all())
                                    and bool((iters[~live] == 0).all())):
        fail("inactive trajectory touched", info)
    # records: pack -> unpack round trip in every layout that can be packed
    S = int(rng.integers(1, 6)); Bp = int(rng.integers(1, 9))
    blocks = [t32(rng.standard_normal((Bp, S) + s)) for s in ((n, n), (n, m), (n,), (m,), (n, n), (m, m), (m, n))]
    for lay in ([_lib.LAYOUT_ROWMAJOR, _lib.LAYOUT_TILE16] if n == 12 else [_lib.LAYOUT_ROWMAJOR]):
        rec, _ = ops.pack_derivs(*blocks, layout=lay)
        back = ops.unpack_derivs(rec, Bp, n, m, lay)
        for name, src in zip(("A", "B", "lx", "lu", "lxx", "luu", "lux"), blocks):
            if not torch.equal(back[name], src):
                fail("pack/unpack", f"{kind} layout {lay} {name}")
    n_cases += 1
    if n_cases % 500 == 0:
        print(f"{n_cases} cases, {sum(fails.values())} mismatches", flush=True)
print(f"done: {n_cases} cases, mismatches: {fails} (seed {seed})")
sys.exit(1 if fails else 0)
