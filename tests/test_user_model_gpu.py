"""User-compiled device models (quattro_ilqr_amd.user_model; csrc/user_model.h, csrc/dual.h) against the oracle.

The reference's iLQR_TF accepts any callables f, L, Lf (quattro_ilqr_tf.py:66-84).  Here a problem the reference does not
ship — a planar two-rotor vehicle with a non-diagonal, non-quadratic cost — is written once as C++ bodies, compiled
for the GPU, and compared with the fp64 oracle running the reference's algorithm (finite differences and all) on the
same problem written as Python callables; and the reference's own quadrotor, re-entered as a *user* model, must reproduce
the built-in kernels (which are pinned to the golden vectors elsewhere).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]

from oracle import ilqr as O  # noqa: E402

pytestmark = pytest.mark.gpu

# ---------------------------------------------------------------------------------------------- the planar vehicle
PHYS = (1.0, 0.05, 0.2, 9.81)                      # mass, inertia, arm, gravity
Q = np.array([1.0, 1.0, 1.0, 0.1, 0.1, 0.1]); R = np.array([0.01, 0.02]); QF = np.full(6, 10.0)
XREF = np.array([0.0, 1.0, 0.0, 0.0, 0.0, 0.0])
DT = 0.02

PLANAR_RATE = """
T s, c;
sincos(x[2], &s, &c);
const T thrust = (u[0] + u[1]) / P[0];
xd[0] = x[3];  xd[1] = x[4];  xd[2] = x[5];
xd[3] = -(thrust * s);
xd[4] = thrust * c - P[3];
xd[5] = (u[0] - u[1]) * (P[2] / P[1]);
"""
PLANAR_L = "return default_stage_cost<T>(p, x, u) + x[0] * x[2] * 0.3f + exp(u[0] * 0.1f) * 0.01f;"
PLANAR_LF = "return default_final_cost<T>(p, x) + x[0] * x[1] * 0.5f;"


def planar_rate(x, u):
    m, inertia, arm, g = PHYS
    s, c = np.sin(x[2]), np.cos(x[2])
    th = (u[0] + u[1]) / m
    return np.array([x[3], x[4], x[5], -th * s, th * c - g, (u[0] - u[1]) * arm / inertia])


def planar_f(integrator):
    def f(x, u):
        if integrator == "euler":
            return x + DT * planar_rate(x, u)
        k1 = planar_rate(x, u)
        k2 = planar_rate(x + 0.5 * DT * k1, u)
        k3 = planar_rate(x + 0.5 * DT * k2, u)
        k4 = planar_rate(x + DT * k3, u)
        return x + DT / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
    return f


def planar_L(x, u):
    d = x - XREF
    return np.sum(Q * d * d) + np.sum(R * u * u) + 0.3 * x[0] * x[2] + 0.01 * np.exp(0.1 * u[0])


def planar_Lf(x):
    d = x - XREF
    return np.sum(QF * d * d) + 0.5 * x[0] * x[1]


def planar_model(integrator):
    import quattro_ilqr_amd as q
    return q.compile_model("planar", 6, 2, rate=PLANAR_RATE, stage_cost=PLANAR_L, final_cost=PLANAR_LF, dt=DT,
                           integrator=integrator, phys=PHYS, q=Q, r=R, qf=QF, x_ref=XREF)


def planar_batch(B, N, seed=0):
    rng = np.random.default_rng(seed)
    x0 = XREF + rng.normal(0, 0.3, (B, 6)) * np.array([1, 1, 0.3, 0.5, 0.5, 0.5])
    u0 = np.full((B, N, 2), PHYS[0] * PHYS[3] / 2) + rng.normal(0, 0.2, (B, N, 2))
    return x0, u0


def complex_step_jac(fun, z, h=1e-30):
    """Exact (to fp64 round-off) Jacobian of an analytic function by complex-step differentiation."""
    z = np.asarray(z, dtype=np.complex128)
    cols = []
    for i in range(z.size):
        zp = z.copy(); zp[i] += 1j * h
        cols.append(np.imag(np.atleast_1d(fun(zp))) / h)
    return np.array(cols).T


def exact_derivs(f, L, x, u):
    n, m = x.size, u.size
    z = np.concatenate([x, u])
    F = complex_step_jac(lambda zz: f(zz[:n], zz[n:]), z)
    grad = lambda zz: complex_step_jac(lambda w: L(w[:n], w[n:]), zz)[0]
    g = grad(z)
    H = np.zeros((n + m, n + m))
    eps = 1e-6
    for i in range(n + m):
        e = np.zeros(n + m); e[i] = eps
        H[:, i] = (grad(z + e) - grad(z - e)) / (2 * eps)
    return F[:, :n], F[:, n:], g[:n], g[n:], H[:n, :n], H[n:, n:], H[n:, :n]


# ---------------------------------------------------------------------------------------------- tests
@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_user_model_derivative_records_are_exact(integrator):
    """Records from forward-mode differentiation on the device vs exact fp64 derivatives of the same problem in Python
    (complex step), and vs the reference's finite differences (oracle) at their own noise level."""
    import torch
    from quattro_ilqr_amd import _lib, ops
    md = planar_model(integrator)
    f = planar_f(integrator)
    B, N = 5, 12
    x0, u0 = planar_batch(B, N, 1)
    dev = torch.device("cuda:0")
    x, cost = ops.simulate(md, torch.as_tensor(x0, dtype=torch.float32, device=dev),
                           torch.as_tensor(u0, dtype=torch.float32, device=dev))
    xs = x.double().cpu().numpy()
    for b in range(B):                                   # rollout + cost of the float instantiation
        xo = O.rollout(f, x0[b], u0[b])
        assert np.max(np.abs(xs[b] - xo)) < 2e-5
        assert abs(cost[b].item() - O.trajectory_cost(planar_L, planar_Lf, xo, u0[b])) < 1e-5 * abs(cost[b].item())
    rec, VxN, VxxN, layout = ops.linearize(md, x, torch.as_tensor(u0, dtype=torch.float32, device=dev))
    assert layout == _lib.LAYOUT_ROWMAJOR_TILE          # ROWMAJOR records, swept by the MFMA tile kernel (n <= 12, m <= 4)
    blocks = {k: v.double().cpu().numpy() for k, v in ops.unpack_derivs(rec, B, 6, 2, layout, lib=_lib.load_for(md)).items()}
    worst = {}
    for b in range(B):
        for t in range(N):
            A, Bm, lx, lu, lxx, luu, lux = exact_derivs(f, planar_L, xs[b, t], u0[b, t].astype(np.float32).astype(np.float64))
            for name, ref in (("A", A), ("B", Bm), ("lx", lx), ("lu", lu), ("lxx", lxx), ("luu", luu), ("lux", lux)):
                err = np.max(np.abs(blocks[name][b, t] - ref)) / max(1.0, np.max(np.abs(ref)))
                worst[name] = max(worst.get(name, 0.0), err)
    print("user-model records vs exact derivatives (max abs / max(1, |ref|)):", {k: f"{v:.2e}" for k, v in worst.items()})
    assert max(worst.values()) < 1e-6                    # measured 4e-8
    # terminal pair
    gN = complex_step_jac(lambda w: planar_Lf(w), xs[0, N])[0]
    assert np.max(np.abs(VxN[0].double().cpu().numpy() - gN)) < 1e-5 * max(1.0, np.max(np.abs(gN)))
    HN = VxxN[0].double().cpu().numpy()
    assert np.max(np.abs(HN - HN.T)) < 1e-6 and np.max(np.abs(np.diag(HN) - 2 * QF)) < 1e-5 and abs(HN[0, 1] - 0.5) < 1e-6
    # the reference's own finite differences of the same callables (what its backward pass would consume)
    d = O.linearize_fd(f, planar_L, planar_Lf, xs[0], [u for u in u0[0].astype(np.float32).astype(np.float64)])
    assert np.max(np.abs(blocks["A"][0] - d["A"])) < 1e-5 and np.max(np.abs(blocks["B"][0] - d["B"])) < 1e-5
    assert np.max(np.abs(blocks["lxx"][0] - d["lxx"])) < 5e-4 and np.max(np.abs(blocks["lux"][0] - d["lux"])) < 5e-4


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_user_model_solve_matches_the_oracle(integrator):
    """Whole solves of the planar problem: the reference's algorithm in fp64 (oracle.optimize on Python callables) vs
    QuattroILQR on the compiled model — same accepted step sizes, same iteration count, same optimum."""
    import torch
    import quattro_ilqr_amd as q
    md = planar_model(integrator)
    f = planar_f(integrator)
    B, N = 6, 30
    x0, u0 = planar_batch(B, N, 2)
    s = q.QuattroILQR(md, N, max_iter=40, tol=1e-3, device="cuda:0")
    out = s.solve(torch.as_tensor(x0, dtype=torch.float32), torch.as_tensor(u0, dtype=torch.float32))
    u_dev, x_dev = out["u"].double().cpu().numpy(), out["x"].double().cpu().numpy()
    it_dev, cost_dev = out["iters"].cpu().numpy(), out["cost"].cpu().numpy()
    same_iters = 0
    for b in range(B):
        u_o, x_o, logs = O.optimize(f, planar_L, planar_Lf, x0[b], list(u0[b].astype(np.float32).astype(np.float64)), N,
                                    max_iter=40, tol=1e-3)
        J_o = O.trajectory_cost(planar_L, planar_Lf, x_o, u_o)
        same_iters += int(len(logs) == it_dev[b])
        print(f"planar/{integrator} b={b}: iterations oracle {len(logs)} device {it_dev[b]}, cost {J_o:.6f} vs {cost_dev[b]:.6f}, "
              f"max|dx| {np.max(np.abs(x_dev[b] - x_o)):.2e} max|du| {np.max(np.abs(u_dev[b] - np.array(u_o))):.2e}")
        assert abs(len(logs) - it_dev[b]) <= 1            # the stop test |dJ| < tol sits on fp32-vs-fp64 round-off at the end
        if len(logs) == it_dev[b]:                        # measured: 2e-7 relative cost, 5e-7 in x, 3e-6 in u
            assert abs(J_o - cost_dev[b]) < 1e-6 * abs(J_o)
            assert np.max(np.abs(x_dev[b] - x_o)) < 1e-5 and np.max(np.abs(u_dev[b] - np.array(u_o))) < 3e-5
    assert same_iters >= B - 1
    # first iteration alone, from identical nominals: gains against the reference's FD backward pass
    s2 = q.QuattroILQR(md, N, device="cuda:0")
    s2.solve(torch.as_tensor(x0, dtype=torch.float32), torch.as_tensor(u0, dtype=torch.float32), max_iter=0)
    xs = s2.x.double().cpu().numpy().copy()
    s2.iterate()
    K_dev, k_dev = s2.K.double().cpu().numpy(), s2.k.double().cpu().numpy()
    for b in range(2):
        k_o, K_o = O.backward_pass(f, planar_L, planar_Lf, xs[b], list(u0[b].astype(np.float32).astype(np.float64)))
        K_o, k_o = np.array(K_o), np.array(k_o)
        eK = np.linalg.norm(K_dev[b] - K_o) / np.linalg.norm(K_o)
        ek = np.linalg.norm(k_dev[b] - k_o) / np.linalg.norm(k_o)
        print(f"planar/{integrator} first sweep b={b}: rel-Fro K {eK:.2e} k {ek:.2e}")
        assert eK < 1e-4 and ek < 1e-4                    # measured 1.2e-5 / 1.5e-5: the reference's FD noise on this problem


QUAD_RATE = """
const float mass = P[0], Ix = P[1], Iy = P[2], Iz = P[3], arm = P[4], grav = P[5], kyaw = P[6];
T sph, cph, sth, cth, sps, cps;
sincos(x[6], &sph, &cph);  sincos(x[7], &sth, &cth);  sincos(x[8], &sps, &cps);
const T sec = 1.0f / cth, tth = sth * sec;
const T wp = x[9], wq = x[10], wr = x[11];
const T tm = (u[0] + u[1] + u[2] + u[3]) / mass;
xd[0] = x[3];  xd[1] = x[4];  xd[2] = x[5];
xd[3] = tm * (sps * sph + cps * sth * cph);
xd[4] = tm * (cps * sph - sps * sth * cph);
xd[5] = tm * (cth * cph) - grav;
const T mix = wq * sph + wr * cph;
xd[6] = wp + mix * tth;
xd[7] = wq * cph - wr * sph;
xd[8] = mix * sec;
xd[9] = (wq * wr) * ((Iy - Iz) / Ix) + ((u[1] + u[2]) - (u[0] + u[3])) * (arm / Ix);
xd[10] = (wp * wr) * ((Iz - Ix) / Iy) + ((u[0] + u[1]) - (u[2] + u[3])) * (arm / Iy);
xd[11] = (wp * wq) * ((Ix - Iy) / Iz) + (u[0] - u[1] + u[2] - u[3]) * (kyaw / Iz);
"""


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_quadrotor_reentered_as_user_model_reproduces_the_builtin_kernels(integrator):
    """quadrotor_dynamics.py:63-164 + the barrier cost typed in as a user model (n = 12, m = 4, default cost): differentiated
    automatically and solved by the generic kernels, it must agree with the hand-derived built-in path."""
    import torch
    import quattro_ilqr_amd as q
    from bench import synthetic_batch
    builtin = q.quadrotor_model(integrator=integrator)
    user = q.compile_model("quadrotor_user", 12, 4, rate=QUAD_RATE, dt=builtin.dt, integrator=integrator, phys=builtin.phys,
                           q=builtin.q, r=builtin.r, qf=builtin.qf, x_ref=builtin.x_ref,
                           barrier_alpha=builtin.barrier_alpha, barrier_beta=builtin.barrier_beta)
    B, N = 64, 50
    x0, u0 = synthetic_batch(B, 0)
    x0 = torch.as_tensor(x0, dtype=torch.float32, device="cuda:0"); u0 = torch.as_tensor(u0, dtype=torch.float32, device="cuda:0")
    sb = q.QuattroILQR(builtin, N, device="cuda:0"); su = q.QuattroILQR(user, N, device="cuda:0")
    sb.solve(x0, u0, max_iter=1); su.solve(x0, u0, max_iter=1)
    eK = (torch.linalg.norm((sb.K - su.K).double().reshape(B, -1), dim=1) / torch.linalg.norm(sb.K.double().reshape(B, -1), dim=1)).max().item()
    ek = (torch.linalg.norm((sb.k - su.k).double().reshape(B, -1), dim=1) / torch.linalg.norm(sb.k.double().reshape(B, -1), dim=1)).max().item()
    print(f"user-model quadrotor vs built-in ({integrator}), first sweep: rel-Fro K {eK:.2e} k {ek:.2e}")
    assert eK < 2e-5 and ek < 2e-5
    assert torch.equal(sb.alpha_idx, su.alpha_idx)
    ob = {k: v.clone() for k, v in sb.solve(x0, u0).items()}
    ou = su.solve(x0, u0)
    same = (ob["iters"] == ou["iters"]).float().mean().item()
    rel = ((ob["cost"] - ou["cost"]).abs() / ob["cost"].abs()).max().item()
    print(f"converged solves: same iteration count on {100 * same:.0f} %, max rel cost difference {rel:.2e}, "
          f"max|du| {(ob['u'] - ou['u']).abs().max().item():.2e}")
    assert same >= 0.9 and rel < 1e-4


def test_user_model_through_the_dropin_class_and_the_mpc_loop():
    """iLQR_TF(model=<user model>) — the reference's class surface — and BatchedMPC run on the same compiled problem."""
    import torch
    import quattro_ilqr_amd as q
    md = planar_model("rk4")
    N = 25
    x0, u0 = planar_batch(3, N, 3)
    il = q.iLQR_TF(md, md, md, x0[0], [u for u in u0[0]], N, dt=DT, max_iter=30, tol=1e-3, device="cuda:0")
    u_opt, x_opt = il.optimize(XREF)
    s = q.QuattroILQR(md, N, max_iter=30, tol=1e-3, device="cuda:0")
    out = s.solve(torch.as_tensor(x0[:1], dtype=torch.float32), torch.as_tensor(u0[:1], dtype=torch.float32))
    assert np.max(np.abs(np.array(u_opt) - out["u"][0].double().cpu().numpy())) < 1e-6
    assert len(il.logs) == int(out["iters"][0])
    k_seq, K_seq = il.backward_pass(il.simulate([u for u in u0[0]]), [u for u in u0[0]])
    assert np.array(K_seq).shape == (N, 2, 6) and np.array(k_seq).shape == (N, 2)
    mpc = q.BatchedMPC(md, N, max_iter=30, device="cuda:0")
    run = mpc.run(torch.as_tensor(x0, dtype=torch.float32), 8)
    xs = run["x"].double().cpu().numpy()
    d0 = np.linalg.norm(xs[:, 0, :3] - XREF[:3], axis=1); d1 = np.linalg.norm(xs[:, -1, :3] - XREF[:3], axis=1)
    assert run["x"].shape == (3, 9, 6) and np.all(np.isfinite(xs)) and np.all(d1 < d0)


def test_user_model_is_refused_by_the_stock_library_and_by_bad_dims():
    import ctypes
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import _lib
    md = planar_model("euler")
    p = md.c_params()
    assert _lib.load().quattro_model_layout(ctypes.byref(p)) == -1               # libquattro_hip.so has no model 3
    assert _lib.load_for(md).quattro_model_layout(ctypes.byref(p)) == _lib.LAYOUT_ROWMAJOR_TILE
    with pytest.raises(ValueError):
        q.compile_model("too_big", 17, 2, rate="xd[0] = x[0];")
    with pytest.raises(_lib.QuattroError):
        q.compile_model("broken", 2, 1, rate="xd[0] = undefined_symbol;")


def test_user_model_whole_workflow_collect_fit_hybrid():
    """The reference's actual use case on a problem it does not ship: iLQR logs of the user model -> training set -> fit a
    gain predictor (state_dim 6, control_dim m (1 + n) = 14: the layer-wise kernels, no fused shape) -> hybrid solves where
    the predictor supplies the first N - P steps' gains (training_data_collection.py:196-214, transformer_ilqr.py:102-208,
    quattro_ilqr_tf.py:476-591)."""
    import torch
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import datagen
    md = planar_model("rk4")
    N, P, B = 30, 5, 128
    x0, u0 = planar_batch(B, N, 5)
    log = datagen.collect(q.QuattroILQR(md, N, max_iter=6, tol=1e-3, device="cuda:0"), x0, u0)
    assert len(log) > B and log.K_seq.shape[1:] == (N, 2, 6) and log.x_seq.shape[1:] == (N + 1, 6)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(log))
    n_train = int(0.8 * len(log))
    tf = q.TransformerILQR(6, 14, prompt_len=P, d_model=64, nhead=4, num_decoder_layers=2, dim_feedforward=128, dropout=0.0,
                           max_seq_len=80, device="cuda:0")
    tf.fit(log.select(perm[:n_train]), log.select(perm[n_train:]), num_epochs=20, batch_size=32, learning_rate=1e-3, patience=20)
    assert tf.train_loss_history[-1] < 0.5 * tf.train_loss_history[0]
    hyb = q.QuattroILQR(md, N, max_iter=4, tol=1e-3, tf=tf, device="cuda:0")
    out = hyb.solve(x0[:16], u0[:16])
    _, cost0 = q.ops.simulate(md, torch.as_tensor(x0[:16], dtype=torch.float32, device="cuda:0"),
                              torch.as_tensor(u0[:16], dtype=torch.float32, device="cuda:0"))
    assert bool(torch.isfinite(out["cost"]).all()) and int((out["iters"] >= 1).sum()) == 16
    assert bool((out["cost"] <= cost0).all())               # accepted steps never increase the cost
    # the drop-in class takes the same decisions with the same predictor
    il = q.iLQR_TF(md, md, md, x0[0], [u for u in u0[0]], N, dt=DT, max_iter=4, tol=1e-3, tf=tf, device="cuda:0")
    u_fin, x_fin = il.optimize(XREF)
    assert len(il.logs) == int(out["iters"][0])
    assert np.max(np.abs(x_fin - out["x"][0].double().cpu().numpy())) < 1e-4


@pytest.mark.parametrize("integrator,B,N", [("rk4", 1, 30), ("euler", 37, 25), ("rk4", 300, 50), ("euler", 5, 7)])
def test_user_model_device_resident_solve_and_mpc_equal_the_host_driven_loops(integrator, B, N):
    """A user model's library has a persistent kernel of its own (csrc/solve_user.hip: one wave per trajectory, the generic
    device bodies): quattro_ilqr_solve_f32 / quattro_mpc_run_f32 as ONE launch against the host-driven loops of the same
    library, bit for bit — real exit tests, capped and fixed iteration counts, warm and cold starts, a disturbed closed loop
    and a second run that continues from the first one's warm start."""
    import torch
    import quattro_ilqr_amd as q
    md = planar_model(integrator)
    assert q.ops.model_can_device_loop(md) and not q.ops.model_has_device_loop(md)     # exists; enqueued iterations are the default
    x0, u0 = planar_batch(B, N, 11 * B + N)
    keys = ("K", "k", "x", "u", "cost", "iters", "alpha", "status")
    tw = 0 if N <= 10 else 10
    for kw in (dict(), dict(max_iter=2), dict(max_iter=3, fixed_iters=True)):
        for u_init in (u0, None):
            dev = q.QuattroILQR(md, N, max_iter=20, device="cuda:0", device_loop="always", tf_window=tw)
            host = q.QuattroILQR(md, N, max_iter=20, device="cuda:0", device_loop=False, check_every=1, tf_window=tw)
            od = {k: v.clone() for k, v in dev.solve(x0, u_init, **kw).items()}
            oh = host.solve(x0, u_init, **kw)
            for key in keys:
                assert torch.equal(od[key], oh[key]), (integrator, B, N, kw, u_init is None, key)
            assert torch.equal(dev.active, host.active) and torch.equal(dev.alpha_idx, host.alpha_idx)
    steps = 3
    rng = np.random.default_rng(B)
    dist = torch.as_tensor(1e-3 * rng.standard_normal((steps, B, 6)), dtype=torch.float32, device="cuda:0")
    a = q.BatchedMPC(md, N, max_iter=6, device="cuda:0", check_every=1, tf_window=tw)
    b = q.BatchedMPC(md, N, max_iter=6, device="cuda:0", check_every=1, tf_window=tw)
    for rep, d in enumerate((dist, None)):
        start = x0.astype(np.float32) if rep == 0 else oa["x"][:, -1].clone()
        oa = a.run(start, steps, disturbance=d, device_loop="always")
        ob = b.run(start, steps, disturbance=d, device_loop=False)
        for key in ("x", "u", "iters"):
            assert torch.equal(oa[key], ob[key].to(oa[key].dtype)), (rep, key)
        assert torch.equal(a.u_warm, b.u_warm)
        for name in ("K", "k", "x", "cost", "alpha_idx", "status"):
            assert torch.equal(getattr(a.solver, name), getattr(b.solver, name)), (rep, name)


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_user_model_against_the_reference_run_on_the_same_problem(integrator):
    """G13 (tests/golden/user_planar.npz): the REFERENCE's iLQR_TF, run in the build container on this problem as plain Python
    callables, against the compiled model on the GPU: iteration counts and accepted step sizes exactly, first-iteration gains
    at the reference's finite-difference noise, trajectories and optimum to fp32 round-off."""
    import torch
    import quattro_ilqr_amd as q
    from conftest import load_golden
    g = load_golden("user_planar.npz")
    assert np.array_equal(g["phys"], PHYS) and np.array_equal(g["Q"], Q) and np.array_equal(g["R"], R) and float(g["dt"]) == DT
    md = planar_model(integrator)
    N, S = int(g["N"]), g["x0"].shape[0]
    s = q.QuattroILQR(md, N, max_iter=int(g["max_iter"]), tol=float(g["tol"]), device="cuda:0", check_every=1)
    out = s.solve(torch.as_tensor(g["x0"], dtype=torch.float32), torch.as_tensor(g["u_init"], dtype=torch.float32))
    for b in range(S):
        key = f"{integrator}_s{b}_"
        n_it = int(g[key + "n_iter"])
        assert int(out["iters"][b]) == n_it
        assert float(out["alpha"][b]) == float(g[key + "alpha"][n_it - 1])          # last accepted step size
        ex = np.max(np.abs(out["x"][b].double().cpu().numpy() - g[key + "x_final"]))
        eu = np.max(np.abs(out["u"][b].double().cpu().numpy() - g[key + "u_final"]))
        print(f"planar/{integrator} b={b} vs the reference: {n_it} iterations, max|dx| {ex:.2e}, max|du| {eu:.2e}")
        assert ex < 1e-5 and eu < 3e-5
    # the whole alpha sequence and the first iteration's gains, through the drop-in class (one trajectory, the reference's loop)
    il = q.iLQR_TF(md, md, md, g["x0"][0], [u for u in g["u_init"][0]], N, dt=DT, max_iter=int(g["max_iter"]), tol=float(g["tol"]),
                   device="cuda:0")
    il.optimize(g["x_ref"])
    key = f"{integrator}_s0_"
    assert len(il.logs) == int(g[key + "n_iter"])
    assert [(-1.0 if lg["alpha"] is None else lg["alpha"]) for lg in il.logs] == list(g[key + "alpha"][:len(il.logs)])
    K0 = np.array(il.logs[0]["K_seq"])
    eK = np.linalg.norm(K0 - g[key + "K"][0]) / np.linalg.norm(g[key + "K"][0])
    print(f"planar/{integrator} first-iteration gains vs the reference: rel-Fro {eK:.2e}")
    assert eK < 1e-4


# ---------------------------------------------------------------------------------------------- the largest dims the header allows
CHAIN_RATE = """
// NU pendulum-like links coupled to their neighbours: x = (angles, rates), u = torques
for (int i = 0; i < NU; ++i) {
  xd[i] = x[NU + i];
  const T left = i > 0 ? x[i - 1] : x[i], right = i < NU - 1 ? x[i + 1] : x[i];
  xd[NU + i] = u[i] - sin(x[i]) * P[0] - x[NU + i] * P[1] + (left + right - x[i] * 2.0f) * P[2];
}
"""


def chain_rate(x, u, P=(2.0, 0.3, 1.5)):
    W = u.shape[0]
    xd = np.zeros(2 * W, dtype=np.result_type(x, u))
    for i in range(W):
        xd[i] = x[W + i]
        left = x[i - 1] if i > 0 else x[i]
        right = x[i + 1] if i < W - 1 else x[i]
        xd[W + i] = u[i] - np.sin(x[i]) * P[0] - x[W + i] * P[1] + (left + right - 2.0 * x[i]) * P[2]
    return xd


@pytest.mark.parametrize("W", [5, 8])
def test_user_model_at_the_largest_dimensions(W):
    """A chain of W links: n = 2 W, m = W.  W = 8 is n = 16, m = 8 (QUATTRO_MAX_NX / QUATTRO_MAX_NU: 32 lanes per linearisation
    item, an 8 x 8 pivoted inverse per sweep step, 128-float gain rows in the rollouts); W = 5 (n = 10, m = 5) is the middle
    case, 16 lanes per item.  Default diagonal cost with a control barrier: records vs complex-step fp64 derivatives, first
    sweep and a whole solve vs the oracle on the same problem as Python callables; the model's own device-resident loop bit
    for bit against its host-driven one."""
    import torch
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import _lib, ops
    dt, N, B = 0.02, 12, 4
    n, m = 2 * W, W
    qd = np.concatenate([np.full(W, 2.0), np.full(W, 0.2)]); rd = np.full(W, 0.05); qfd = np.concatenate([np.full(W, 20.0), np.full(W, 2.0)])
    xref = np.concatenate([0.2 * np.arange(W) / W, np.zeros(W)])
    md = q.compile_model(f"chain{n}x{m}", n, m, rate=CHAIN_RATE, dt=dt, integrator="rk4", phys=(2.0, 0.3, 1.5), q=qd, r=rd, qf=qfd,
                         x_ref=xref, barrier_alpha=0.5, barrier_beta=4.0)

    def f(x, u):
        k1 = chain_rate(x, u); k2 = chain_rate(x + 0.5 * dt * k1, u); k3 = chain_rate(x + 0.5 * dt * k2, u); k4 = chain_rate(x + dt * k3, u)
        return x + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)

    def softplus(z, beta):
        return np.log1p(np.exp(-np.abs(beta * z))) / beta + np.maximum(z, 0)       # (real arguments only)

    def L(x, u):
        d = x - xref
        return np.sum(qd * d * d) + np.sum(rd * u * u) + 0.5 * np.sum(softplus(-u, 4.0) ** 2)

    def Lf(x):
        d = x - xref
        return np.sum(qfd * d * d)

    rng = np.random.default_rng(4)
    x0 = (xref + 0.3 * rng.standard_normal((B, n))).astype(np.float32).astype(np.float64)
    u0 = (0.4 + 0.3 * rng.standard_normal((B, N, m))).astype(np.float32).astype(np.float64)
    dev = torch.device("cuda:0")
    x, _ = ops.simulate(md, torch.as_tensor(x0, dtype=torch.float32, device=dev), torch.as_tensor(u0, dtype=torch.float32, device=dev))
    xs = x.double().cpu().numpy()
    assert np.max(np.abs(xs[0] - O.rollout(f, x0[0], u0[0]))) < 2e-5
    rec, VxN, VxxN, layout = ops.linearize(md, x, torch.as_tensor(u0, dtype=torch.float32, device=dev))
    blocks = {k: v.double().cpu().numpy() for k, v in ops.unpack_derivs(rec, B, n, m, layout, lib=_lib.load_for(md)).items()}
    A = complex_step_jac(lambda z: f(z[:n], z[n:]), np.concatenate([xs[1, 3], u0[1, 3]]))
    assert np.max(np.abs(blocks["A"][1, 3] - A[:, :n])) < 2e-6 and np.max(np.abs(blocks["B"][1, 3] - A[:, n:])) < 2e-6
    d = O.linearize_fd(f, L, Lf, xs[1], [u for u in u0[1]])                  # the reference's finite differences
    assert np.max(np.abs(blocks["luu"][1] - d["luu"])) < 5e-4 and np.max(np.abs(blocks["lxx"][1] - d["lxx"])) < 5e-4
    assert np.max(np.abs(blocks["lu"][1] - d["lu"])) < 1e-5 * max(1.0, np.max(np.abs(d["lu"])))
    s = q.QuattroILQR(md, N, max_iter=15, tol=1e-3, device="cuda:0", tf_window=0)
    s.solve(x0, u0, max_iter=0)
    s.iterate()
    k_o, K_o = O.backward_pass(f, L, Lf, xs[0], [u for u in u0[0]])
    eK = np.linalg.norm(s.K[0].double().cpu().numpy() - np.array(K_o)) / np.linalg.norm(np.array(K_o))
    print(f"chain {n} x {m} first sweep vs the oracle: rel-Fro K {eK:.2e}")
    assert eK < 2e-4 and int(s.status.abs().sum()) == 0
    out = {k: v.clone() for k, v in s.solve(x0, u0).items()}
    for b in range(2):
        u_o, x_o, logs = O.optimize(f, L, Lf, x0[b], [u for u in u0[b]], N, max_iter=15, tol=1e-3)
        assert abs(len(logs) - int(out["iters"][b])) <= 1
        if len(logs) == int(out["iters"][b]):
            assert np.max(np.abs(out["x"][b].double().cpu().numpy() - x_o)) < 2e-4
    alw = q.QuattroILQR(md, N, max_iter=15, tol=1e-3, device="cuda:0", tf_window=0, device_loop="always")
    oa = alw.solve(x0, u0)
    for key in ("K", "k", "x", "u", "cost", "iters", "alpha"):
        assert torch.equal(oa[key], out[key]), key


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_user_model_runs_on_the_tile_sweep_and_equals_the_generic_one(integrator):
    """A user model with n <= 12, m <= 4 (the planar example: 6, 2) gets the MFMA tile sweep on its own ROWMAJOR records
    (layout ROWMAJOR_TILE, padded inside the kernel) — VERDICT r3 #8.  Against the generic pivoting sweep of the same library
    on the same records: gains per step <= 2e-6; whole solves take the same decisions."""
    import torch
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import _lib, ops
    md = planar_model(integrator)
    assert ops.model_layout(md) == _lib.LAYOUT_ROWMAJOR_TILE
    B, N = 301, 50
    x0, u0 = planar_batch(B, N, 17)
    x0t, u0t = torch.as_tensor(x0, dtype=torch.float32, device="cuda:0"), torch.as_tensor(u0, dtype=torch.float32, device="cuda:0")
    x, _ = ops.simulate(md, x0t, u0t)
    rec, VxN, VxxN, lay = ops.linearize(md, x, u0t)
    lib = _lib.load_for(md)
    Kt, kt, st = ops.riccati_sweep(rec, VxN, VxxN, 6, 2, _lib.LAYOUT_ROWMAJOR_TILE, lib=lib)
    Kg, kg, sg = ops.riccati_sweep(rec.reshape(B, N, -1), VxN, VxxN, 6, 2, _lib.LAYOUT_ROWMAJOR, lib=lib)
    assert int(st.abs().sum()) == 0 and int(sg.abs().sum()) == 0
    num = (Kt.double() - Kg.double()).flatten(2).norm(dim=2)
    eK = float((num / Kg.double().flatten(2).norm(dim=2)).max())
    # (k passes through zero along a trajectory: relative to the step's norm, floored at 1 % of the largest)
    ek = float(((kt.double() - kg.double()).norm(dim=2) / kg.double().norm(dim=2).clamp_min(1e-2 * float(kg.double().norm(dim=2).max()))).max())
    print(f"planar/{integrator}: tile sweep vs generic sweep on the same records: per-step K {eK:.2e} k {ek:.2e}")
    assert eK < 2e-6 and ek < 2e-5


def test_user_model_of_the_quadrotors_shape_one_sweep_in_all_three_paths():
    """a user model with (n, m) = (12, 4) — the quadrotor typed in again — used to linearise into TILE16 records for
    the host-driven path while its persistent kernel ran the generic sweep on ROWMAJOR records.  Now quattro_model_layout says
    ROWMAJOR_TILE for it and the host-driven loop, the one-call iteration and the persistent kernel run the SAME sweep body on the
    same records: bit for bit."""
    import torch
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import _lib, ops
    from bench import synthetic_batch
    builtin = q.quadrotor_model()
    user = q.compile_model("quadrotor_user", 12, 4, rate=QUAD_RATE, dt=builtin.dt, integrator="euler", phys=builtin.phys,
                           q=builtin.q, r=builtin.r, qf=builtin.qf, x_ref=builtin.x_ref,
                           barrier_alpha=builtin.barrier_alpha, barrier_beta=builtin.barrier_beta)
    assert ops.model_layout(user) == _lib.LAYOUT_ROWMAJOR_TILE
    B, N = 33, 50
    x0, u0 = synthetic_batch(B, 0)
    dev = q.QuattroILQR(user, N, max_iter=6, device="cuda:0", device_loop="always")
    host = q.QuattroILQR(user, N, max_iter=6, device="cuda:0", device_loop=False, check_every=1)
    od = {k: v.clone() for k, v in dev.solve(x0, u0).items()}
    oh = host.solve(x0, u0)
    for key in ("K", "k", "x", "u", "cost", "iters", "alpha", "status"):
        assert torch.equal(od[key], oh[key]), key
    assert int(od["iters"].max()) >= 2
