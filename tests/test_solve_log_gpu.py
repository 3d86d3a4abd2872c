"""The device-side per-iteration log of a solve (include/quattro_hip.h: quattro_solve_log, csrc/solve_log.h) and the
iLQR_TF drop-in that is built on it (one launch + one download per optimize() instead of a host round trip per step).

What the records must be: exactly what a host-driven loop sees between its kernels — the nominal entering the iteration,
its cost, the gains, the accepted step, the cost after it — for every trajectory and iteration, whoever wrote them (a
persistent kernel between its phases, or the stand-alone record kernel between the launches of an enqueued loop).
Reference for the contents: the log dict of iLQR_TF.optimize, quattro_ilqr_tf/quattro_ilqr_tf.py:453-466 / :565-578.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, rel_fro

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _pkg():
    import quattro_ilqr_amd as q
    return q


def _models():
    q = _pkg()
    from quattro_ilqr_amd import user_model
    return {
        "cartpole-euler": (q.cartpole_model(), 30), "cartpole-rk4": (q.cartpole_model(integrator="rk4"), 17),
        "quadrotor-euler": (q.quadrotor_model(), 50), "quadrotor-rk4": (q.quadrotor_model(integrator="rk4"), 26),
        "planar-user": (user_model.example_planar_model(), 25),
    }


def _batch(md, B, N, seed):
    rng = np.random.default_rng(seed)
    x0 = np.asarray(md.x_ref) + 0.2 * rng.standard_normal((B, md.n))
    hover = {"quadrotor": 2.4525, "planar_example": 4.905}.get(md.name, 0.0)
    u0 = hover + 0.1 * rng.standard_normal((B, N, md.m))
    return (torch.as_tensor(x0, dtype=torch.float32, device=DEV), torch.as_tensor(u0, dtype=torch.float32, device=DEV))


def _host_driven_records(q, md, N, x0, u0, max_iter, tol):
    """The same loop with one C call per iteration and a snapshot between the calls: list over iterations of dicts."""
    B = x0.shape[0]
    x, cost = q.ops.simulate(md, x0, u0)
    u = u0.clone()
    K = torch.zeros((B, N, md.m, md.n), dtype=torch.float32, device=DEV)
    k = torch.zeros((B, N, md.m), dtype=torch.float32, device=DEV)
    active = torch.ones((B,), dtype=torch.int32, device=DEV)
    iters = torch.zeros((B,), dtype=torch.int32, device=DEV)
    aidx = torch.full((B,), -1, dtype=torch.int32, device=DEV)
    status = torch.zeros((B,), dtype=torch.int32, device=DEV)
    ws = q.ops.workspace(md, B, N, DEV)
    recs = []
    for _ in range(max_iter):
        if int(active.sum()) == 0:
            break
        snap = dict(active=active.clone().cpu().numpy(), x=x.clone().cpu().numpy(), u=u.clone().cpu().numpy(),
                    cost_pre=cost.clone().cpu().numpy())
        q.ops.ilqr_iterate(md, x, u, K, k, cost, tol, ws, alpha_idx=aidx, active=active, iters=iters, status=status)
        snap.update(K=K.clone().cpu().numpy(), k=k.clone().cpu().numpy(), alpha_idx=aidx.clone().cpu().numpy(),
                    cost_new=cost.clone().cpu().numpy())
        recs.append(snap)
    return recs, dict(x=x.cpu().numpy(), u=u.cpu().numpy(), iters=iters.cpu().numpy())


@pytest.mark.parametrize("name", ["cartpole-euler", "cartpole-rk4", "quadrotor-euler", "quadrotor-rk4", "planar-user"])
@pytest.mark.parametrize("enqueue", [False, True])
def test_logged_solve_records_equal_the_host_driven_loop(name, enqueue):
    """Every record of the device log — persistent kernel (one launch) and enqueued iterations with the record kernel between
    them — against snapshots taken between the calls of a host-driven loop: bit for bit (the RK4 quadrotor's persistent loop
    linearises with another code than its record path: round-off there), ragged batch, stamps ordered."""
    q = _pkg()
    md, N = _models()[name]
    B, max_iter, tol = 7, 9, 1e-3
    x0, u0 = _batch(md, B, N, 3)
    ref, fin = _host_driven_records(q, md, N, x0, u0, max_iter, tol)
    sv = q.QuattroILQR(md, N, max_iter=max_iter, tol=tol, device=DEV, device_loop="always")
    log = q.ops.SolveLog(md, N, B, max_iter, DEV)
    sv._alloc(B)
    sv._upload(x0, u0)
    sv._ws = q.ops.workspace(md, B, N, DEV)
    q.ops.ilqr_solve(md, sv.x, sv.u, sv.K, sv.k, sv.cost, tol, max_iter, sv._ws, x0=sv._x0, alpha_idx=sv.alpha_idx,
                     active=sv.active, iters=sv.iters, status=sv.status, reset=True, log=log, persistent=not enqueue,
                     enqueue=enqueue)
    st = sv.download_state()
    exact = not (name == "quadrotor-rk4" and not enqueue)
    same = (lambda a, b: np.array_equal(a, b)) if exact else (lambda a, b: rel_fro(a, b) < 2e-4)
    if exact:
        assert np.array_equal(st["iters"], fin["iters"])
        assert np.array_equal(st["u"], fin["u"]) and np.array_equal(st["x"], fin["x"])
    for b in range(B):
        n_it = int(st["iters"][b])
        rows = log.rows(b, n_it)
        assert list(rows["iteration"]) == list(range(n_it))
        stp = rows["stamps"].astype(np.int64)
        assert np.all(np.diff(stp, axis=1) >= 0) and np.all(stp[1:, 0] >= stp[:-1, 3]) and np.all(stp[:, 3] > stp[:, 0])
        if not exact:
            continue
        for i in range(n_it):
            r = ref[i]
            assert r["active"][b] == 1
            assert same(rows["x"][i], r["x"][b]) and same(rows["u"][i], r["u"][b])
            assert same(rows["K"][i], r["K"][b]) and same(rows["k"][i], r["k"][b])
            assert rows["cost"][i, 0] == r["cost_pre"][b] and rows["cost"][i, 1] == r["cost_new"][b]
            assert rows["alpha_idx"][i] == r["alpha_idx"][b]
        assert n_it == len(ref) or ref[n_it]["active"][b] == 0          # ... and there is no further one


def test_log_ring_wraps_and_header_only_logs():
    """capacity < iterations: iteration i lands in slot i % capacity (the last `capacity` iterations survive); a header-only
    ring (no trajectories, no gains: what the drop-in uses with enable_log=False) still times and counts every iteration."""
    q = _pkg()
    md, N, B, max_iter = q.quadrotor_model(), 50, 3, 12
    x0, u0 = _batch(md, B, N, 11)
    sv = q.QuattroILQR(md, N, max_iter=max_iter, tol=1e-9, device=DEV)
    full = q.ops.SolveLog(md, N, B, max_iter, DEV)
    out = sv.solve(x0, u0, log=full)
    iters = out["iters"].cpu().numpy()
    assert int(iters.max()) > 4
    small = q.ops.SolveLog(md, N, B, 4, DEV)
    head = q.ops.SolveLog(md, N, B, max_iter, DEV, traj=False, gains=False)
    assert head.rec_bytes == 64 and small.rec_bytes == full.rec_bytes
    sv.solve(x0, u0, log=small)
    sv.solve(x0, u0, log=head)
    for b in range(B):
        n_it = int(iters[b])
        rf, rs, rh = full.rows(b, n_it), small.rows(b, 4), head.rows(b, n_it)
        assert "x" not in rh and np.array_equal(rh["alpha_idx"], rf["alpha_idx"]) and np.array_equal(rh["cost"], rf["cost"])
        for i in range(max(0, n_it - 4), n_it):
            s = i % 4
            assert rs["iteration"][s] == i and np.array_equal(rs["x"][s], rf["x"][i]) and np.array_equal(rs["K"][s], rf["K"][i])


def test_dropin_optimize_is_one_launch_and_matches_the_step_by_step_loop():
    """iLQR_TF.optimize through the persistent kernel + log ring against the same class driven step by step through its own
    per-step methods (simulate, backward_pass, forward_pass: the reference's loop): same log entries, bit for bit, same time
    list lengths; and with enable_log=False nothing is logged but everything is timed."""
    q = _pkg()
    g = load_golden("opt_quadrotor.npz")
    md, N = q.quadrotor_model(), 50
    mk = lambda **kw: q.iLQR_TF(None, None, None, g["s1_x0"], [np.zeros(4) for _ in range(N)], N, model=md,
                                max_iter=int(g["max_iter"]), tol=float(g["tol"]), device=DEV, **kw)
    a, b = mk(), mk()
    ua, xa = a.optimize(md.x_ref)
    ub, xb = b._optimize_host_loop(md, md.x_ref)
    assert len(a.logs) == len(b.logs) == int(g["s1_n_iter"])
    assert np.array_equal(np.array(ua), np.array(ub)) and np.array_equal(xa, xb)
    for la, lb in zip(a.logs, b.logs):
        assert set(la) == set(lb)
        for key in ("x_seq", "u_seq", "k_seq", "K_seq", "new_x_seq", "new_u_seq"):
            assert np.array_equal(np.array(la[key]), np.array(lb[key])), key
        assert la["alpha"] == lb["alpha"] and la["found_update"] == lb["found_update"] and la["iteration"] == lb["iteration"]
        # costs: the step-by-step loop re-evaluates them with the stand-alone cost kernel (another summation code than the
        # rollouts' own accumulation): fp32 stage-cost round-off
        assert abs(la["current_cost"] - lb["current_cost"]) <= 1e-6 * abs(lb["current_cost"])
        assert abs(la["new_cost"] - lb["new_cost"]) <= 1e-6 * abs(lb["new_cost"])
        assert isinstance(la["k_seq"], list) and la["k_seq"][0].shape == (4,) and la["K_seq"][0].shape == (4, 12)
        assert la["x_seq"].dtype == np.float64 and la["x_seq"].shape == (N + 1, 12)
    n_it = len(a.logs)
    assert len(a.backward_pass_time) == n_it and len(a.forward_pass_time) == n_it and len(a.total_time) == 1
    assert all(1e-6 < t < 1e-2 for t in a.backward_pass_time + a.forward_pass_time)
    assert sum(a.backward_pass_time) + sum(a.forward_pass_time) <= a.total_time[0]
    c = mk(enable_log=False)
    uc, xc = c.optimize(md.x_ref)
    assert c.logs == [] and len(c.backward_pass_time) == n_it and np.array_equal(np.array(uc), np.array(ua))
    # a second call continues from the warm start the first one left (self.u), like the reference
    u2, x2 = a.optimize(md.x_ref)
    assert len(a.logs) > n_it and a.logs[n_it]["iteration"] == 0 and len(a.total_time) == 2
    assert np.array_equal(a.logs[n_it]["x_seq"], xa)


def test_hybrid_dropin_on_the_device_takes_the_decisions_of_the_host_driven_loop():
    """Hybrid optimize() with the shipped cart-pole predictor: the device path (captured graph per iteration: tail sweep,
    predictor writing the gain stack, line search, log records) against the same predictor driven from the host through
    predict() (what a foreign predictor gets): same iteration count and accepted steps, same log keys, trajectories within
    the difference between the kernel's two output modes."""
    q = _pkg()
    g = load_golden("hybrid_cartpole.npz")

    def run(hide):
        tf = q.TransformerILQR(4, 5, device=DEV).load(os.path.join(GOLDEN, "tf_weights_cartpole.npz"))
        if hide:
            class HostOnly:                 # only the reference's duck type: predict() + prompt_len
                prompt_len = tf.prompt_len
                predict = staticmethod(tf.predict)
            tf_used = HostOnly()
        else:
            tf_used = tf
        mpc = q.CartPoleMPC(horizon=30, dt=0.01, integration_method="euler", transformer_model=tf_used, ilqr_tf_only=True,
                            device=DEV)
        mpc.ilqr.max_iter = int(g["max_iter"])
        mpc.ilqr.x0 = g["x0"]
        u, x = mpc.ilqr.optimize(mpc.x_ref)
        return mpc.ilqr, np.array(u), x

    dev, ud, xd = run(False)
    host, uh, xh = run(True)
    assert len(dev.logs) == len(host.logs) == int(g["n_iter"])
    assert [l["alpha"] for l in dev.logs] == [l["alpha"] for l in host.logs]
    assert set(dev.logs[0]) == set(host.logs[0]) and "K_seq_seg" in dev.logs[0]
    assert np.array(dev.logs[0]["K_seq_seg"]).shape == (5, 1, 4) and np.array(dev.logs[0]["k_seq_seg"]).shape == (5, 1)
    assert np.array_equal(np.array(dev.logs[0]["K_seq_seg"]), np.array(host.logs[0]["K_seq_seg"]))
    assert rel_fro(xd, xh) < 1e-3 and rel_fro(ud, uh) < 1e-2
    n_it = len(dev.logs)
    assert len(dev.inference_time) == n_it and len(dev.backward_pass_time) == n_it and len(dev.get_time()) == 4
    assert all(t > 0 for t in dev.inference_time)


# ------------------------------------------------------------------------------------------------ reference problem objects
class _StubDyn:
    pass


def _reference_shaped_problem(which, integrator="euler", mass_scale=1.0):
    """An object with exactly the attribute set of the reference's QuadrotorMPC / CartPoleMPC (examples/*/…_mpc.py) whose
    three methods evaluate the oracle's restatement of the reference's functions (the reference does not travel to the GPU box)."""
    from oracle import models as o_models
    integ = {"euler": o_models.INTEGRATOR_EULER, "rk4": o_models.INTEGRATOR_RK4}[integrator]
    spec = (o_models.quadrotor_spec if which == "quadrotor" else o_models.cartpole_spec)(dt=0.01, integrator=integ)
    dyn = _StubDyn()
    if which == "quadrotor":
        spec.phys["mass"] *= mass_scale
        dyn.m, dyn.Ix, dyn.Iy, dyn.Iz, dyn.arm, dyn.g = (spec.phys[k] for k in ("mass", "Ix", "Iy", "Iz", "arm", "gravity"))
    else:
        spec.phys["m_cart"] *= mass_scale
        dyn.m_cart, dyn.m_pole, dyn.length, dyn.gravity = (spec.phys[k] for k in ("m_cart", "m_pole", "length", "gravity"))

    class Problem:
        def discrete_dynamics(self, x, u):
            return spec.f(x, u)

        def running_cost(self, x, u):
            return spec.L(x, u)

        def final_cost(self, x):
            return spec.Lf(x)
    pb = Problem()
    pb.dynamics, pb.dt, pb.integration_method = dyn, 0.01, integrator
    pb.x_ref, pb.Q, pb.R, pb.Qf = spec.x_ref, spec.Q, spec.R, spec.Qf
    if which == "quadrotor":
        pb.alpha, pb.beta = spec.barrier_alpha, spec.barrier_beta
    return pb, spec


@pytest.mark.parametrize("which,N,integrator", [("quadrotor", 50, "euler"), ("cartpole", 30, "rk4"), ("quadrotor", 30, "rk4")])
def test_reference_shaped_problem_objects_bind_and_are_verified_on_the_device(which, N, integrator):
    """iLQR_TF(obj.discrete_dynamics, obj.running_cost, obj.final_cost, ...) with obj carrying the reference's attribute set:
    recognised, PROBED on the device against its own callables, and solved — the same solve as through the mirror classes;
    changed physical constants travel through the attributes; a callable that disagrees with the attributes is refused."""
    q = _pkg()
    g = load_golden(f"opt_{which}{'_rk4' if integrator == 'rk4' else ''}.npz")
    pb, spec = _reference_shaped_problem(which, integrator)
    m = spec.m
    il = q.iLQR_TF(pb.discrete_dynamics, pb.running_cost, pb.final_cost, g["s0_x0"], [np.zeros(m) for _ in range(N)], N,
                   max_iter=int(g["max_iter"]), tol=float(g["tol"]), device=DEV)
    md = il._model()
    assert md == q.model_by_name(which, integrator=integrator) and il._model() is md
    u, x = il.optimize(pb.x_ref)
    n_it = int(g["s0_n_iter"])
    assert len(il.logs) == n_it and [(-1.0 if l["alpha"] is None else l["alpha"]) for l in il.logs] == list(g["s0_alpha"][:n_it])
    ref = q.iLQR_TF(None, None, None, g["s0_x0"], [np.zeros(m) for _ in range(N)], N, model=md, max_iter=int(g["max_iter"]),
                    tol=float(g["tol"]), device=DEV)
    u2, x2 = ref.optimize(pb.x_ref)
    assert np.array_equal(np.array(u), np.array(u2)) and np.array_equal(x, x2)
    # other physical constants: carried by the attributes, verified by the probe, visible in the solution
    pb2, _ = _reference_shaped_problem(which, integrator, mass_scale=1.3)
    il2 = q.iLQR_TF(pb2.discrete_dynamics, pb2.running_cost, pb2.final_cost, g["s0_x0"], [np.zeros(m) for _ in range(N)], N,
                    max_iter=3, device=DEV)
    assert abs(il2._model().phys[0] - 1.3 * md.phys[0]) < 1e-6
    u3, _ = il2.optimize(pb2.x_ref)
    assert not np.array_equal(np.array(u3), np.array(u))
    # attributes that claim one mass while the callable integrates another: refused at construction
    pb3, _ = _reference_shaped_problem(which, integrator, mass_scale=1.3)
    if which == "quadrotor":
        pb3.dynamics.m = 1.0
    else:
        pb3.dynamics.m_cart = 1.0
    with pytest.raises(NotImplementedError, match="do not compute"):
        q.iLQR_TF(pb3.discrete_dynamics, pb3.running_cost, pb3.final_cost, g["s0_x0"], [np.zeros(m) for _ in range(N)], N, device=DEV)


# ------------------------------------------------------------------------------------------------ torch-free hybrid backward pass
@pytest.mark.parametrize("name", ["cartpole-euler", "cartpole-rk4", "quadrotor-euler", "quadrotor-rk4"])
def test_tail_sweep_written_in_place_equals_the_segment_form(name):
    """quattro_linearize_sweep_rows_f32 (k_rows = N): the swept steps t_start .. N-1 land at rows t_start .. N-1 of the FULL gain
    stacks, bit-identical to the segment form (index t - t_start); the rows below and inactive trajectories are not touched."""
    q = _pkg()
    md, N = _models()[name]
    B = 9
    x0, u0 = _batch(md, B, N, 5)
    x, _ = q.ops.simulate(md, x0, u0)
    active = torch.ones((B,), dtype=torch.int32, device=DEV)
    active[3] = 0
    for t_start in (N - 1, N - 5, max(0, N - 13), 0):
        Ks, ks, _ = q.ops.linearize_sweep(md, x, u0, t_start)
        K = torch.full((B, N, md.m, md.n), -7.0, dtype=torch.float32, device=DEV)
        k = torch.full((B, N, md.m), -7.0, dtype=torch.float32, device=DEV)
        q.ops.linearize_sweep(md, x, u0, t_start, K=K, k=k, active=active, in_place=True)
        live = active.bool()
        assert torch.equal(K[live][:, t_start:], Ks[live]) and torch.equal(k[live][:, t_start:], ks[live])
        assert bool((K[:, :t_start] == -7.0).all()) and bool((k[:, :t_start] == -7.0).all())
        assert bool((K[3] == -7.0).all()) and bool((k[3] == -7.0).all())


@pytest.mark.parametrize("which", ["quadrotor", "cartpole"])
def test_predictor_reads_its_prompt_from_the_gain_rows(which):
    """quattro_tf_gains_* with prompt = NULL reads [k | K.flat] from rows N - P .. N - 1 of the stacks: bit-identical to handing
    it the packed prompt (quattro_ilqr_tf.py:498-502), for the shipped checkpoints (P = 1 and P = 5)."""
    q = _pkg()
    n, m, N = (12, 4, 50) if which == "quadrotor" else (4, 1, 30)
    tf = q.TransformerILQR(n, m * (1 + n), device=DEV).load(os.path.join(GOLDEN, f"tf_weights_{which}.npz"))
    P, T = tf.prompt_len, tf.target_len
    assert P + T == N
    rng = np.random.default_rng(8)
    B = 6
    x = torch.as_tensor(rng.standard_normal((B, N + 1, n)) * 0.2, dtype=torch.float32, device=DEV)
    K = torch.as_tensor(rng.standard_normal((B, N, m, n)), dtype=torch.float32, device=DEV)
    k = torch.as_tensor(rng.standard_normal((B, N, m)), dtype=torch.float32, device=DEV)
    K2, k2 = K.clone(), k.clone()
    prompt = torch.cat([k[:, N - P:], K[:, N - P:].reshape(B, P, -1)], dim=-1).contiguous()
    tf.predict_gains(x, prompt, K, k)
    tf.predict_gains(x, None, K2, k2)
    assert torch.equal(K, K2) and torch.equal(k, k2)
    assert not torch.equal(K[:, :T], torch.zeros_like(K[:, :T]))


def test_dropin_edge_cases_of_the_one_launch_optimize(capsys):
    """What callers of the reference's class may do between calls: max_iter = 0, max_iter raised beyond the log ring, enable_log
    toggled, verbose output, the default-constructed QuadrotorMPC (horizon 30, RK4) over several warm-started control steps."""
    q = _pkg()
    md, N = q.quadrotor_model(), 50
    g = load_golden("opt_quadrotor.npz")
    x0 = g["s0_x0"]
    il = q.iLQR_TF(None, None, None, x0, [np.zeros(4) for _ in range(N)], N, model=md, max_iter=0, device=DEV)
    u, x = il.optimize(md.x_ref)                                      # no iteration at all: (u, simulate(u)), nothing logged
    assert il.logs == [] and len(il.total_time) == 1 and il.backward_pass_time == []
    assert np.array_equal(np.array(u), np.zeros((N, 4))) and np.array_equal(x, il.simulate(u))
    il.max_iter = 2
    il.optimize(md.x_ref)
    assert len(il.logs) == 2 and il.total_iter == 1
    il.max_iter = 40                                                  # beyond the ring built for max_iter = 2: a new ring
    il.u = [np.zeros(4) for _ in range(N)]
    il.logs = []
    il.optimize(md.x_ref, verbose=True)
    n_it = int(g["s0_n_iter"]) if int(g["max_iter"]) >= 40 else len(il.logs)
    assert len(il.logs) == n_it >= 6 and [l["iteration"] for l in il.logs] == list(range(n_it))
    out = capsys.readouterr().out
    assert out.count("Iteration") == n_it and "Alpha:" in out
    il.enable_log = False                                             # toggled: header-only ring from now on
    il.u = [np.zeros(4) for _ in range(N)]
    n_logs, n_bt = len(il.logs), len(il.backward_pass_time)
    u2, _ = il.optimize(md.x_ref)
    assert len(il.logs) == n_logs and len(il.backward_pass_time) == n_bt + n_it
    assert np.array_equal(np.array(u2), np.array(il.u))
    # the reference's default construction: QuadrotorMPC() = horizon 30, RK4 (quadrotor_mpc.py:12), warm-started control steps
    mpc = q.QuadrotorMPC(device=DEV)
    assert mpc.horizon == 30 and mpc.integration_method == "rk4"
    mpc.ilqr.max_iter = 5
    x_cur = np.asarray(x0, dtype=np.float64)
    its = []
    for step in range(4):
        n0 = len(mpc.ilqr.logs)
        xs, us = mpc.control_step(x_cur)
        its.append(len(mpc.ilqr.logs) - n0)
        assert xs.shape == (31, 12) and len(us) == 30 and len(mpc.ilqr.u) == 30
        assert np.array_equal(mpc.ilqr.u[-1], mpc.ilqr.u[-2]) and np.array_equal(mpc.ilqr.u[0], us[1])
        x_cur = mpc.discrete_dynamics(x_cur, us[0])
    assert its[0] >= 1 and its[-1] <= its[0] and np.all(np.isfinite(x_cur))
    assert len(mpc.ilqr.total_time) == 4 and len(mpc.ilqr.backward_pass_time) == sum(its)
