"""GPU parity of the solver layer: the iLQR_TF drop-in and the batched QuattroILQR against the reference's logged
optimize() runs (golden G6/G8/G9) and against each other.

What can match the reference and how closely (measured in tests/test_kernels_gpu.py): the reference's gains
carry its own finite-difference round-off (K moves 1e-4 on these trajectories when only its second differences are
replaced by exact values), so after a few iterations states/controls agree to ~1e-4, not 1e-5; the discrete decisions
(iteration count, accepted alpha per iteration, found_update) are asserted EXACTLY.
"""
import numpy as np
import pytest

from conftest import load_golden, rel_fro

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _pkg():
    import quattro_ilqr_amd as q
    return q


def _alpha_list(logs):
    return [(-1.0 if lg["alpha"] is None else lg["alpha"]) for lg in logs]


# End-to-end distance of the drop-in to the reference's logged optimize() runs: bounds = 2 x the values measured on MI355X in
# round 3 (gpurun_out/pytest_r03g.log; printed again on every run), with a floor of 1e-6 where the measured value is at fp32
# round-off.  K0 = gains of the first iteration (identical nominal): the reference's finite-difference noise floor.
#                          measured:  K0       x        cost     x_final  u_final
#   cart-pole  Euler                  1.30e-4  5.43e-5  1.28e-6  2.45e-6  8.17e-6
#   quadrotor  Euler                  3.18e-6  1.89e-5  5.83e-5  3.86e-5  2.05e-4
#   cart-pole  RK4                    1.27e-4  5.29e-5  9.60e-7  2.93e-6  4.96e-6
#   quadrotor  RK4 (N = 30)           3.74e-6  2.32e-7  1.75e-7  8.32e-8  1.02e-7
_E2E_BOUNDS = {
    ("cartpole", "euler"): dict(K0=2.6e-4, x=1.1e-4, cost=2.6e-6, x_final=5.0e-6, u_final=1.7e-5),
    ("quadrotor", "euler"): dict(K0=6.4e-6, x=3.8e-5, cost=1.2e-4, x_final=7.8e-5, u_final=4.1e-4),
    ("cartpole", "rk4"): dict(K0=2.6e-4, x=1.1e-4, cost=2.0e-6, x_final=6.0e-6, u_final=1.0e-5),
    ("quadrotor", "rk4"): dict(K0=7.5e-6, x=1.0e-6, cost=1.0e-6, x_final=1.0e-6, u_final=1.0e-6),
}


# ------------------------------------------------------------------------------------------------ G6: drop-in vs reference logs
@pytest.mark.parametrize("model,N,integ", [("cartpole", 30, "euler"), ("quadrotor", 50, "euler"),
                                           ("cartpole", 30, "rk4"), ("quadrotor", 30, "rk4")])
def test_ilqr_tf_dropin_follows_reference_logs(model, N, integ):
    """Euler = what both shipped simulators pass; RK4 = the default integration_method of the MPC classes."""
    q = _pkg()
    g = load_golden(f"opt_{model}{'_rk4' if integ == 'rk4' else ''}.npz")
    md = q.model_by_name(model, integrator=integ)
    meas = dict(K0=0.0, x=0.0, cost=0.0, x_final=0.0, u_final=0.0)
    for s in range(int(g["n_states"])):
        il = q.iLQR_TF(None, None, None, g[f"s{s}_x0"], [np.zeros(md.m) for _ in range(N)], N, model=md,
                       max_iter=int(g["max_iter"]), tol=float(g["tol"]), device=DEV)
        u_fin, x_fin = il.optimize(md.x_ref)
        n_it = int(g[f"s{s}_n_iter"])
        assert len(il.logs) == n_it, (s, len(il.logs), n_it)
        assert _alpha_list(il.logs) == list(g[f"s{s}_alpha"][:n_it]), s
        assert [int(lg["found_update"]) for lg in il.logs] == list(g[f"s{s}_found"][:n_it])
        # return conventions of the reference
        assert isinstance(u_fin, list) and len(u_fin) == N and u_fin[0].shape == (md.m,) and u_fin[0].dtype == np.float64
        assert x_fin.shape == (N + 1, md.n) and il.u is u_fin
        assert set(il.logs[0]) == {"iteration", "x_seq", "u_seq", "current_cost", "k_seq", "K_seq", "alpha", "new_x_seq",
                                   "new_u_seq", "new_cost", "found_update"}
        assert len(il.backward_pass_time) == n_it and len(il.total_time) == 1 and len(il.forward_pass_time) >= n_it
        # first iteration: identical nominal, so everything is at fp32 accuracy
        lg0 = il.logs[0]
        assert rel_fro(lg0["x_seq"], g[f"s{s}_x_seq"][0]) < 1e-6
        assert abs(lg0["current_cost"] - g[f"s{s}_current_cost"][0]) <= 1e-6 * abs(g[f"s{s}_current_cost"][0])
        meas["K0"] = max(meas["K0"], rel_fro(np.array(lg0["K_seq"]), g[f"s{s}_K"][0]))     # reference FD noise floor (see module doc)
        # whole run
        for i, lg in enumerate(il.logs):
            meas["x"] = max(meas["x"], rel_fro(lg["x_seq"], g[f"s{s}_x_seq"][i]))
            meas["cost"] = max(meas["cost"], abs(lg["current_cost"] - g[f"s{s}_current_cost"][i]) / abs(g[f"s{s}_current_cost"][i]))
        meas["x_final"] = max(meas["x_final"], rel_fro(x_fin, g[f"s{s}_x_final"]))
        meas["u_final"] = max(meas["u_final"], np.max(np.abs(np.array(u_fin) - g[f"s{s}_u_final"])) / max(1.0, np.max(np.abs(g[f"s{s}_u_final"]))))
    # End-to-end distance to the reference's logged runs.  It is NOT this path's error: the reference's gains carry the
    # round-off of its own finite-difference Hessians (4 eps |L| / (4 eps_fd^2); control experiment in
    # tests/test_kernels_gpu.py: the reference's own fp64 sweep with exact second derivatives moves K by the same amount), and
    # the iterations amplify it.  The bounds are <= 2x the values measured on MI355X (round 3; printed on every run).
    bound = _E2E_BOUNDS[(model, integ)]
    print(f"end-to-end vs reference logs, {model} {integ}: " + ", ".join(f"{k} {v:.2e} (bound {bound[k]:.1e})" for k, v in meas.items()))
    for k, v in meas.items():
        assert v <= bound[k], (model, integ, k, v, bound[k])


def test_ilqr_tf_methods_match_reference_single_calls():
    """backward_pass / backward_pass_segment / forward_pass / simulate / compute_total_cost, one call each (G4/G5)."""
    q = _pkg()
    g = load_golden("fwd_quadrotor.npz")
    md = q.quadrotor_model()
    N = 50
    il = q.iLQR_TF(None, None, None, g["x0"][0], list(g["u_seq"][0]), N, model=md, device=DEV)
    xs = il.simulate(list(g["u_seq"][0]))
    assert xs.dtype == np.float64 and rel_fro(xs, g["x_seq"][0]) < 1e-6
    assert abs(il.compute_total_cost(g["x_seq"][0], list(g["u_seq"][0])) - g["cost0"][0]) < 1e-6 * g["cost0"][0]
    k_seq, K_seq = il.backward_pass(g["x_seq"][0], list(g["u_seq"][0]))
    assert len(k_seq) == N and K_seq[0].shape == (4, 12) and k_seq[0].shape == (4,)
    assert rel_fro(np.array(K_seq), g["K"][0]) < 5e-4
    ks, Ks = il.backward_pass_segment(g["x_seq"][0], list(g["u_seq"][0]), N - 7)
    assert len(ks) == 7 and np.array_equal(np.array(Ks), np.array(K_seq)[N - 7:])      # same recursion, index t - start
    nx, nu, nj = il.forward_pass(g["x_seq"][0], list(g["u_seq"][0]), list(g["k"][0]), list(g["K"][0]), 0.25)
    assert rel_fro(nx, g["new_x"][0, 2]) < 1e-5 and rel_fro(np.array(nu), g["new_u"][0, 2]) < 1e-5
    assert abs(nj - g["new_cost"][0, 2]) < 1e-5 * g["new_cost"][0, 2]
    with pytest.raises(IndexError):
        il.forward_pass(g["x_seq"][0], list(g["u_seq"][0]), list(g["k"][0])[:30], list(g["K"][0])[:30], 1.0)
    assert len(il.backward_pass_time) == 2 and len(il.forward_pass_time) == 1


# ------------------------------------------------------------------------------------------------ batched solver
@pytest.mark.parametrize("model,N", [("cartpole", 30), ("quadrotor", 50)])
def test_batched_solve_equals_per_trajectory_dropin(model, N):
    """QuattroILQR.solve over a batch == iLQR_TF.optimize one trajectory at a time (same kernels, same decisions)."""
    q = _pkg()
    g = load_golden(f"opt_{model}.npz")
    md = q.model_by_name(model)
    S = int(g["n_states"])
    x0 = np.stack([g[f"s{s}_x0"] for s in range(S)])
    solver = q.QuattroILQR(md, N, max_iter=int(g["max_iter"]), tol=float(g["tol"]), device=DEV, check_every=1)
    out = solver.solve(x0)
    iters = out["iters"].cpu().numpy()
    assert list(iters) == [int(g[f"s{s}_n_iter"]) for s in range(S)]
    assert out["K"].shape == (S, N, md.m, md.n) and out["x"].shape == (S, N + 1, md.n) and out["cost"].dtype == torch.float64
    for s in range(S):
        il = q.iLQR_TF(None, None, None, x0[s], [np.zeros(md.m) for _ in range(N)], N, model=md,
                       max_iter=int(g["max_iter"]), tol=float(g["tol"]), device=DEV)
        u_fin, x_fin = il.optimize(md.x_ref)
        assert np.array_equal(out["u"][s].double().cpu().numpy(), np.array(u_fin))
        assert np.array_equal(out["x"][s].double().cpu().numpy(), x_fin)
        assert np.array_equal(out["K"][s].double().cpu().numpy(), np.array(il.logs[-1]["K_seq"]))
        last_alpha = il.logs[-1]["alpha"]
        assert float(out["alpha"][s]) == (-1.0 if last_alpha is None else np.float32(last_alpha))
        assert rel_fro(out["x"][s].cpu().numpy(), g[f"s{s}_x_final"]) < 2e-4
    assert int(out["status"].abs().sum()) == 0


def test_batched_solve_large_batch_properties():
    """BASELINE-size batch: size-independent properties (cost never increases, replicated inputs give replicated
    outputs, inactive trajectories are frozen)."""
    q = _pkg()
    md = q.quadrotor_model()
    N, B = 50, 4096
    rng = np.random.default_rng(1234)
    x0 = np.asarray(md.x_ref) + rng.uniform(-1, 1, (B, 12)) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    x0[B // 2:] = x0[:B // 2]                                   # second half replicates the first
    u0 = 2.4525 + 0.1 * rng.standard_normal((B, N, 4))
    u0[B // 2:] = u0[:B // 2]
    solver = q.QuattroILQR(md, N, max_iter=12, device=DEV)
    x0_t = torch.as_tensor(x0, dtype=torch.float32, device=DEV)
    u0_t = torch.as_tensor(u0, dtype=torch.float32, device=DEV)
    _, J0 = q.ops.simulate(md, x0_t, u0_t)
    out = solver.solve(x0, u0)
    J = out["cost"]
    assert bool((J <= J0).all())
    assert torch.equal(out["u"][:B // 2], out["u"][B // 2:]) and torch.equal(out["K"][:B // 2], out["K"][B // 2:])
    assert torch.equal(out["iters"][:B // 2], out["iters"][B // 2:])
    assert int(out["status"].abs().sum()) == 0
    assert int(out["iters"].min()) >= 1 and int(out["iters"].max()) <= 12
    # x is the rollout of u from x0 (the reference returns simulate(u_seq))
    xs, Js = q.ops.simulate(md, x0_t, out["u"].contiguous())
    assert torch.equal(xs, out["x"]) and torch.equal(Js, J)


# ------------------------------------------------------------------------------------------------ G8: hybrid control flow
@pytest.mark.parametrize("model", ["quadrotor", "cartpole"])
def test_hybrid_dropin_replays_reference_run(model):
    """iLQR_TF with a transformer: prompt layout [k | K.flat], x_err = x - x_ref + offset, prediction unpacked as
    (T, m, 1+n), gain stack = predicted T + swept P steps (quadrotor P = 1, cart-pole P = 5: a multi-row prompt and a
    5-step swept segment).  The predictor here replays the reference's logged predictions, so the test isolates the
    solver's indexing and layout (bit-exact integer behaviour)."""
    q = _pkg()
    g = load_golden(f"hybrid_{model}.npz")
    N, n, m = (50, 12, 4) if model == "quadrotor" else (30, 4, 1)
    P = int(g["tf_window"])
    calls = []

    class Replay:
        prompt_len = P

        def predict(self, x_err, prompt):
            calls.append((np.array(x_err), np.array(prompt)))
            return g["prediction"][len(calls) - 1]

    if model == "quadrotor":
        mpc = q.QuadrotorMPC(horizon=N, dt=0.01, integration_method="euler", transformer_model=Replay(), device=DEV)
    else:
        mpc = q.CartPoleMPC(horizon=N, dt=0.01, integration_method="euler", transformer_model=Replay(), ilqr_tf_only=True,
                            device=DEV)
    mpc.ilqr.max_iter = int(g["max_iter"])
    mpc.ilqr.x0 = g["x0"]
    assert mpc.ilqr.tf_window == P and np.array_equal(mpc.ilqr.get_state_offset(), g["state_offset"])
    u_fin, x_fin = mpc.ilqr.optimize(mpc.x_ref)
    n_it = int(g["n_iter"])
    assert len(mpc.ilqr.logs) == n_it and len(calls) == n_it
    assert _alpha_list(mpc.ilqr.logs) == list(g["alpha"][:n_it])
    assert set(mpc.ilqr.logs[0]) >= {"k_seq_seg", "K_seq_seg"} and "K_seq" not in mpc.ilqr.logs[0]
    assert len(mpc.ilqr.inference_time) == n_it and len(mpc.ilqr.get_time()) == 4
    for i, (x_err, prompt) in enumerate(calls):
        assert prompt.shape == (P, m * (1 + n)) and x_err.shape == (N + 1, n)
        assert rel_fro(x_err, g["x_err"][i]) < 2e-4
        assert rel_fro(prompt, g["prompt"][i]) < 5e-4
        lg = mpc.ilqr.logs[i]
        assert np.array_equal(prompt[:, :m], np.array(lg["k_seq_seg"])) and \
            np.array_equal(prompt[:, m:], np.array(lg["K_seq_seg"]).reshape(P, m * n))
    assert rel_fro(x_fin, g["x_final"]) < 2e-4


def test_hybrid_cartpole_with_the_hip_predictor():
    """The shipped cart-pole checkpoint (prompt 5, target 25, L = 61: the 64-token variant of the transformer kernel) in
    the loop: drop-in and batched solver agree with each other and follow the reference's run (its predictor ran in fp16
    on the CPU, this one in bf16 on MFMA: same decisions, trajectories within the predictors' difference)."""
    import os
    from conftest import GOLDEN
    q = _pkg()
    g = load_golden("hybrid_cartpole.npz")
    tf = q.TransformerILQR(4, 5, device=DEV).load(os.path.join(GOLDEN, "tf_weights_cartpole.npz"))
    assert tf.prompt_len == 5 and tf.target_len == 25
    mpc = q.CartPoleMPC(horizon=30, dt=0.01, integration_method="euler", transformer_model=tf, ilqr_tf_only=True, device=DEV)
    mpc.ilqr.max_iter = int(g["max_iter"])
    mpc.ilqr.x0 = g["x0"]
    u_fin, x_fin = mpc.ilqr.optimize(mpc.x_ref)
    n_it = int(g["n_iter"])
    assert len(mpc.ilqr.logs) == n_it and _alpha_list(mpc.ilqr.logs) == list(g["alpha"][:n_it])
    assert rel_fro(x_fin, g["x_final"]) < 5e-2
    s = q.QuattroILQR(mpc.device_model(), 30, max_iter=int(g["max_iter"]), tol=1e-1, tf=tf, device=DEV)
    out = s.solve(g["x0"][None])
    assert int(out["iters"][0]) == n_it
    assert np.array_equal(out["x"][0].double().cpu().numpy(), x_fin)            # one device path for both (round 4)


def test_hybrid_solve_with_a_default_shaped_predictor():
    """A predictor of the reference constructor's default shape (d_model 64, 8 heads: no fused kernel, forward through the
    layer-wise fp32 kernels) in the hybrid loop: the drop-in's single-trajectory optimize() and the batched solver take the
    same decisions and end on the same trajectory."""
    q = _pkg()
    from quattro_ilqr_amd import training
    n, m, N, P = 4, 1, 30, 5
    c, T = m * (1 + n), N - P                               # hybrid window: T predicted steps + P swept steps = N
    params, buffers = training.init_params(n, c, 64, 8, 2, 128, 100, T, seed=11, device="cpu")
    W = {k: (0.3 * v).detach().numpy() for k, v in params.items()}
    W["pos_encoder.pe"] = buffers["pos_encoder.pe"].numpy()
    norm = dict(x_mean=np.zeros(n), x_std=np.ones(n), u_mean=np.zeros(c), u_std=0.05 * np.ones(c))
    hp = dict(target_len=T, prompt_len=P, state_dim=n, control_dim=c, d_model=64, nhead=8, num_decoder_layers=2,
              dim_feedforward=128, dropout=0.1, max_seq_len=100)
    tf = q.TransformerILQR(n, c, device=DEV).load_arrays(W, norm, hp)
    assert not tf.fused_kernel_covers()
    mpc = q.CartPoleMPC(horizon=N, dt=0.01, integration_method="euler", transformer_model=tf, ilqr_tf_only=True, device=DEV)
    mpc.ilqr.max_iter = 4
    x0 = np.array([0.1, 0.0, 0.15, 0.0])
    mpc.ilqr.x0 = x0
    u_fin, x_fin = mpc.ilqr.optimize(mpc.x_ref)
    s = q.QuattroILQR(mpc.device_model(), N, max_iter=4, tol=mpc.ilqr.tol, tf=tf, device=DEV)
    out = s.solve(x0[None])
    assert int(out["iters"][0]) == len(mpc.ilqr.logs)
    assert rel_fro(out["x"][0].double().cpu().numpy(), x_fin) < 1e-4
    with pytest.raises(NotImplementedError):
        q.QuattroILQR(mpc.device_model(), N, max_iter=2, tf=tf, device=DEV, use_graph=True).solve(x0[None])


# ------------------------------------------------------------------------------------------------ G9: MPC mirrors
@pytest.mark.parametrize("model,N", [("cartpole", 30), ("quadrotor", 50)])
def test_mpc_control_step_warm_start(model, N):
    q = _pkg()
    g = load_golden(f"warm_{model}.npz")
    if model == "quadrotor":
        mpc = q.QuadrotorMPC(horizon=N, dt=0.01, integration_method="euler", device=DEV)
    else:
        mpc = q.CartPoleMPC(horizon=N, dt=0.01, integration_method="euler", ilqr_only=True, device=DEV)
        assert mpc.ilqr.tol == 1e-1
    mpc.ilqr.max_iter = int(g["max_iter"])
    xs1, r1 = mpc.control_step(g["x_a"])
    assert len(mpc.ilqr.logs) == int(g["n_iter1"])
    assert len(mpc.ilqr.u) == N and rel_fro(np.array(mpc.ilqr.u), g["u_warm"]) < 1e-3
    assert np.array_equal(mpc.ilqr.u[-1], mpc.ilqr.u[-2])                      # last input held
    n1 = len(mpc.ilqr.logs)
    xs2, r2 = mpc.control_step(g["x_b"])
    assert len(mpc.ilqr.logs) - n1 == int(g["n_iter2"])
    assert rel_fro(xs2, g["x_step2"]) < 2e-4
    assert rel_fro(np.array(mpc.ilqr.u), g["u_warm2"]) < 1e-3
    # the reference callables exist on the mirror and evaluate the same functions (one point, on the device)
    dc = load_golden(f"dyn_cost_{model}.npz")
    x, u = dc["x"][0], dc["u"][0]
    assert np.max(np.abs(mpc.discrete_dynamics(x, u) - dc["f_euler"][0])) < 1e-5
    assert abs(mpc.running_cost(x, u) - dc["L"][0]) < 1e-5 * abs(dc["L"][0])
    assert abs(mpc.final_cost(x) - dc["Lf"][0]) < 1e-5 * abs(dc["Lf"][0])


# ------------------------------------------------------------------------------------------------ §8f rank 1: batched MPC loop
@pytest.mark.parametrize("model,N", [("cartpole", 30), ("quadrotor", 50)])
def test_batched_mpc_equals_per_controller_dropin(model, N):
    """BatchedMPC.run (B controllers, device plant, warm-start shift on the device) == one reference-style
    QuadrotorMPC / CartPoleMPC per controller driven step by step, bit for bit."""
    q = _pkg()
    g = load_golden(f"opt_{model}.npz")
    S = 3
    x0 = np.stack([g[f"s{s}_x0"] for s in range(S)]).astype(np.float32).astype(np.float64)
    steps, max_iter = 3, 4
    if model == "quadrotor":
        make = lambda: q.QuadrotorMPC(horizon=N, dt=0.01, integration_method="euler", device=DEV)
    else:
        make = lambda: q.CartPoleMPC(horizon=N, dt=0.01, integration_method="euler", ilqr_only=True, device=DEV)
    proto = make()
    bm = q.BatchedMPC(proto.device_model(), N, max_iter=max_iter, tol=proto.ilqr.tol, device=DEV, check_every=1)
    out = bm.run(x0, steps)
    assert out["x"].shape == (S, steps + 1, proto.device_model().n) and out["iters"].shape == (S, steps)
    for s in range(S):
        mpc = make()
        mpc.ilqr.max_iter = max_iter
        x = x0[s]
        for t in range(steps):
            n_logs = len(mpc.ilqr.logs)
            res = mpc.control_step(x)
            u0 = res[1][0] if model == "quadrotor" else res[1]
            assert int(out["iters"][s, t]) == len(mpc.ilqr.logs) - n_logs
            assert np.array_equal(out["u"][s, t].double().cpu().numpy(), np.asarray(u0))
            x = mpc.discrete_dynamics(x, u0)
            assert np.array_equal(out["x"][s, t + 1].double().cpu().numpy(), x)


def test_graph_replay_equals_eager_launches():
    """use_graph=True: one iteration captured into a hipGraph and replayed gives bit-identical results (pure + hybrid)."""
    q = _pkg()
    import os
    from conftest import GOLDEN
    for model, N, B in (("cartpole", 50, 1024), ("quadrotor", 50, 256)):
        md = q.model_by_name(model)
        rng = np.random.default_rng(5)
        x0 = np.asarray(md.x_ref) + 0.2 * rng.standard_normal((B, md.n))
        outs = []
        for use_graph in (False, True):
            s = q.QuattroILQR(md, N, max_iter=6, device=DEV, use_graph=use_graph)
            o = s.solve(x0)
            outs.append({k: v.clone() for k, v in o.items()})
        for k in ("x", "u", "K", "k", "cost", "iters"):
            assert torch.equal(outs[0][k], outs[1][k]), (model, k)
    tf = q.TransformerILQR(12, 52, device=DEV).load(os.path.join(GOLDEN, "tf_weights_quadrotor.npz"))
    md = q.quadrotor_model()
    x0 = np.asarray(md.x_ref) + 0.05 * np.random.default_rng(6).standard_normal((64, 12))
    outs = []
    for use_graph in (False, True):
        s = q.QuattroILQR(md, 50, max_iter=3, tf=tf, device=DEV, use_graph=use_graph, state_offset=np.eye(12)[2] * 0.5)
        o = s.solve(x0)
        outs.append({k: v.clone() for k, v in o.items()})
    for k in ("x", "u", "K", "cost", "iters"):
        assert torch.equal(outs[0][k], outs[1][k]), ("hybrid", k)


@pytest.mark.parametrize("model,N,B", [("cartpole", 30, 257), ("quadrotor", 50, 300)])
def test_fused_iterate_equals_the_three_separate_calls(model, N, B):
    """quattro_ilqr_iterate_f32 (one host call, caller's workspace) == the separate calls on the same inputs, bit for bit,
    over several iterations: quattro_linearize_sweep_f32 + quattro_linesearch_f32 (both models fuse the linearisation
    into the sweep), and for the quadrotor also quattro_linearize_f32 + quattro_riccati_sweep_f32 + quattro_linesearch_f32
    through records (bit-identical there; the cart-pole's record path agrees to round-off, tests/test_kernels_gpu.py)."""
    q = _pkg()
    ops = q.ops
    md = q.model_by_name(model)
    rng = np.random.default_rng(11)
    x0 = torch.as_tensor(np.asarray(md.x_ref) + 0.1 * rng.standard_normal((B, md.n)), dtype=torch.float32, device=DEV)
    u0 = torch.as_tensor(0.05 * rng.standard_normal((B, N, md.m)), dtype=torch.float32, device=DEV)
    layout = ops.preferred_layout(md.n, md.m)
    state = []
    modes = ("separate-fused-sweep", "iterate") + (("separate-records",) if model == "quadrotor" else ())
    for mode in modes:
        fused = mode == "iterate"
        u = u0.clone()
        x, cost = ops.simulate(md, x0, u)
        K = torch.zeros((B, N, md.m, md.n), dtype=torch.float32, device=DEV)
        k = torch.zeros((B, N, md.m), dtype=torch.float32, device=DEV)
        active = torch.ones(B, dtype=torch.int32, device=DEV)
        iters = torch.zeros(B, dtype=torch.int32, device=DEV)
        status = torch.zeros(B, dtype=torch.int32, device=DEV)
        aidx = torch.full((B,), -1, dtype=torch.int32, device=DEV)
        ws = ops.workspace(md, B, N, DEV) if fused else None
        for _ in range(4):
            if fused:
                ops.ilqr_iterate(md, x, u, K, k, cost, 1e-3, ws, alpha_idx=aidx, active=active, iters=iters, status=status)
            elif mode == "separate-fused-sweep":
                ops.linearize_sweep(md, x, u, 0, K=K, k=k, status=status, active=active)
                ops.linesearch(md, x, u, K, k, cost, 1e-3, alpha_idx=aidx, active=active, iters=iters)
            else:
                rec, VxN, VxxN, _ = ops.linearize(md, x, u, layout=layout)
                ops.riccati_sweep(rec, VxN, VxxN, md.n, md.m, layout, K=K, k=k, status=status, active=active)
                ops.linesearch(md, x, u, K, k, cost, 1e-3, alpha_idx=aidx, active=active, iters=iters)
        state.append((x, u, K, k, cost, active, iters, aidx, status))
    for other in state[1:]:
        for a, b in zip(state[0], other):
            assert torch.equal(a, b)
    assert int(state[0][6].max()) >= 2          # the loop really iterated


def test_cartpole_blending_mode_follows_the_reference():
    """CartPoleMPC(ilqr_tf_blend=True) (transformer None: pure iLQR blended with the LQR law), one fresh controller per
    state as in the fixture (G11): full LQR below the lower threshold, the weighted mix in between, the iLQR control
    above; and a controller built with NO flag takes the same branch, like the reference."""
    q = _pkg()
    from conftest import load_golden
    g = load_golden("lqr_cartpole.npz")
    for flags in (dict(ilqr_tf_blend=True), dict()):
        for x, u_ref, w_ref, xseq_ref in zip(g["xb"], g["ub"], g["wb"], g["xseq_b"]):
            mpc = q.CartPoleMPC(horizon=30, dt=0.01, integration_method="euler", device=DEV, **flags)
            xseq, u = mpc.control_step(x)
            assert abs(mpc.switcher.get_blending_weight(0.01) - w_ref) < 1e-12
            # u_primary comes from the fp32 device solver, the reference's from fp64 finite differences
            assert np.allclose(np.atleast_1d(u), u_ref, rtol=2e-3, atol=2e-3), (x, u, u_ref)
            if w_ref <= 0.05:
                assert xseq == []
            else:
                assert np.allclose(np.asarray(xseq), xseq_ref, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("model", ["cartpole", "quadrotor"])
def test_dropin_point_derivative_methods(model):
    """iLQR_TF._compute_dynamics_jacobians / _compute_cost_derivatives / _finite_diff_*_final of the drop-in against the
    reference's finite-difference outputs stored with the sweep fixtures (first derivatives to 1e-5 relative, second
    derivatives to the FD noise of the reference)."""
    q = _pkg()
    from conftest import load_golden
    g = load_golden(f"sweep_{model}_N30.npz")
    mpc = (q.QuadrotorMPC if model == "quadrotor" else q.CartPoleMPC)(horizon=30, dt=0.01, integration_method="euler", device=DEV)
    il = mpc.ilqr
    x, u = g["x_seq"][0][4], g["u_seq"][0][4]
    A, B = il._compute_dynamics_jacobians(x, u)
    assert np.allclose(A, g["A"][0][4], rtol=1e-5, atol=2e-6) and np.allclose(B, g["B"][0][4], rtol=1e-5, atol=2e-6)
    L, Lx, Lu, Lxx, Luu, Lxu = il._compute_cost_derivatives(x, u)
    assert Lxu.shape == (A.shape[0], B.shape[1])
    assert np.allclose(Lx, g["lx"][0][4], rtol=1e-5, atol=1e-5) and np.allclose(Lu, g["lu"][0][4], rtol=1e-4, atol=1e-5)
    assert np.allclose(Lxx, g["lxx"][0][4], atol=1e-5 * (1 + np.abs(g["lxx"][0][4]).max()))
    assert np.allclose(Luu, g["luu"][0][4], atol=1e-5 * (1 + np.abs(g["luu"][0][4]).max()))
    assert np.allclose(Lxu, g["lux"][0][4].T, atol=1e-5)
    assert abs(L - mpc.running_cost(x, u)) <= 1e-6 * abs(L)
    xN = g["x_seq"][0][-1]
    assert np.allclose(il._finite_diff_gradient_final(xN), g["VxN"][0], rtol=1e-5, atol=1e-5)
    assert np.allclose(il._finite_diff_hessian_final(xN), g["VxxN"][0], atol=1e-5 * (1 + np.abs(g["VxxN"][0]).max()))


def test_full_size_batch_against_oracle_samples_and_permutation():
    """BASELINE configs[2] size (B = 4096, N = 50): (i) the first iteration's gains and the accepted step of a few sampled
    trajectories equal the fp64 oracle's on the same inputs; (ii) permuting the batch permutes every output bit for bit
    (no cross-trajectory coupling, no dependence on the position inside a wave / workgroup)."""
    q = _pkg()
    from oracle import ilqr as o_ilqr, linearize as o_lin, models as o_models
    md = q.quadrotor_model()
    spec = o_models.quadrotor_spec(0.01, 0)
    N, B = 50, 4096
    rng = np.random.default_rng(77)
    x0 = np.asarray(md.x_ref) + rng.uniform(-1, 1, (B, 12)) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    u0 = 2.4525 + 0.1 * rng.standard_normal((B, N, 4))
    s = q.QuattroILQR(md, N, max_iter=1, device=DEV)
    out = {k: v.clone() for k, v in s.solve(x0, u0).items()}
    x32, u32 = x0.astype(np.float32).astype(np.float64), u0.astype(np.float32).astype(np.float64)
    for b in (0, 1234, 4095):
        xs, _ = o_lin.rollout_batched(spec, x32[b:b + 1], u32[b:b + 1])
        blocks = o_lin.linearize_analytic(spec, xs, u32[b:b + 1])
        kr, Kr = o_ilqr.riccati_sweep_batched(blocks)
        assert rel_fro(out["K"][b].double().cpu().numpy(), Kr[0]) < 5e-6, b
        assert rel_fro(out["k"][b].double().cpu().numpy(), kr[0]) < 5e-6, b
        J0 = o_lin.rollout_batched(spec, x32[b:b + 1], u32[b:b + 1])[1][0]
        want = -1.0
        for a in q.ops.ALPHAS:
            _, _, Jc = o_lin.closed_loop_rollout_batched(spec, x32[b:b + 1], xs, u32[b:b + 1], kr, Kr, a)
            if Jc[0] <= J0:
                want = a
                break
        assert abs(float(out["alpha"][b]) - want) < 1e-7, (b, float(out["alpha"][b]), want)
    perm = rng.permutation(B)
    s2 = q.QuattroILQR(md, N, max_iter=1, device=DEV)
    out2 = s2.solve(x0[perm], u0[perm])
    pt = torch.as_tensor(perm, device=DEV)
    for key in ("K", "k", "x", "u", "cost", "iters", "alpha", "status"):
        assert torch.equal(out2[key], out[key][pt]), key


def test_full_size_runs_are_bitwise_reproducible():
    """Race screen at BASELINE configs[2] / configs[4] size (B = 4096, N = 50): every kernel of the pure and of the hybrid
    iteration, run five times on the same inputs by fresh solvers, gives bit-identical gains, trajectories, costs and
    decisions (one wave / one workgroup per trajectory, LDS hand-offs ordered by wave fences and barriers, no atomics on
    the data path: there is nothing that may legitimately differ from run to run)."""
    q = _pkg()
    md = q.quadrotor_model()
    N, B = 50, 4096
    rng = np.random.default_rng(5)
    x0 = np.asarray(md.x_ref) + rng.uniform(-1, 1, (B, 12)) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    u0 = 2.4525 + 0.1 * rng.standard_normal((B, N, 4))
    off = np.zeros(12)
    off[2] = 0.5
    for hybrid in (False, True):
        tf = (q.TransformerILQR.random_init(12, 52, prompt_len=1, target_len=N - 1, d_model=128, nhead=4,
                                            num_decoder_layers=3, dim_feedforward=512, max_seq_len=110, device=DEV)
              if hybrid else None)
        ref = None
        for rep in range(5):
            s = q.QuattroILQR(md, N, max_iter=3, device=DEV, tf=tf, state_offset=off if hybrid else None)
            out = {k: v.clone() for k, v in s.solve(x0, u0).items()}
            if ref is None:
                ref = out
                continue
            for key in ("K", "k", "x", "u", "cost", "iters", "alpha", "status"):
                assert torch.equal(out[key], ref[key]), (hybrid, rep, key)


def test_hybrid_graph_fresh_predictor_and_changed_reference():
    """(a) The FIRST hybrid solve of a freshly loaded predictor runs with use_graph=True (the token-bias upload and the C
    struct must be built before the capture starts: host-to-device copies are illegal inside one).  (b) A second solve
    with another x_ref and another state offset replays the same graph: the shifted normalisation mean lives in a
    fixed-address buffer whose CONTENTS are rewritten per solve.  Both equal an eager solver on a second fresh predictor."""
    q = _pkg()
    import os
    from conftest import GOLDEN
    md = q.quadrotor_model()
    rng = np.random.default_rng(11)
    x0 = np.asarray(md.x_ref) + 0.05 * rng.standard_normal((48, 12))
    off = np.eye(12)[2] * 0.5
    xr2 = np.asarray(md.x_ref, dtype=np.float64).copy()
    xr2[0] += 0.3
    xr2[2] += 0.1
    tf_g = q.TransformerILQR(12, 52, device=DEV).load(os.path.join(GOLDEN, "tf_weights_quadrotor.npz"))
    tf_e = q.TransformerILQR(12, 52, device=DEV).load(os.path.join(GOLDEN, "tf_weights_quadrotor.npz"))
    sg = q.QuattroILQR(md, 50, max_iter=3, tf=tf_g, device=DEV, use_graph=True, state_offset=off)
    se = q.QuattroILQR(md, 50, max_iter=3, tf=tf_e, device=DEV, use_graph=False, state_offset=off)
    keys = ("x", "u", "K", "k", "cost", "iters")
    og = {k: v.clone() for k, v in sg.solve(x0).items()}                 # graph captured on a cold predictor
    oe = {k: v.clone() for k, v in se.solve(x0).items()}
    for k in keys:
        assert torch.equal(og[k], oe[k]), ("first solve", k)
    graph_before = sg._graph
    og2 = {k: v.clone() for k, v in sg.solve(x0, x_ref=xr2).items()}
    oe2 = {k: v.clone() for k, v in se.solve(x0, x_ref=xr2).items()}
    assert sg._graph is graph_before                                     # replayed, not re-captured
    for k in keys:
        assert torch.equal(og2[k], oe2[k]), ("changed x_ref", k)
    assert not torch.equal(og2["K"], og["K"])                            # the reference shift really reached the kernel
    # a solver of another shape in between must not disturb the captured graph's scratch (each solver owns its own)
    other = q.QuattroILQR(q.cartpole_model(), 30, max_iter=2, device=DEV)
    other.solve(np.zeros((333, 4)) + 0.1)
    q.ops.linesearch_scratch(md, 7, 50, DEV)
    og3 = {k: v.clone() for k, v in sg.solve(x0).items()}
    for k in keys:
        assert torch.equal(og3[k], og[k]), ("after foreign solves", k)


def test_hybrid_full_size_batch_properties_and_dropin_samples():
    """BASELINE configs[4] at its size (B = 4096, N = 50, shipped quadrotor checkpoint, transformer-predicted gains for
    t < 49 + swept tail step): cost never increases, replicated inputs give identical outputs, permuting the batch
    permutes the outputs bit for bit, x is the rollout of the returned u, and three sampled trajectories equal the
    single-trajectory iLQR_TF drop-in (which follows the reference's hybrid control flow step by step)."""
    q = _pkg()
    import os
    from conftest import GOLDEN
    md = q.quadrotor_model()
    N, B = 50, 4096
    off = np.eye(12)[2] * 0.5
    rng = np.random.default_rng(2024)
    x0 = np.asarray(md.x_ref) + rng.uniform(-1, 1, (B, 12)) * np.array([0.3, 0.3, 0.01, 0, 0, 0, 0.1, 0.1, 0.2, 0, 0, 0])
    x0[1] = x0[0]                                                        # replicated input
    tf = q.TransformerILQR(12, 52, device=DEV).load(os.path.join(GOLDEN, "tf_weights_quadrotor.npz"))
    s = q.QuattroILQR(md, N, max_iter=4, tf=tf, device=DEV, state_offset=off)
    x0_t = torch.as_tensor(x0, dtype=torch.float32, device=DEV)
    J0 = q.ops.simulate(md, x0_t, torch.zeros((B, N, 4), dtype=torch.float32, device=DEV))[1].clone()
    out = {k: v.clone() for k, v in s.solve(x0).items()}
    assert int((out["status"] != 0).sum()) == 0
    assert bool((out["cost"] <= J0).all())
    assert int(out["iters"].min()) >= 1 and int(out["iters"].max()) <= 4
    for k in ("x", "u", "K", "k", "cost", "iters"):
        assert torch.equal(out[k][0], out[k][1]), k
    xs, Js = q.ops.simulate(md, x0_t, out["u"].contiguous())
    assert torch.equal(xs, out["x"])
    assert torch.equal(Js, out["cost"])
    perm = rng.permutation(B)
    out2 = q.QuattroILQR(md, N, max_iter=4, tf=tf, device=DEV, state_offset=off).solve(x0[perm])
    pt = torch.as_tensor(perm, device=DEV)
    for k in ("x", "u", "K", "k", "cost", "iters", "alpha", "status"):
        assert torch.equal(out2[k], out[k][pt]), k
    for b in (0, 1777, 4095):
        il = q.iLQR_TF(None, None, None, x0[b].astype(np.float32).astype(np.float64), [np.zeros(4) for _ in range(N)], N,
                       max_iter=4, tf=tf, model=md, device=DEV)
        il.set_state_offset(off)
        u_seq, x_seq = il.optimize(np.asarray(md.x_ref, dtype=np.float64))
        # round 4: the drop-in runs the SAME device path as the batched solver (tail sweep in place, predictor with the
        # shifted mean, fused line search — as a captured graph at B = 1), so a trajectory's solve does not depend on the
        # batch it is in: bit for bit (rounds 2-3 fed the B = 1 kernel a host-formed x_err and compared at 5e-3)
        assert len(il.logs) == int(out["iters"][b]), b
        assert np.array_equal(np.asarray(u_seq), out["u"][b].double().cpu().numpy()), b
        assert np.array_equal(x_seq, out["x"][b].double().cpu().numpy()), b
        assert np.array_equal(np.array(il.logs[-1]["K_seq_seg"]), out["K"][b, N - 1:].double().cpu().numpy())


def test_rccl_world1_all_gather_in_a_child_process():
    """init_process_group("nccl") — RCCL on ROCm — in a FRESH process (RANK=0, WORLD_SIZE=1, 127.0.0.1), then the gain
    gather of parallel.all_gather_gains on device tensors through both code paths (equal shards: one
    all_gather_into_tensor of the packed [k | K]; generic: size exchange + padded gather).  World size 1 cannot show
    xGMI traffic, but it runs the RCCL communicator bootstrap, the collective launch and the stream ordering the 8-GPU
    bench depends on, on real hardware."""
    import os
    import subprocess
    import sys
    from conftest import PKG_DIR, ROOT
    code = r'''
import os, sys
sys.path[:0] = [%r, %r]
import torch, torch.distributed as dist
from quattro_ilqr_amd import parallel
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
g = torch.Generator(device="cpu").manual_seed(3)
K = torch.randn((257, 50, 4, 12), generator=g).to(dev)
k = torch.randn((257, 50, 4), generator=g).to(dev)
# world == 1 short-circuits inside all_gather_gains; call the collective the way world > 1 does
mine = parallel.pack_gains(K, k)
out = torch.empty_like(mine)
dist.all_gather_into_tensor(out, mine)
torch.cuda.synchronize()
K2, k2 = parallel.unpack_gains(out)
assert torch.equal(K2, K) and torch.equal(k2, k)
sizes = [torch.zeros(1, dtype=torch.int64, device=dev)]
dist.all_gather(sizes, torch.tensor([mine.shape[0]], dtype=torch.int64, device=dev))
assert int(sizes[0].item()) == 257
Ka, ka = parallel.all_gather_gains(K, k, equal_shards=True)
Kb, kb = parallel.all_gather_gains(K, k, equal_shards=False)
assert torch.equal(Ka, K) and torch.equal(kb, k)
# the benchmark's exchange: ONE collective on the flat [K | k] buffer the solver allocates, no repacking; at world size 1
# the class copies locally, so issue the collective itself on the same buffers as well
flat = torch.cat([K.reshape(-1), k.reshape(-1)])
gg = parallel.GainGather(257, 50, 4, 12, torch.float32, dev)
Kg, kg = gg(flat)
assert torch.equal(Kg[0], K) and torch.equal(kg[0], k)
gg.recv.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
dist.all_gather_into_tensor(gg.recv.view(-1), flat)
e1.record()
torch.cuda.synchronize()
assert e0.elapsed_time(e1) > 0.0
Kg, kg = gg.views()
assert torch.equal(Kg[0], K) and torch.equal(kg[0], k)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == 1.5
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
''' % (ROOT, PKG_DIR)
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_config3_global_batch_on_one_gpu():
    """BASELINE configs[3]'s GLOBAL batch (B = 32 768, N = 50) on one GPU — the 8-GPU run shards exactly this problem
    4096 per rank with no data-path collective, so its results must equal this single-GPU solve shard by shard:
    (i) three iterations never increase a trajectory's cost and flag nothing; (ii) the gains of the first iteration of
    sampled trajectories equal the fp64 oracle's; (iii) every 4096-trajectory shard solved on its own (what a rank of the
    sharded run does) reproduces its slice of the big batch bit for bit; (iv) K and k are views of ONE flat [K | k]
    buffer (what parallel.GainGather sends)."""
    q = _pkg()
    from oracle import ilqr as o_ilqr, linearize as o_lin, models as o_models
    from quattro_ilqr_amd.parallel import shard_bounds
    md = q.quadrotor_model()
    spec = o_models.quadrotor_spec(0.01, 0)
    N, B, W = 50, 32768, 8
    rng = np.random.default_rng(1234)
    x0 = np.asarray(md.x_ref) + rng.uniform(-1, 1, (B, 12)) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    u0 = (2.4525 + 0.1 * rng.standard_normal((B, N, 4))).astype(np.float32)
    x0 = x0.astype(np.float32)
    s = q.QuattroILQR(md, N, device=DEV)
    first = {k: v.clone() for k, v in s.solve(x0, u0, max_iter=1).items()}
    assert s.K.data_ptr() == s.gains_flat.data_ptr() and s.k.data_ptr() == s.gains_flat.data_ptr() + 4 * s.K.numel()
    assert int(first["status"].abs().sum()) == 0
    x64, u64 = x0.astype(np.float64), u0.astype(np.float64)
    for b in (0, 4095, 4096, 20000, B - 1):
        xs, _ = o_lin.rollout_batched(spec, x64[b:b + 1], u64[b:b + 1])
        kr, Kr = o_ilqr.riccati_sweep_batched(o_lin.linearize_analytic(spec, xs, u64[b:b + 1]))
        assert rel_fro(first["K"][b].double().cpu().numpy(), Kr[0]) < 5e-6, b
        assert rel_fro(first["k"][b].double().cpu().numpy(), kr[0]) < 5e-6, b
    three = {k: v.clone() for k, v in s.solve(x0, u0, max_iter=3).items()}
    J0 = q.ops.simulate(md, torch.as_tensor(x0, device=DEV), torch.as_tensor(u0, device=DEV))[1]
    assert bool((first["cost"] <= J0).all()) and bool((three["cost"] <= first["cost"]).all())
    assert int(three["status"].abs().sum()) == 0 and int(three["iters"].max()) <= 3
    shard = q.QuattroILQR(md, N, device=DEV)
    for r in (0, 3, W - 1):
        lo, hi = shard_bounds(B, r, W)
        assert hi - lo == 4096
        out = shard.solve(x0[lo:hi], u0[lo:hi], max_iter=3)
        for key in ("K", "k", "x", "u", "cost", "iters", "alpha", "status"):
            assert torch.equal(out[key], three[key][lo:hi]), (r, key)


def test_bench_self_launch_two_ranks_share_the_gpu():
    """`python bench.py --gpus 2` with no launcher around it on real kernels: the parent starts two ranks,
    which share this box's one GPU over gloo (QT_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device), run the
    headline workload at a reduced batch, gather [K | k] with parallel.GainGather inside the timed region, and rank 0
    prints ONE JSON line."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(QT_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--batch", "512", "--clock-settle-ms", "5", "--no-extras"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 1024 and out["gather_ms"] > 0 and len(out["ms_per_step_per_rank"]) == 2
    assert out["comm"]["gather_bytes_received_per_rank"] == 512 * 50 * 4 * 13 * 4
    assert out["flagged_trajectories"] == 0 and out["accepted_fraction"] > 0.99
    assert "cpu_baseline" not in out and out["value_no_settle"] > 0


# ------------------------------------------------------------------------------------------------ device-resident loops
@pytest.mark.parametrize("B", [1, 2, 301, 4096])
def test_device_resident_solve_equals_host_driven_loop(B):
    """quattro_ilqr_solve_f32 (ONE persistent launch: rollout, every iteration, every trajectory's own stop test) against the
    host-driven loop (quattro_ilqr_iterate_f32 once per iteration + convergence checks), bit for bit: odd and tiny batches
    (the second wave of the last workgroup has no trajectory), real exit tests with early-stopping trajectories inside a
    workgroup whose other trajectory goes on, capped max_iter, and fixed-iteration mode."""
    q = _pkg()
    md = q.quadrotor_model()
    assert q.ops.model_has_device_loop(md)
    N = 50
    rng = np.random.default_rng(100 + B)
    x0 = np.asarray(md.x_ref) + rng.uniform(-1, 1, (B, 12)) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    u0 = 2.4525 + 0.1 * rng.standard_normal((B, N, 4))
    keys = ("K", "k", "x", "u", "cost", "iters", "alpha", "status")
    for kw in (dict(), dict(max_iter=3), dict(max_iter=5, fixed_iters=True)):
        for u_init in (u0, None):                       # warm nominal / the reference's cold start (zeros)
            dev = q.QuattroILQR(md, N, max_iter=40, device=DEV, device_loop=True)
            host = q.QuattroILQR(md, N, max_iter=40, device=DEV, device_loop=False, check_every=1)
            od = {k: v.clone() for k, v in dev.solve(x0, u_init, **kw).items()}
            oh = host.solve(x0, u_init, **kw)
            for key in keys:
                assert torch.equal(od[key], oh[key]), (B, kw, u_init is None, key)
            assert torch.equal(dev.active, host.active) and torch.equal(dev.alpha_idx, host.alpha_idx)
            if not kw:
                assert int(od["iters"].max()) > int(od["iters"].min()) or B == 1      # trajectories really stop at different times
            assert int(od["status"].abs().sum()) == 0


@pytest.mark.parametrize("B,steps", [(3, 4), (257, 3), (4096, 4)])
def test_device_resident_mpc_loop_equals_host_driven_loop(B, steps):
    """quattro_mpc_run_f32 (all control steps of all controllers in ONE launch: solve, plant step, disturbance, warm-start
    shift) against BatchedMPC's host-driven loop, bit for bit, including a second run() that continues from the first
    one's warm start."""
    q = _pkg()
    md = q.quadrotor_model()
    N = 50
    rng = np.random.default_rng(7 + B)
    x0 = (np.asarray(md.x_ref) + rng.uniform(-1, 1, (B, 12)) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])).astype(np.float32)
    dist = torch.as_tensor(1e-3 * rng.standard_normal((steps, B, 12)), dtype=torch.float32, device=DEV)
    a = q.BatchedMPC(md, N, max_iter=6, tol=1e-3, device=DEV, check_every=1)
    b = q.BatchedMPC(md, N, max_iter=6, tol=1e-3, device=DEV, check_every=1)
    for rep, d in enumerate((dist, None)):
        start = x0 if rep == 0 else oa["x"][:, -1].clone()
        oa = a.run(start, steps, disturbance=d, device_loop=True)
        ob = b.run(start, steps, disturbance=d, device_loop=False)
        for key in ("x", "u", "iters"):
            assert torch.equal(oa[key], ob[key].to(oa[key].dtype)), (rep, key)
        assert torch.equal(a.u_warm, b.u_warm)
        # state of the LAST solve (gains, nominal states, cost): a wave that kept sweeping after its partner had moved on to
        # the next control step — the race the barrier in front of mpc_advance closes — overwrites exactly these
        for name in ("K", "k", "x", "cost", "alpha_idx", "status"):
            assert torch.equal(getattr(a.solver, name), getattr(b.solver, name)), (rep, name)
        assert int(oa["iters"].min()) >= 1 and int(oa["iters"].max()) <= 6
        assert torch.equal(oa["x"][:, 0], torch.as_tensor(start, device=DEV))


def _cartpole_batch(B, N, seed):
    rng = np.random.default_rng(seed)
    x0 = np.zeros((B, 4))
    x0[:, 0] = rng.uniform(-0.5, 0.5, B)
    x0[:, 2] = rng.uniform(-0.5, 0.5, B)
    return x0.astype(np.float32), (0.3 * rng.standard_normal((B, N, 1))).astype(np.float32)


@pytest.mark.parametrize("integ,B,N", [("euler", 1, 30), ("euler", 1024, 50), ("euler", 77, 67), ("rk4", 5, 30), ("rk4", 259, 50)])
def test_cartpole_device_resident_solve_equals_host_driven_loop(integ, B, N):
    """The cart-pole's persistent kernel (csrc/solve_cartpole.hip: a 16-lane row per trajectory, wave-private loop) against
    the host-driven loop, bit for bit: BASELINE configs[1] size, ragged batches (rows of the last wave without a trajectory),
    horizons longer than one LDS record stage, both integrators, real exit tests / capped / fixed iteration counts."""
    q = _pkg()
    md = q.cartpole_model(dt=0.01, integrator=integ)
    assert q.ops.model_has_device_loop(md)
    x0, u0 = _cartpole_batch(B, N, 11 * B + N)
    keys = ("K", "k", "x", "u", "cost", "iters", "alpha", "status")
    for kw in (dict(), dict(max_iter=2), dict(max_iter=4, fixed_iters=True)):
        for u_init in (u0, None):
            dev = q.QuattroILQR(md, N, max_iter=30, tol=1e-3, device=DEV, device_loop=True)
            host = q.QuattroILQR(md, N, max_iter=30, tol=1e-3, device=DEV, device_loop=False, check_every=1)
            od = {k: v.clone() for k, v in dev.solve(x0, u_init, **kw).items()}
            oh = host.solve(x0, u_init, **kw)
            for key in keys:
                assert torch.equal(od[key], oh[key]), (integ, B, kw, u_init is None, key)
            assert torch.equal(dev.active, host.active) and torch.equal(dev.alpha_idx, host.alpha_idx)
            assert int(od["status"].abs().sum()) == 0


@pytest.mark.parametrize("integ,B,steps", [("euler", 6, 4), ("rk4", 130, 3), ("euler", 1024, 3)])
def test_cartpole_device_resident_mpc_loop_equals_host_driven_loop(integ, B, steps):
    q = _pkg()
    md = q.cartpole_model(dt=0.01, integrator=integ)
    N = 30
    x0, _ = _cartpole_batch(B, N, 5 * B)
    rng = np.random.default_rng(B)
    dist = torch.as_tensor(1e-3 * rng.standard_normal((steps, B, 4)), dtype=torch.float32, device=DEV)
    a = q.BatchedMPC(md, N, max_iter=8, tol=1e-2, device=DEV, check_every=1)
    b = q.BatchedMPC(md, N, max_iter=8, tol=1e-2, device=DEV, check_every=1)
    for rep, d in enumerate((dist, None)):
        start = x0 if rep == 0 else oa["x"][:, -1].clone()
        oa = a.run(start, steps, disturbance=d, device_loop=True)
        ob = b.run(start, steps, disturbance=d, device_loop=False)
        for key in ("x", "u", "iters"):
            assert torch.equal(oa[key], ob[key].to(oa[key].dtype)), (rep, key)
        assert torch.equal(a.u_warm, b.u_warm)
        for name in ("K", "k", "x", "cost", "alpha_idx", "status"):
            assert torch.equal(getattr(a.solver, name), getattr(b.solver, name)), (rep, name)


@pytest.mark.parametrize("integ,N", [("euler", 1), ("rk4", 1), ("euler", 2), ("rk4", 3)])
def test_cartpole_device_resident_mpc_loop_on_the_shortest_horizons(integ, N):
    """N = 1: nothing shifts in the warm start, but the plant still steps (found by scripts/fuzz_device_loops.py: the plant step
    rode on the first pass of the shift loop, which did not run)."""
    q = _pkg()
    md = q.cartpole_model(dt=0.01, integrator=integ)
    x0, _ = _cartpole_batch(9, N, 3 + N)
    a = q.BatchedMPC(md, N, max_iter=5, tol=1e-2, device=DEV, check_every=1, tf_window=0)
    b = q.BatchedMPC(md, N, max_iter=5, tol=1e-2, device=DEV, check_every=1, tf_window=0)
    oa, ob = a.run(x0, 4, device_loop=True), b.run(x0, 4, device_loop=False)
    for key in ("x", "u", "iters"):
        assert torch.equal(oa[key], ob[key].to(oa[key].dtype)), key
    assert float((oa["x"][:, -1] - oa["x"][:, 0]).abs().max()) > 0.0            # the plant moved


@pytest.mark.parametrize("B", [2, 301])
def test_rk4_quadrotor_device_resident_solve_equals_host_driven_loop(B):
    """The RK4 quadrotor (the default integrator of the reference's QuadrotorMPC, quadrotor_mpc.py:12) in the persistent kernel:
    its sweep linearises through the four RK4 stages by forward mode on the matrix pipe (MODE_FUSED_RK4), the host-driven
    loop goes through TILE16R records — two different linearisation codes, so the comparison is to fp32 round-off, not bit
    for bit: same iteration counts and accepted steps, states / controls / gains to 1e-4 after the whole solve."""
    q = _pkg()
    md = q.quadrotor_model(integrator="rk4")
    assert q.ops.model_has_device_loop(md)
    N = 30
    rng = np.random.default_rng(300 + B)
    x0 = np.asarray(md.x_ref) + rng.uniform(-1, 1, (B, 12)) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    u0 = 2.4525 + 0.1 * rng.standard_normal((B, N, 4))
    for kw in (dict(max_iter=1), dict(max_iter=4, fixed_iters=True), dict()):
        dev = q.QuattroILQR(md, N, max_iter=25, device=DEV, device_loop=True)
        host = q.QuattroILQR(md, N, max_iter=25, device=DEV, device_loop=False, check_every=1)
        od = {k: v.clone() for k, v in dev.solve(x0, u0, **kw).items()}
        oh = host.solve(x0, u0, **kw)
        assert int(od["status"].abs().sum()) == 0
        same = od["iters"] == oh["iters"]
        frac = float(same.float().mean())
        print(f"RK4 device loop vs host loop B={B} {kw}: iteration counts equal for {100 * frac:.1f} % of the trajectories")
        assert frac >= 0.99          # measured 99.3-100 % (a near-tie of the accept or stop test may fall the other way)
        if kw.get("max_iter") == 1 or kw.get("fixed_iters"):
            assert torch.equal(od["iters"], oh["iters"])
        same = same & (od["alpha"] == oh["alpha"])               # (a near-tie of the accept test may fall the other way)
        assert float(same.float().mean()) >= 0.985
        sel = same.nonzero().flatten()
        for key, tol in (("x", 1e-4), ("u", 2e-4), ("K", 2e-4), ("k", 5e-4)):
            e = rel_fro(od[key][sel].double().cpu().numpy(), oh[key][sel].double().cpu().numpy())
            assert e < tol, (B, kw, key, e)
        assert rel_fro(od["cost"][sel].cpu().numpy(), oh["cost"][sel].cpu().numpy()) < 1e-5


@pytest.mark.parametrize("N", [1, 2, 7, 26, 51, 77])
def test_device_resident_solve_on_ragged_horizons(N):
    """Horizons that are not a multiple of anything the kernels batch by (the fused sweep refills its LDS stage every 25 steps,
    the RK4 table every 12, the rollouts prefetch 2-4 steps ahead, the line search copies two records per lane pair):
    persistent kernel vs host-driven loop, Euler bit for bit, RK4 to round-off; plus the closed-loop (MPC) form."""
    q = _pkg()
    B = 5
    rng = np.random.default_rng(N)
    for integ in ("euler", "rk4"):
        md = q.quadrotor_model(integrator=integ)
        x0 = np.asarray(md.x_ref) + rng.uniform(-1, 1, (B, 12)) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
        u0 = 2.4525 + 0.1 * rng.standard_normal((B, N, 4))
        for kw in (dict(max_iter=3, fixed_iters=True), dict()):
            dev = q.QuattroILQR(md, N, max_iter=12, device=DEV, device_loop=True, tf_window=0)
            host = q.QuattroILQR(md, N, max_iter=12, device=DEV, device_loop=False, check_every=1, tf_window=0)
            od = {k: v.clone() for k, v in dev.solve(x0, u0, **kw).items()}
            oh = host.solve(x0, u0, **kw)
            assert int(od["status"].abs().sum()) == 0 and bool(torch.isfinite(od["cost"]).all())
            if integ == "euler":
                for key in ("K", "k", "x", "u", "cost", "iters", "alpha"):
                    assert torch.equal(od[key], oh[key]), (N, integ, kw, key)
            elif kw.get("fixed_iters"):
                assert torch.equal(od["iters"], oh["iters"])
                same = (od["alpha"] == oh["alpha"]).nonzero().flatten()
                assert same.numel() >= B - 1
                for key, tol in (("x", 1e-4), ("u", 5e-4), ("K", 5e-4)):
                    assert rel_fro(od[key][same].double().cpu().numpy(), oh[key][same].double().cpu().numpy()) < tol, (N, key)
        if integ == "euler":
            a = q.BatchedMPC(md, N, max_iter=4, device=DEV, check_every=1, tf_window=0)
            b = q.BatchedMPC(md, N, max_iter=4, device=DEV, check_every=1, tf_window=0)
            oa, ob = a.run(x0.astype(np.float32), 3, device_loop=True), b.run(x0.astype(np.float32), 3, device_loop=False)
            for key in ("x", "u", "iters"):
                assert torch.equal(oa[key], ob[key].to(oa[key].dtype)), (N, key)


def test_forward_pass_segment_mirrors_the_reference_method():
    """iLQR_TF.forward_pass_segment (quattro_ilqr_tf.py:402-421; dead code in the reference, part of the class surface):
    start_idx = 0 is the forward pass from x_seq[0]; a proper tail raises IndexError exactly where the reference does
    (its compute_total_cost indexes `horizon` controls of an S-long list)."""
    q = _pkg()
    md = q.cartpole_model(dt=0.01, integrator="euler")
    N = 30
    rng = np.random.default_rng(3)
    il = q.iLQR_TF(None, None, None, np.array([0.1, 0.0, 0.2, 0.0]), [np.zeros(1) for _ in range(N)], N, model=md, device=DEV)
    u_seq = [0.2 * rng.standard_normal(1) for _ in range(N)]
    x_seq = il.simulate(u_seq)
    k_seq, K_seq = il.backward_pass(x_seq, u_seq)
    n_fp = len(il.forward_pass_time)
    xs, us, c = il.forward_pass_segment(x_seq, u_seq, k_seq, K_seq, 0, alpha=0.5)
    xf, uf, cf = il.forward_pass(x_seq, u_seq, k_seq, K_seq, alpha=0.5)
    assert len(il.forward_pass_time) == n_fp + 2                         # timed like forward_pass (measure_time decorator)
    assert np.array_equal(xs, xf) and c == cf and all(np.array_equal(a, b) for a, b in zip(us, uf))
    assert xs.shape == (N + 1, 4) and isinstance(us, list) and len(us) == N
    with pytest.raises(IndexError):
        il.forward_pass_segment(x_seq, u_seq, k_seq[20:], K_seq[20:], 20)
    with pytest.raises(IndexError):
        il.forward_pass_segment(x_seq, u_seq, k_seq[:3], K_seq[:3], 0)    # gain stacks shorter than the segment


@pytest.mark.parametrize("W", [10, 20, 30, 40])
def test_hybrid_windows_of_the_published_ladder(W):
    """The reference's headline figure (figures/quadrotor_result.png) quotes iLQR(W) + TF(N - W) for
    W = 40, 30, 20, 10, 1: the last W steps swept, the first N - W predicted, always L = 101 tokens.  Only W = 1 (the shipped
    checkpoint) has weights; the others run here with random-init predictors of the same architecture (prompt_len = W,
    target_len = N - W): the batched solver against the single-trajectory drop-in (which follows the reference's hybrid
    control flow step by step) — same iteration counts, trajectories to bf16 tolerance — plus the index arithmetic the
    window moves around: W swept rows land at t >= N - W, the prompt has W rows of [k | K.flat]."""
    q = _pkg()
    md = q.quadrotor_model()
    N, B = 50, 64
    off = np.eye(12)[2] * 0.5
    rng = np.random.default_rng(500 + W)
    x0 = (np.asarray(md.x_ref) + rng.uniform(-1, 1, (B, 12)) * np.array([0.3, 0.3, 0.01, 0, 0, 0, 0.1, 0.1, 0.2, 0, 0, 0])).astype(np.float32)
    tf = q.TransformerILQR.random_init(12, 52, prompt_len=W, target_len=N - W, d_model=128, nhead=4, num_decoder_layers=3,
                                       dim_feedforward=512, max_seq_len=110, device=DEV, seed=W)
    assert tf.prompt_len == W and tf.target_len == N - W
    s = q.QuattroILQR(md, N, max_iter=3, tf=tf, device=DEV, state_offset=off)
    assert s.tf_window == W
    u0 = (2.4525 + 0.05 * rng.standard_normal((B, N, 4))).astype(np.float32)
    J0 = q.ops.simulate(md, torch.as_tensor(x0, device=DEV), torch.as_tensor(u0, device=DEV))[1].clone()
    out = {k: v.clone() for k, v in s.solve(x0, u0).items()}
    assert s.t_start == N - W
    assert int((out["status"] != 0).sum()) == 0 and bool((out["cost"] <= J0).all())
    # (the swept tail is written in place at rows t >= N - W of the gain stack, where the predictor reads its W-row prompt:
    #  tests/test_solve_log_gpu.py::test_tail_sweep_written_in_place_equals_the_segment_form, ::test_predictor_reads_its_prompt_...)
    for b in (0, B - 1):
        il = q.iLQR_TF(None, None, None, x0[b].astype(np.float64), [u0[b, t].astype(np.float64) for t in range(N)], N,
                       max_iter=3, tf=tf, model=md, device=DEV)
        il.set_state_offset(off)
        u_seq, x_seq = il.optimize(np.asarray(md.x_ref, dtype=np.float64))
        assert il.tf_window == W and len(il.logs) == int(out["iters"][b]), (W, b)
        assert np.array(il.logs[0]["K_seq_seg"]).shape == (W, 4, 12)
        # (the same device path at B = 1 and B = 64: bit for bit since round 4)
        assert np.array_equal(np.asarray(u_seq), out["u"][b].double().cpu().numpy()), (W, b)
        assert np.array_equal(x_seq, out["x"][b].double().cpu().numpy()), (W, b)


# ------------------------------------------------------------------------------------------------ anchored per-iteration parity
ALPHAS_REF = (1.0, 0.5, 0.25, 0.1, 0.05, 0.01)


def _anchored_linesearch(q, md, N, g, key, n_it, u_first, tol):
    """Every logged iteration of one reference run through quattro_linesearch_f32 on the REFERENCE's own states: nominal
    x_seq[i], nominal controls (the previous iteration's u_after, or the start), its gains K[i], k[i] and its current_cost[i].
    -> worst relative errors of the committed controls / states / cost, and the list of decision mismatches."""
    worst = dict(u=0.0, x=0.0, cost=0.0)
    wrong = []
    for i in range(n_it):
        u_nom = u_first if i == 0 else g[key + "u_after"][i - 1]
        x = torch.as_tensor(g[key + "x_seq"][i][None], dtype=torch.float32, device=DEV).contiguous()
        u = torch.as_tensor(np.asarray(u_nom)[None], dtype=torch.float32, device=DEV).contiguous()
        K = torch.as_tensor(g[key + "K"][i][None], dtype=torch.float32, device=DEV).contiguous()
        k = torch.as_tensor(g[key + "k"][i][None], dtype=torch.float32, device=DEV).contiguous()
        cost = torch.tensor([float(g[key + "current_cost"][i])], dtype=torch.float64, device=DEV)
        active = torch.ones((1,), dtype=torch.int32, device=DEV)
        aidx = q.ops.linesearch(md, x, u, K, k, cost, tol, ALPHAS_REF, active=active)
        a_ref = float(g[key + "alpha"][i])
        got = ALPHAS_REF[int(aidx[0])] if int(aidx[0]) >= 0 else -1.0
        found_ref = bool(g[key + "found"][i])
        if got != (a_ref if found_ref else -1.0):
            wrong.append((key, i, got, a_ref))
            continue
        if not found_ref:
            assert int(active[0]) == 0
            continue
        J_ref = float(g[key + "new_cost"][i])
        worst["cost"] = max(worst["cost"], abs(float(cost[0]) - J_ref) / abs(J_ref))
        worst["u"] = max(worst["u"], rel_fro(u[0].double().cpu().numpy(), g[key + "u_after"][i]))
        x_next = g[key + "x_seq"][i + 1] if i + 1 < n_it else g[key + "x_final"]
        worst["x"] = max(worst["x"], rel_fro(x[0].double().cpu().numpy(), x_next))
        # the stop test of :472 on the reference's numbers
        assert int(active[0]) == (0 if abs(float(g[key + "current_cost"][i]) - J_ref) < tol else 1), (key, i)
    return worst, wrong


@pytest.mark.parametrize("model,N,integ", [("cartpole", 30, "euler"), ("quadrotor", 50, "euler"),
                                           ("cartpole", 30, "rk4"), ("quadrotor", 30, "rk4")])
def test_line_search_on_the_references_logged_states_iteration_by_iteration(model, N, integ):
    """The north star's 1e-5 on REAL solver states: no drift can enter, because every iteration starts from
    the reference's own logged nominal, gains and cost (G6).  Accepted alpha and the stop decision exactly — the late
    iterations' near-ties (|cand - cur| ~ tol) included — committed controls, states and cost within 1e-5."""
    q = _pkg()
    g = load_golden(f"opt_{model}{'_rk4' if integ == 'rk4' else ''}.npz")
    md = q.model_by_name(model, integrator=integ)
    worst, wrong = dict(u=0.0, x=0.0, cost=0.0), []
    for s in range(int(g["n_states"])):
        w, wr = _anchored_linesearch(q, md, N, g, f"s{s}_", int(g[f"s{s}_n_iter"]), np.zeros((N, md.m)), float(g["tol"]))
        worst = {k: max(worst[k], w[k]) for k in worst}
        wrong += wr
    print(f"anchored line search, {model} {integ}: u {worst['u']:.2e}  x {worst['x']:.2e}  cost {worst['cost']:.2e}  "
          f"decisions wrong: {wrong}")
    assert not wrong
    assert worst["u"] <= 1e-5 and worst["x"] <= 1e-5 and worst["cost"] <= 1e-5


@pytest.mark.parametrize("integ", ["euler", "rk4"])
def test_line_search_on_the_references_logged_states_user_model(integ):
    """The same on G13: the reference's iLQR_TF run on a problem it does not ship (40-iteration runs: many late iterations)."""
    q = _pkg()
    from test_user_model_gpu import planar_model
    g = load_golden("user_planar.npz")
    md = planar_model(integ)
    N = int(g["N"])
    worst, wrong = dict(u=0.0, x=0.0, cost=0.0), []
    for s in range(g["x0"].shape[0]):
        key = f"{integ}_s{s}_"
        w, wr = _anchored_linesearch(q, md, N, g, key, int(g[key + "n_iter"]), g["u_init"][s], float(g["tol"]))
        worst = {k: max(worst[k], w[k]) for k in worst}
        wrong += wr
    print(f"anchored line search, planar {integ}: u {worst['u']:.2e}  x {worst['x']:.2e}  cost {worst['cost']:.2e}  wrong: {wrong}")
    assert not wrong
    assert worst["u"] <= 1e-5 and worst["x"] <= 1e-5 and worst["cost"] <= 1e-5
