#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Needs /root/reference (read-only checkout of salemon/quattro-transformer-ilqr).  The reference never
travels to the GPU box, so what this script writes is committed: plain float/int arrays in .npz files
(inputs and the reference's outputs).  No reference source, bytecode or pickled object is stored.

The shipped checkpoints are read with loaders that execute nothing from the files
(`torch.load(weights_only=True)`, `numpy.load` without pickle) and re-exported as plain fp16 arrays.

Fixture families:
  G1/G2  dyn_cost_<model>.npz      f(x,u) Euler+RK4, L(x,u), Lf(x) on random points
  G3/G4  sweep_<model>_N<N>.npz    reference FD derivative blocks (inputs) + backward_pass K,k (outputs)
                                   + backward_pass_segment outputs for three start indices
  G5     fwd_<model>.npz           forward_pass for all six alphas
  G6     opt_<model>[_rk4].npz     per-iteration logs of optimize() (pure iLQR) from several x0 (Euler; RK4 = the MPC classes' default)
  G7     tf_<model>.npz            TransformerILQR.predict in/out (fp16 CPU) + fp32 module outputs
         tf_weights_<model>.npz    the checkpoint as plain arrays (fp16) + normaliser + hparams
  G8     hybrid_<model>.npz        optimize() with the transformer: prompts, predictions, logs (quadrotor P = 1, cart-pole P = 5)
  G9     warm_<model>.npz          two consecutive control_step() calls (warm-start shift)
  G10    dataset_<model>.npz       optimize() logs -> TransformerILQR._create_dataset -> DataNormalizer.fit and the
                                   prompt/target slices of TransformerILQR.fit (the training-set format)

  G11    lqr_cartpole.npz          CartPoleMPC's LQR / blending modes: DARE gain, switcher weights, control_step outputs
  G13    user_planar.npz           the reference's iLQR_TF on a problem it does NOT ship, handed over as plain Python callables
                                   (quattro_ilqr_tf.py:66-84): a planar two-rotor vehicle with a non-diagonal, non-quadratic cost,
                                   Euler and RK4 — optimize() logs for several starts; pins the user-compiled models of
                                   quattro_ilqr_amd.compile_model (tests/test_user_model_gpu.py holds the same problem as C++ bodies)

`--only dataset` regenerates G10 alone, `--only lqr` G11, `--only user_planar` G13.
"""
import os
import sys

import numpy as np

REF = "/root/reference"
sys.path[:0] = [REF, os.path.join(REF, "examples/quadrotor"), os.path.join(REF, "examples/cartpole")]
sys.dont_write_bytecode = True
OUT = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402
from cartpole_mpc import CartPoleMPC  # noqa: E402
from quadrotor_mpc import QuadrotorMPC  # noqa: E402
from quattro_ilqr_tf.transformer_ilqr import TransformerILQR  # noqa: E402
from quattro_ilqr_tf.transformer_model import TransformerPredictor  # noqa: E402

CKPT = {
    "quadrotor": os.path.join(REF, "examples/quadrotor/dec3_dmodel128_nhead4_ff512_drop0.1_epoch200_promptlen1_616.2k"),
    "cartpole": os.path.join(REF, "examples/cartpole/dec3_dmodel128_nhead4_ff256_drop0.1_epoch200_promptlen5_402.7k"),
}


def make_mpc(model, horizon, method, tf=None):
    if model == "quadrotor":
        return QuadrotorMPC(horizon=horizon, dt=0.01, integration_method=method, transformer_model=tf)
    if tf is None:
        return CartPoleMPC(horizon=horizon, dt=0.01, integration_method=method, ilqr_only=True)
    return CartPoleMPC(horizon=horizon, dt=0.01, integration_method=method, transformer_model=tf, ilqr_tf_only=True)


def sample_x0(model, rng):
    """Synthetic initial states, SURVEY §8(d) (the reference's own LHS ranges)."""
    if model == "quadrotor":
        x0 = np.zeros(12)
        x0[2] = 0.5
        x0 += rng.uniform(-1, 1, 12) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
        return x0
    return np.array([rng.uniform(-0.5, 0.5), 0.0, rng.uniform(-0.5, 0.5), 0.0])


def sample_u(model, N, rng, scale=0.3):
    if model == "quadrotor":
        return [np.full(4, 2.4525) + scale * rng.normal(size=4) for _ in range(N)]
    return [scale * 5 * rng.normal(size=1) for _ in range(N)]


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------ G1/G2
def gen_dyn_cost(model):
    rng = np.random.default_rng(11)
    n, m = (12, 4) if model == "quadrotor" else (4, 1)
    mpc_e = make_mpc(model, 30, "euler")
    mpc_r = make_mpc(model, 30, "rk4")
    P = 64
    X = np.zeros((P, n)); U = np.zeros((P, m))
    fe = np.zeros((P, n)); fr = np.zeros((P, n)); Lv = np.zeros(P); Lfv = np.zeros(P)
    for i in range(P):
        x = mpc_e.x_ref + rng.normal(size=n) * 0.3
        u = rng.normal(size=m) * 2.0 + (2.0 if model == "quadrotor" else 0.0)
        X[i], U[i] = x, u
        fe[i] = mpc_e.discrete_dynamics(x, u)
        fr[i] = mpc_r.discrete_dynamics(x, u)
        Lv[i] = mpc_e.running_cost(x, u)
        Lfv[i] = mpc_e.final_cost(x)
    save(f"dyn_cost_{model}.npz", x=X, u=U, f_euler=fe, f_rk4=fr, L=Lv, Lf=Lfv, dt=0.01)


# ------------------------------------------------------------------ G3/G4
def gen_sweep(model, N, n_traj, method="euler", tag=""):
    rng = np.random.default_rng(100 + N)
    mpc = make_mpc(model, N, method)
    il = mpc.ilqr
    n, m = (12, 4) if model == "quadrotor" else (4, 1)
    keys = ["x_seq", "u_seq", "A", "B", "lx", "lu", "lxx", "luu", "lux", "VxN", "VxxN", "K", "k"]
    acc = {k: [] for k in keys}
    seg_starts = [N - 1, N - 5, N - 10]
    seg = {s: dict(K=[], k=[]) for s in seg_starts}
    for _ in range(n_traj):
        il.x0 = sample_x0(model, rng)
        u_seq = sample_u(model, N, rng)
        xs = il.simulate(u_seq)
        A = np.zeros((N, n, n)); B = np.zeros((N, n, m)); lx = np.zeros((N, n)); lu = np.zeros((N, m))
        lxx = np.zeros((N, n, n)); luu = np.zeros((N, m, m)); lux = np.zeros((N, m, n))
        for t in range(N):
            A[t], B[t] = il._compute_dynamics_jacobians(xs[t], u_seq[t])
            _, lx[t], lu[t], lxx[t], luu[t], lxu = il._compute_cost_derivatives(xs[t], u_seq[t])
            lux[t] = lxu.T
        VxN = il._finite_diff_gradient_final(xs[-1])
        VxxN = il._finite_diff_hessian_final(xs[-1])
        k_seq, K_seq = il.backward_pass(xs, u_seq)
        for kname, val in zip(keys, [xs, np.array(u_seq), A, B, lx, lu, lxx, luu, lux, VxN, VxxN,
                                     np.array(K_seq), np.array(k_seq)]):
            acc[kname].append(val)
        for s in seg_starts:
            ks, Ks = il.backward_pass_segment(xs, u_seq, s)
            seg[s]["K"].append(np.array(Ks)); seg[s]["k"].append(np.array(ks))
    out = {k: np.array(v) for k, v in acc.items()}
    for s in seg_starts:
        out[f"segK_{N - s}"] = np.array(seg[s]["K"])
        out[f"segk_{N - s}"] = np.array(seg[s]["k"])
    out["seg_lengths"] = np.array([N - s for s in seg_starts])
    out["integrator"] = np.array(0 if method == "euler" else 1)
    save(f"sweep_{model}_N{N}{tag}.npz", **out)


# ------------------------------------------------------------------ G5
def gen_forward(model, N):
    rng = np.random.default_rng(7)
    mpc = make_mpc(model, N, "euler")
    il = mpc.ilqr
    alphas = [1.0, 0.5, 0.25, 0.1, 0.05, 0.01]
    X0, XS, US, Ks, ks, NX, NU, NJ, J0 = [], [], [], [], [], [], [], [], []
    for _ in range(3):
        il.x0 = sample_x0(model, rng)
        u_seq = sample_u(model, N, rng)
        xs = il.simulate(u_seq)
        k_seq, K_seq = il.backward_pass(xs, u_seq)
        nx, nu, nj = [], [], []
        for a in alphas:
            cx, cu, cj = il.forward_pass(xs, u_seq, k_seq, K_seq, a)
            nx.append(cx); nu.append(np.array(cu)); nj.append(cj)
        X0.append(il.x0.copy()); XS.append(xs); US.append(np.array(u_seq)); Ks.append(np.array(K_seq)); ks.append(np.array(k_seq))
        NX.append(np.array(nx)); NU.append(np.array(nu)); NJ.append(np.array(nj))
        J0.append(il.compute_total_cost(xs, u_seq))
    save(f"fwd_{model}.npz", x0=np.array(X0), x_seq=np.array(XS), u_seq=np.array(US), K=np.array(Ks), k=np.array(ks),
         alphas=np.array(alphas), new_x=np.array(NX), new_u=np.array(NU), new_cost=np.array(NJ), cost0=np.array(J0))


# ------------------------------------------------------------------ G6 / G9
def _pad_logs(logs, N, n, m, max_it, hybrid=False):
    """Stack per-iteration logs into fixed arrays (n_it valid rows)."""
    it = len(logs)
    o = dict(n_iter=np.array(it),
             x_seq=np.zeros((max_it, N + 1, n)), u_after=np.zeros((max_it, N, m)), current_cost=np.zeros(max_it),
             alpha=np.zeros(max_it), new_cost=np.zeros(max_it), found=np.zeros(max_it, dtype=np.int32))
    if not hybrid:
        o["K"] = np.zeros((max_it, N, m, n)); o["k"] = np.zeros((max_it, N, m))
    for i, lg in enumerate(logs):
        o["x_seq"][i] = lg["x_seq"]; o["u_after"][i] = np.array(lg["u_seq"])
        o["current_cost"][i] = lg["current_cost"]
        o["alpha"][i] = lg["alpha"] if lg["alpha"] is not None else -1.0
        o["new_cost"][i] = lg["new_cost"] if lg["new_cost"] is not None else np.nan
        o["found"][i] = int(lg["found_update"])
        if not hybrid:
            o["K"][i] = np.array(lg["K_seq"]); o["k"][i] = np.array(lg["k_seq"])
    return o


def gen_optimize(model, N, n_states, max_iter, method="euler", tag=""):
    rng = np.random.default_rng(21)
    n, m = (12, 4) if model == "quadrotor" else (4, 1)
    x0s = []
    # the README scenarios first (quadrotor_sim.py:250 roll=0.1; cartpole_sim.py:208 angle=0.1)
    if model == "quadrotor":
        x = np.zeros(12); x[2] = 0.5; x[6] = 0.1
    else:
        x = np.array([0.0, 0.0, 0.1, 0.0])
    x0s.append(x)
    while len(x0s) < n_states:
        x0s.append(sample_x0(model, rng))
    out = {}
    for i, x0 in enumerate(x0s):
        mpc = make_mpc(model, N, method)
        mpc.ilqr.max_iter = max_iter
        mpc.ilqr.x0 = x0
        u_fin, x_fin = mpc.ilqr.optimize(mpc.x_ref)
        lg = _pad_logs(mpc.ilqr.logs, N, n, m, max_iter)
        for k_, v in lg.items():
            out[f"s{i}_{k_}"] = v
        out[f"s{i}_x0"] = x0
        out[f"s{i}_u_final"] = np.array(u_fin)
        out[f"s{i}_x_final"] = x_fin
        print(f"  {model} state {i}: {len(mpc.ilqr.logs)} iterations")
    out["n_states"] = np.array(len(x0s)); out["max_iter"] = np.array(max_iter)
    out["tol"] = np.array(mpc.ilqr.tol)
    save(f"opt_{model}{tag}.npz", **out)


def gen_warm(model, N, max_iter):
    rng = np.random.default_rng(5)
    mpc = make_mpc(model, N, "euler")
    mpc.ilqr.max_iter = max_iter
    x_a = sample_x0(model, rng)
    out = dict(x_a=x_a)
    xs1, u1 = mpc.control_step(x_a)
    if model == "cartpole":            # cartpole control_step returns (x_seq, u0); the full u is ilqr.u pre-shift
        u_after_shift = np.array(mpc.ilqr.u)
        out["u0_step1"] = np.array(u1)
    else:
        out["u_step1"] = np.array(u1)
        u_after_shift = np.array(mpc.ilqr.u)
    out["x_step1"] = np.array(xs1)
    out["u_warm"] = u_after_shift
    x_b = np.array(xs1)[1] + 0.01 * rng.normal(size=len(x_a))   # "plant" moved one step + disturbance
    out["x_b"] = x_b
    n_logs = len(mpc.ilqr.logs)
    xs2, u2 = mpc.control_step(x_b)
    out["x_step2"] = np.array(xs2)
    out["u_warm2"] = np.array(mpc.ilqr.u)
    out["n_iter1"] = np.array(n_logs); out["n_iter2"] = np.array(len(mpc.ilqr.logs) - n_logs)
    out["max_iter"] = np.array(max_iter); out["tol"] = np.array(mpc.ilqr.tol)
    save(f"warm_{model}.npz", **out)


# ------------------------------------------------------------------ transformer
def load_reference_tf(model, quant="float16"):
    """Build the reference wrapper around the shipped checkpoint WITHOUT its load() (which unpickles)."""
    d = np.load(os.path.join(CKPT[model], "tf_model_normalizer.npz"))       # allow_pickle=False
    sd = torch.load(os.path.join(CKPT[model], "tf_model.pt"), map_location="cpu", weights_only=True)
    hp = {k: d[k].item() for k in ["target_len", "prompt_len", "state_dim", "control_dim", "d_model", "nhead",
                                   "num_decoder_layers", "dim_feedforward", "dropout", "max_seq_len"]}
    wrap = TransformerILQR(state_dim=hp["state_dim"], control_dim=hp["control_dim"], prompt_len=hp["prompt_len"],
                           d_model=hp["d_model"], nhead=hp["nhead"], num_decoder_layers=hp["num_decoder_layers"],
                           dim_feedforward=hp["dim_feedforward"], dropout=hp["dropout"], max_seq_len=hp["max_seq_len"],
                           quant_mode=quant)
    wrap.device = torch.device("cpu")
    wrap.target_len = hp["target_len"]
    for k in ["x_mean", "x_std", "u_mean", "u_std"]:
        setattr(wrap.normalizer, k, d[k])
    net = TransformerPredictor(state_dim=hp["state_dim"], control_dim=hp["control_dim"], d_model=hp["d_model"],
                               nhead=hp["nhead"], num_decoder_layers=hp["num_decoder_layers"],
                               dim_feedforward=hp["dim_feedforward"], dropout=hp["dropout"],
                               max_seq_len=hp["max_seq_len"], target_len=hp["target_len"], prompt_len=hp["prompt_len"])
    net.load_state_dict(sd)
    net = net.half() if quant == "float16" else net.float()
    net.eval()
    wrap.model = net
    return wrap, hp, d, sd


def export_weights(model):
    _, hp, d, sd = load_reference_tf(model)
    arrays = {k: v.numpy() for k, v in sd.items()}          # fp16 as shipped
    for k in ["x_mean", "x_std", "u_mean", "u_std"]:
        arrays["norm." + k] = d[k]
    for k, v in hp.items():
        arrays["hp." + k] = np.array(v)
    save(f"tf_weights_{model}.npz", **arrays)


def gen_tf(model, N):
    rng = np.random.default_rng(3)
    wrap16, hp, d, _ = load_reference_tf(model, "float16")
    wrap32, _, _, _ = load_reference_tf(model, "none")
    n, c, P = hp["state_dim"], hp["control_dim"], hp["prompt_len"]
    m = 4 if model == "quadrotor" else 1
    S = 8
    xs = np.zeros((S, N + 1, n)); prompts = np.zeros((S, P, c))
    p16 = np.zeros((S, hp["target_len"], c)); p32 = np.zeros_like(p16)
    mpc = make_mpc(model, N, "euler")
    for i in range(S):
        mpc.ilqr.x0 = sample_x0(model, rng)
        u_seq = sample_u(model, N, rng, scale=0.1)
        x_seq = mpc.ilqr.simulate(u_seq)
        k_seg, K_seg = mpc.ilqr.backward_pass_segment(x_seq, u_seq, N - P)
        k_arr, K_arr = np.array(k_seg), np.array(K_seg)
        prompt = np.concatenate([k_arr, K_arr.reshape(P, m * n)], axis=-1)
        x_err = x_seq - mpc.x_ref + mpc.ilqr.get_state_offset()
        xs[i], prompts[i] = x_err, prompt
        p16[i] = wrap16.predict(x_err, prompt)
        p32[i] = wrap32.predict(x_err, prompt)
    # layer-by-layer hidden states (fp32 module) for sample 0
    with torch.no_grad():
        net = wrap32.model
        xn = torch.tensor(((xs[0] - d["x_mean"]) / d["x_std"]).astype(np.float32))[None]
        un = torch.tensor(((prompts[0] - d["u_mean"]) / d["u_std"]).astype(np.float32))[None]
        full = torch.cat([net.state_embed(xn), net.control_embed(un), net.target_embedding[None]], dim=1)
        full = net.pos_encoder(full)
        L = full.size(1)
        mask = torch.triu(torch.ones(L, L), diagonal=1).bool()
        hidden = [full[0].numpy().copy()]
        h = full
        for layer in net.transformer_decoder.layers:
            h = layer(h, src_mask=mask)
            hidden.append(h[0].numpy().copy())
    save(f"tf_{model}.npz", x_err=xs, prompt=prompts, pred_fp16=p16, pred_fp32=p32, hidden_fp32=np.array(hidden),
         torch_version=np.array(torch.__version__), N=np.array(N))


def gen_hybrid(model="quadrotor", max_iter=4):
    N, n, m = (50, 12, 4) if model == "quadrotor" else (30, 4, 1)
    wrap, hp, _, _ = load_reference_tf(model, "float16")
    captured = []
    inner = wrap.predict

    def spy(x_err, prompt):
        y = inner(x_err, prompt)
        captured.append((np.array(x_err), np.array(prompt), np.array(y)))
        return y
    wrap.predict = spy
    mpc = make_mpc(model, N, "euler", tf=wrap)
    mpc.ilqr.max_iter = max_iter
    if model == "quadrotor":
        x0 = np.zeros(12); x0[2] = 0.5; x0[6] = 0.1
    else:
        x0 = np.array([0.0, 0.0, 0.1, 0.0])                  # cartpole_sim.py:208
    mpc.ilqr.x0 = x0
    u_fin, x_fin = mpc.ilqr.optimize(mpc.x_ref)
    lg = _pad_logs(mpc.ilqr.logs, N, n, m, max_iter, hybrid=True)
    it = len(mpc.ilqr.logs)
    lg["x_err"] = np.array([c[0] for c in captured]); lg["prompt"] = np.array([c[1] for c in captured])
    lg["prediction"] = np.array([c[2] for c in captured])
    lg["K_seg"] = np.array([np.array(l["K_seq_seg"]) for l in mpc.ilqr.logs])
    lg["k_seg"] = np.array([np.array(l["k_seq_seg"]) for l in mpc.ilqr.logs])
    lg["tf_window"] = np.array(mpc.ilqr.tf_window); lg["x0"] = x0
    lg["u_final"] = np.array(u_fin); lg["x_final"] = x_fin; lg["state_offset"] = mpc.ilqr.get_state_offset()
    lg["max_iter"] = np.array(max_iter)
    print(f"  hybrid {model}: {it} iterations")
    save(f"hybrid_{model}.npz", **lg)


# ------------------------------------------------------------------ G10
def gen_dataset(model, N, n_states, max_iter, prompt_len):
    """Reference logs -> DataFrame (transformer_training.py:31-42) -> _create_dataset (transformer_ilqr.py:70-92) ->
    DataNormalizer.fit (transformer_model.py:27-31) and the slices of fit() (transformer_ilqr.py:108-114)."""
    import pandas as pd
    rng = np.random.default_rng(77)
    n, m = (12, 4) if model == "quadrotor" else (4, 1)
    logs, x0s = [], []
    for i in range(n_states):
        x0 = sample_x0(model, rng)
        mpc = make_mpc(model, N, "euler")
        mpc.ilqr.max_iter = max_iter
        mpc.ilqr.x0 = x0
        mpc.ilqr.optimize(mpc.x_ref)
        logs += mpc.ilqr.logs
        x0s.append(x0)
    df = pd.DataFrame([{k_: v for k_, v in e.items()} for e in logs])
    tf = TransformerILQR(state_dim=n, control_dim=m * (1 + n), prompt_len=prompt_len)
    x_data, kK_data = tf._create_dataset(df)
    tf.normalizer.fit(x_data, kK_data)
    T = x_data.shape[1]
    kK_norm = tf.normalizer.transform_u(kK_data)
    out = dict(x0=np.array(x0s), n_entries=np.array(len(logs)), prompt_len=np.array(prompt_len), max_iter=np.array(max_iter),
               log_x_seq=np.array([e["x_seq"] for e in logs]), log_k_seq=np.array([np.array(e["k_seq"]) for e in logs]),
               log_K_seq=np.array([np.array(e["K_seq"]) for e in logs]),
               log_iteration=np.array([e["iteration"] for e in logs]),
               x_data=x_data, kK_data=kK_data, x_mean=tf.normalizer.x_mean, x_std=tf.normalizer.x_std,
               u_mean=tf.normalizer.u_mean, u_std=tf.normalizer.u_std,
               x_norm=tf.normalizer.transform_x(x_data), u_prompt=kK_norm[:, -prompt_len:, :],
               u_target=kK_norm[:, :T - prompt_len, :], target_len=np.array(T - prompt_len))
    print(f"  {model}: {len(logs)} log entries, x_data {x_data.shape} {x_data.dtype}, kK_data {kK_data.shape}")
    save(f"dataset_{model}.npz", **out)


# ------------------------------------------------------------------ G11
def gen_lqr():
    """cartpole_mpc.py: linearized_dynamics :272-285, compute_linear_lqr_control :287-301, ControllerSwitcher :10-116,
    control_step in lqr_only and ilqr_tf_blend modes :303-359 (transformer None: blending of pure iLQR with LQR)."""
    rng = np.random.default_rng(99)
    mpc = CartPoleMPC(horizon=30, dt=0.01, integration_method="euler", lqr_only=True)
    A_d, B_d = mpc.linearized_dynamics(mpc.dt)
    xs = rng.uniform(-1, 1, (6, 4)) * np.array([0.5, 0.5, 0.3, 0.5])
    u_lqr = np.array([mpc.compute_linear_lqr_control(x) for x in xs])
    u_step = np.array([mpc.control_step(x)[1] for x in xs])
    # switcher weights along an error sequence (default thresholds of CartPoleMPC: 0.5 / 1.5)
    errs = np.array([[0.1, 0, 0, 0], [0.4, 0.2, 0.2, 0], [0.8, 0.1, 0.3, 0.2], [1.2, 0.5, 0.1, 0.1], [2.0, 0, 0, 0]])
    sw = CartPoleMPC(horizon=30, dt=0.01, integration_method="euler", ilqr_tf_blend=True).switcher
    w = []
    for e in errs:
        sw.update_error(e)
        w.append(sw.get_blending_weight(0.01))
    # blending control steps: one fresh controller per state so that warm starts do not couple them
    xb = np.array([[0.2, 0.0, 0.1, 0.0], [0.9, 0.0, 0.2, 0.0], [1.0, 0.3, -0.3, 0.2], [1.8, 0.0, 0.3, 0.0]])
    ub, wb, x1 = [], [], []
    for x in xb:
        m = CartPoleMPC(horizon=30, dt=0.01, integration_method="euler", ilqr_tf_blend=True)
        xseq, u = m.control_step(x)
        ub.append(np.atleast_1d(u)); wb.append(m.switcher.get_blending_weight(0.01))
        x1.append(np.zeros((31, 4)) if len(xseq) == 0 else xseq)
    save("lqr_cartpole.npz", A_d=A_d, B_d=B_d, xs=xs, u_lqr=u_lqr, u_step=u_step, errs=errs, w=np.array(w),
         xb=xb, ub=np.array(ub), wb=np.array(wb), xseq_b=np.array(x1), Q_lqr=mpc.Q_lqr, R_lqr=mpc.R_lqr)


def gen_int8(model):
    """G12: what the reference's save(quant_mode="int8") stores (transformer_ilqr.py:225-226 applies
    torch.quantization.quantize_dynamic(model, {nn.Linear}, dtype=torch.qint8) and saves THAT state dict) and what its
    int8 model predicts on the G7 inputs — the shipped checkpoint's weights in fp32, the reference module class, the
    reference's own quantisation call.  Stored: the state-dict key list, per quantised layer the scale / zero point and
    a CRC of the int8 weights, and the predictions (the reference's int8 path also quantises activations per call)."""
    import zlib
    import torch.nn as nn
    wrap, hp, d, sd = load_reference_tf(model, quant="none")
    qnet = torch.quantization.quantize_dynamic(wrap.model, {nn.Linear}, dtype=torch.qint8)
    qsd = qnet.state_dict()
    keys = np.array(sorted(qsd.keys()), dtype="S")
    layers, scales, zps, crcs = [], [], [], []
    for k in sorted(qsd.keys()):
        if k.endswith("._packed_params._packed_params"):
            qw, _ = qsd[k]
            layers.append(k[: -len("._packed_params._packed_params")])
            scales.append(qw.q_scale()); zps.append(qw.q_zero_point())
            crcs.append(zlib.crc32(qw.int_repr().numpy().tobytes()))
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"tf_{model}.npz"))
    wrap.model, wrap.quant_mode = qnet, "int8"
    try:      # with torch 2.10 the reference's int8 model cannot run: nn.TransformerEncoderLayer.forward's fast-path check
        pred = np.array([wrap.predict(g["x_err"][i], g["prompt"][i]) for i in range(g["x_err"].shape[0])])   # reads
        err = ""                                                  # linear1.weight, a METHOD of a dynamic-quantised Linear
    except Exception as e:                                        # -> AttributeError; recorded instead of predictions
        pred, err = np.zeros((0,)), f"{type(e).__name__}: {e}"
    save(f"tf_int8_{model}.npz", keys=keys, layers=np.array(layers, dtype="S"), scales=np.array(scales),
         zero_points=np.array(zps), crcs=np.array(crcs, dtype=np.int64), pred_int8=pred,
         predict_error=np.array(err, dtype="S"), torch_version=np.array(torch.__version__, dtype="S"))


def gen_user_planar(n_states=4, N=30, max_iter=40):
    """G13: the reference class itself on callables of our own (no device model, nothing of the examples)."""
    from quattro_ilqr_tf.quattro_ilqr_tf import iLQR_TF
    phys = np.array([1.0, 0.05, 0.2, 9.81])                    # mass, inertia, arm, gravity
    Q = np.array([1.0, 1.0, 1.0, 0.1, 0.1, 0.1]); R = np.array([0.01, 0.02]); QF = np.full(6, 10.0)
    xref = np.array([0.0, 1.0, 0.0, 0.0, 0.0, 0.0])
    dt = 0.02

    def rate(x, u):
        m, inertia, arm, g = phys
        s, c = np.sin(x[2]), np.cos(x[2])
        th = (u[0] + u[1]) / m
        return np.array([x[3], x[4], x[5], -th * s, th * c - g, (u[0] - u[1]) * arm / inertia])

    def make_f(method):
        def f(x, u):
            if method == "euler":
                return x + dt * rate(x, u)
            k1 = rate(x, u); k2 = rate(x + 0.5 * dt * k1, u); k3 = rate(x + 0.5 * dt * k2, u); k4 = rate(x + dt * k3, u)
            return x + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
        return f

    def L(x, u):
        d = x - xref
        return np.sum(Q * d * d) + np.sum(R * u * u) + 0.3 * x[0] * x[2] + 0.01 * np.exp(0.1 * u[0])

    def Lf(x):
        d = x - xref
        return np.sum(QF * d * d) + 0.5 * x[0] * x[1]

    rng = np.random.default_rng(2)
    x0s = xref + rng.normal(0, 0.3, (n_states, 6)) * np.array([1, 1, 0.3, 0.5, 0.5, 0.5])
    u0s = np.full((n_states, N, 2), phys[0] * phys[3] / 2) + rng.normal(0, 0.2, (n_states, N, 2))
    u0s = u0s.astype(np.float32).astype(np.float64)            # what the device is handed
    x0s = x0s.astype(np.float32).astype(np.float64)
    out = dict(phys=phys, Q=Q, R=R, QF=QF, x_ref=xref, dt=np.array(dt), x0=x0s, u_init=u0s, N=np.array(N),
               max_iter=np.array(max_iter), tol=np.array(1e-3))
    for method in ("euler", "rk4"):
        for i in range(n_states):
            il = iLQR_TF(make_f(method), L, Lf, x0s[i], [u for u in u0s[i]], N, dt=dt, max_iter=max_iter, tol=1e-3)
            u_fin, x_fin = il.optimize(xref)
            lg = _pad_logs(il.logs, N, 6, 2, max_iter)
            for k_, v in lg.items():
                out[f"{method}_s{i}_{k_}"] = v
            out[f"{method}_s{i}_u_final"] = np.array(u_fin)
            out[f"{method}_s{i}_x_final"] = x_fin
            print(f"  planar/{method} state {i}: {len(il.logs)} iterations")
    save("user_planar.npz", **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(1)
    if sys.argv[1:] == ["--only", "int8"]:
        gen_int8("cartpole"); gen_int8("quadrotor")
        sys.exit(0)
    if sys.argv[1:] == ["--only", "opt_rk4"]:
        gen_optimize("cartpole", 30, 4, 8, method="rk4", tag="_rk4"); gen_optimize("quadrotor", 30, 2, 3, method="rk4", tag="_rk4")
        sys.exit(0)
    if sys.argv[1:] == ["--only", "lqr"]:
        gen_lqr()
        sys.exit(0)
    if sys.argv[1:] == ["--only", "user_planar"]:
        gen_user_planar()
        sys.exit(0)
    if sys.argv[1:] == ["--only", "hybrid_cartpole"]:
        gen_hybrid("cartpole", max_iter=6)
        sys.exit(0)
    if sys.argv[1:] == ["--only", "dataset"]:
        gen_dataset("cartpole", 30, 3, 5, 5); gen_dataset("quadrotor", 50, 2, 3, 1)
        sys.exit(0)
    for mdl in ["cartpole", "quadrotor"]:
        gen_dyn_cost(mdl)
    gen_sweep("cartpole", 30, 4); gen_sweep("cartpole", 50, 4)
    gen_sweep("quadrotor", 30, 2); gen_sweep("quadrotor", 50, 3)
    gen_sweep("cartpole", 30, 2, method="rk4", tag="_rk4"); gen_sweep("quadrotor", 30, 1, method="rk4", tag="_rk4")
    gen_forward("cartpole", 30); gen_forward("quadrotor", 50)
    gen_optimize("cartpole", 30, 8, 12); gen_optimize("quadrotor", 50, 4, 6)
    gen_optimize("cartpole", 30, 4, 8, method="rk4", tag="_rk4"); gen_optimize("quadrotor", 30, 2, 3, method="rk4", tag="_rk4")
    gen_warm("cartpole", 30, 6); gen_warm("quadrotor", 50, 3)
    for mdl in ["cartpole", "quadrotor"]:
        export_weights(mdl)
    gen_tf("cartpole", 30); gen_tf("quadrotor", 50)
    gen_hybrid()
    gen_hybrid("cartpole", max_iter=6)
    gen_dataset("cartpole", 30, 3, 5, 5); gen_dataset("quadrotor", 50, 2, 3, 1)
    gen_lqr()
    gen_int8("cartpole"); gen_int8("quadrotor")
    gen_user_planar()
