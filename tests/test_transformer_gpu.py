"""GPU parity of the fused transformer forward (quattro_tf_forward_bf16) on the reference's shipped checkpoints
(re-exported as plain arrays in tests/golden/tf_weights_*.npz) against golden predictions of the reference module.

Tolerances: the reference deploys fp16 on CPU (5.9e-4 from fp64); a bf16 evaluation of the same
weights sits at ~6e-3.  Asserted here: <= 2e-2 relative Frobenius vs the reference module in fp32, and the distance to
the reference's own fp16 output is checked at the same bound.  The kernel's own arithmetic (fp32 accumulation, fp32
LayerNorm/softmax/residual) is isolated by comparing with the fp64 oracle evaluated on bf16-ROUNDED weights.
"""
import numpy as np
import pytest

from conftest import GOLDEN, load_golden, rel_fro

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import transformer as o_tf  # noqa: E402

DEV = "cuda:0"


def _bf16_round(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def _load(model):
    import os
    from quattro_ilqr_amd import TransformerILQR
    n, c = (12, 52) if model == "quadrotor" else (4, 5)
    return TransformerILQR(n, c, device=DEV).load(os.path.join(GOLDEN, f"tf_weights_{model}.npz"))


@pytest.mark.parametrize("model", ["quadrotor", "cartpole"])
def test_predict_matches_reference_module(model):
    g = load_golden(f"tf_{model}.npz")
    tf = _load(model)
    assert tf.prompt_len == g["prompt"].shape[1] and tf.target_len == g["pred_fp32"].shape[1]
    S = g["x_err"].shape[0]
    got = np.array([tf.predict(g["x_err"][i], g["prompt"][i]) for i in range(S)])
    assert got.shape == g["pred_fp32"].shape and got.dtype == np.float64
    e32, e16 = rel_fro(got, g["pred_fp32"]), rel_fro(got, g["pred_fp16"])
    print(f"{model}: bf16-MFMA vs reference fp32 {e32:.2e}, vs reference fp16 {e16:.2e}")
    assert e32 < 2e-2 and e16 < 2e-2
    for i in range(S):
        assert rel_fro(got[i], g["pred_fp32"][i]) < 3e-2, i
    # batched call == one call per sample, bit for bit
    xb = torch.as_tensor(g["x_err"].astype(np.float32), device=DEV).contiguous()
    pb = torch.as_tensor(g["prompt"].astype(np.float32), device=DEV).contiguous()
    batch = tf.predict_batch(xb, pb).double().cpu().numpy()
    assert np.array_equal(batch, got)


@pytest.mark.parametrize("model", ["quadrotor", "cartpole"])
def test_kernel_arithmetic_vs_oracle_on_bf16_weights(model):
    """Same bf16-rounded weight matrices on both sides: what is left is the rounding of activations to bf16 at the
    MFMA inputs (the oracle keeps them in fp64)."""
    g = load_golden(f"tf_{model}.npz")
    z = load_golden(f"tf_weights_{model}.npz")
    W = {k: z[k].astype(np.float32) for k in z.files if not k.startswith(("norm.", "hp."))}
    norm = {k[5:]: z[k].astype(np.float64) for k in z.files if k.startswith("norm.")}
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
    Wq = dict(W)
    for k in W:
        if k.endswith(("in_proj_weight", "out_proj.weight", "linear1.weight", "linear2.weight")) or k == "output_linear.weight":
            Wq[k] = _bf16_round(W[k])
    tf = _load(model)
    S = 4
    got = np.array([tf.predict(g["x_err"][i], g["prompt"][i]) for i in range(S)])
    want = np.array([o_tf.predict(Wq, norm, g["x_err"][i], g["prompt"][i], hp["nhead"], hp["prompt_len"]) for i in range(S)])
    err = rel_fro(got, want)
    print(f"{model}: kernel vs fp64 oracle on bf16 weights {err:.2e}")
    assert err < 1.5e-2


def test_predict_batch_large_and_input_checks():
    tf = _load("quadrotor")
    g = load_golden("tf_quadrotor.npz")
    B = 1024
    rng = np.random.default_rng(0)
    idx = rng.integers(0, g["x_err"].shape[0], B)
    xb = torch.as_tensor(g["x_err"][idx].astype(np.float32), device=DEV).contiguous()
    pb = torch.as_tensor(g["prompt"][idx].astype(np.float32), device=DEV).contiguous()
    out = tf.predict_batch(xb, pb)
    assert out.shape == (B, 49, 52) and bool(torch.isfinite(out).all())
    # identical inputs anywhere in the batch give identical outputs (no cross-sequence state)
    first = {}
    for j, i in enumerate(idx.tolist()):
        if i in first:
            assert torch.equal(out[j], out[first[i]])
        else:
            first[i] = j
    with pytest.raises(ValueError):
        tf.predict_batch(xb, pb[:, :, :10].contiguous())
    with pytest.raises(ValueError):
        tf.predict_batch(xb.cpu(), pb)
    with pytest.raises(IndexError):                        # 70 + 1 + 49 tokens > max_seq_len 110
        tf.predict_batch(torch.zeros((1, 70, 12), device=DEV), pb[:1].contiguous())


def test_hybrid_batched_solver_equals_dropin_with_device_predictor():
    """QuattroILQR (batched, predict_batch on device) and iLQR_TF (reference control flow, predict per call) with the
    same HIP predictor take the same decisions; and the run stays close to the reference's logged hybrid run."""
    import quattro_ilqr_amd as q
    g = load_golden("hybrid_quadrotor.npz")
    tf = _load("quadrotor")
    N = 50
    mpc = q.QuadrotorMPC(horizon=N, dt=0.01, integration_method="euler", transformer_model=tf, device=DEV)
    mpc.ilqr.max_iter = int(g["max_iter"])
    mpc.ilqr.x0 = g["x0"]
    u_fin, x_fin = mpc.ilqr.optimize(mpc.x_ref)
    n_it = len(mpc.ilqr.logs)
    alphas = [(-1.0 if lg["alpha"] is None else lg["alpha"]) for lg in mpc.ilqr.logs]
    assert n_it == int(g["n_iter"]) and alphas == list(g["alpha"][:n_it])
    # same decisions; the states drift apart over the 4 iterations because every iteration feeds predicted gains
    # that differ at ~1e-3 (bf16 here, fp16 in the reference) back through the rollout: 1.8e-2 measured
    assert rel_fro(x_fin, g["x_final"]) < 5e-2
    assert rel_fro(mpc.ilqr.logs[0]["x_seq"], g["x_seq"][0]) < 1e-6
    solver = q.QuattroILQR(mpc.device_model(), N, max_iter=int(g["max_iter"]), tf=tf, device=DEV, check_every=1,
                           state_offset=mpc.ilqr.get_state_offset())
    x0 = np.stack([g["x0"], g["x0"] + 0.01])
    out = solver.solve(x0, x_ref=mpc.x_ref)
    assert int(out["iters"][0]) == n_it
    # not bit-identical: the drop-in forms x_seq - x_ref + offset in fp64 on the host (like the reference), the batched
    # solver in fp32 on the device, so the predictor sees inputs that differ in the last bit
    assert rel_fro(out["u"][0].double().cpu().numpy(), np.array(u_fin)) < 1e-3
    assert rel_fro(out["x"][0].double().cpu().numpy(), x_fin) < 1e-3


def test_gains_mode_equals_unpacked_prediction():
    """quattro_tf_gains_bf16 writes exactly what predict + the reference's (T, m, 1+n) unpacking gives, into the rows
    t < min(T, N) of K / k, skips inactive trajectories, and drops rows past the horizon of an over-long prediction."""
    import os
    import torch
    from conftest import GOLDEN
    import quattro_ilqr_amd as q
    tf = q.TransformerILQR(12, 52, device=DEV).load(os.path.join(GOLDEN, "tf_weights_quadrotor.npz"))
    B, n, m, T = 9, 12, 4, tf.target_len
    g = torch.Generator(device="cpu").manual_seed(3)
    x_err = (0.3 * torch.randn((B, 51, n), generator=g)).to(DEV)
    prompt = torch.randn((B, 1, 52), generator=g).to(DEV)
    pred = tf.predict_batch(x_err, prompt)
    v = pred.reshape(B, T, m, 1 + n)
    for N in (50, 49, 40):                                   # T = 49: exact fit with one swept step, T = N, T > N
        K = torch.full((B, N, m, n), 7.0, device=DEV); k = torch.full((B, N, m), 7.0, device=DEV)
        active = torch.ones(B, dtype=torch.int32, device=DEV); active[[2, 5]] = 0
        tf.predict_gains(x_err, prompt, K, k, active)
        Tn = min(T, N)
        live = active.bool()
        assert torch.equal(K[live][:, :Tn], v[live][:, :Tn, :, 1:]) and torch.equal(k[live][:, :Tn], v[live][:, :Tn, :, 0])
        assert bool((K[live][:, Tn:] == 7.0).all()) and bool((k[live][:, Tn:] == 7.0).all())
        assert bool((K[~live] == 7.0).all()) and bool((k[~live] == 7.0).all())
    with pytest.raises(ValueError):
        tf.predict_gains(x_err, prompt, torch.zeros((B, 50, 3, 12), device=DEV), torch.zeros((B, 50, 3), device=DEV))


@pytest.mark.parametrize("shape", [
    dict(n=4, c=5, ns=11, P=2, T=8, ff=64, layers=1),        # L = 21: one wave per sequence, the smallest FFN
    dict(n=4, c=5, ns=31, P=5, T=25, ff=192, layers=2),      # L = 61: two waves
    dict(n=12, c=52, ns=51, P=5, T=25, ff=320, layers=2),    # L = 81: three waves (uneven LDS-DMA shares), prompt of 5 rows
    dict(n=12, c=52, ns=51, P=1, T=49, ff=1024, layers=1),   # L = 101: four waves, the 1024-wide parameter block
    dict(n=12, c=52, ns=64, P=32, T=32, ff=512, layers=3),   # L = 128: no padding token, prompt rows spanning a whole tile
])
def test_every_kernel_instantiation_against_the_oracle(shape):
    """The shipped checkpoints exercise two of the eight instantiations of tf_stream_kernel (2 and 4 waves, ff <= 512).
    Random-init models of other shapes — one and three waves, ff up to 1024 and down to 64, prompts that straddle token
    tiles, a sequence that fills all 128 slots — against the fp64 oracle on the bf16-rounded weights."""
    from quattro_ilqr_amd import TransformerILQR
    tf = TransformerILQR.random_init(shape["n"], shape["c"], prompt_len=shape["P"], target_len=shape["T"],
                                     num_decoder_layers=shape["layers"], dim_feedforward=shape["ff"], max_seq_len=128,
                                     seed=7, device=DEV)
    rng = np.random.default_rng(5)
    B = 3
    x = rng.standard_normal((B, shape["ns"], shape["n"]))
    pr = rng.standard_normal((B, shape["P"], shape["c"]))
    got = tf.predict_batch(torch.as_tensor(x, dtype=torch.float32, device=DEV).contiguous(),
                           torch.as_tensor(pr, dtype=torch.float32, device=DEV).contiguous()).double().cpu().numpy()
    Wq = {k: (_bf16_round(v) if k.endswith(("in_proj_weight", "out_proj.weight", "linear1.weight", "linear2.weight",
                                            "embed.weight")) or k == "output_linear.weight" else v)
          for k, v in tf._w.items()}
    want = np.array([o_tf.predict(Wq, tf._norm, x[i].astype(np.float32).astype(np.float64),
                                  pr[i].astype(np.float32).astype(np.float64), 4, shape["P"]) for i in range(B)])
    assert got.shape == want.shape == (B, shape["T"], shape["c"]) and np.isfinite(got).all()
    err = rel_fro(got, want)
    print(f"{shape}: kernel vs fp64 oracle on bf16 weights {err:.2e}")
    assert err < 1.5e-2


@pytest.mark.parametrize("model", ["quadrotor", "cartpole"])
def test_fp16_operand_variant_tracks_the_reference_more_closely(model):
    """precision="fp16" (quattro_tf_forward_f16 / quattro_tf_gains_f16): the same kernel with fp16 MFMA operands — the
    shipped checkpoints are fp16, so the weights are exact and the activations carry three more mantissa bits.  Against
    the reference module's fp32 output the error must be well below the bf16 variant's, and the distance to the
    reference's own fp16 output of the same order as that output's distance to fp32 (transformer_ilqr.py:317-319)."""
    import os
    from quattro_ilqr_amd import TransformerILQR
    g = load_golden(f"tf_{model}.npz")
    n, c = (12, 52) if model == "quadrotor" else (4, 5)
    path = os.path.join(GOLDEN, f"tf_weights_{model}.npz")
    tf16 = TransformerILQR(n, c, device=DEV, precision="fp16").load(path)
    tfb = TransformerILQR(n, c, device=DEV).load(path)
    xb = torch.as_tensor(g["x_err"].astype(np.float32), device=DEV).contiguous()
    pb = torch.as_tensor(g["prompt"].astype(np.float32), device=DEV).contiguous()
    got16 = tf16.predict_batch(xb, pb).double().cpu().numpy()
    gotb = tfb.predict_batch(xb, pb).double().cpu().numpy()
    e16, eb = rel_fro(got16, g["pred_fp32"]), rel_fro(gotb, g["pred_fp32"])
    ref16 = rel_fro(g["pred_fp16"], g["pred_fp32"])
    print(f"{model}: fp16-MFMA vs reference fp32 {e16:.2e} (bf16-MFMA {eb:.2e}; the reference's own fp16 run {ref16:.2e})")
    assert e16 < 0.5 * eb
    assert e16 < 3.0 * max(ref16, 1e-4)
    # single-sample predict == batched, bit for bit; the two precisions are different kernels' results
    one = np.array([tf16.predict(g["x_err"][i], g["prompt"][i]) for i in range(3)])
    assert np.array_equal(one, got16[:3])
    assert not np.array_equal(got16, gotb)


def test_fp16_gains_mode_and_precision_mismatch_is_rejected():
    import ctypes
    import os
    from quattro_ilqr_amd import TransformerILQR, _lib
    g = load_golden("tf_quadrotor.npz")
    path = os.path.join(GOLDEN, "tf_weights_quadrotor.npz")
    tf16 = TransformerILQR(12, 52, device=DEV, precision="fp16").load(path)
    B, N, n, m = 8, 50, 12, 4
    xb = torch.as_tensor(g["x_err"][:B].astype(np.float32), device=DEV).contiguous()
    pb = torch.as_tensor(g["prompt"][:B].astype(np.float32), device=DEV).contiguous()
    pred = tf16.predict_batch(xb, pb)
    K = torch.zeros((B, N, m, n), dtype=torch.float32, device=DEV)
    k = torch.zeros((B, N, m), dtype=torch.float32, device=DEV)
    tf16.predict_gains(xb, pb, K, k)
    T = tf16.target_len
    rows = pred.view(B, T, m, 1 + n)
    assert torch.equal(k[:, :T], rows[..., 0]) and torch.equal(K[:, :T], rows[..., 1:])
    # a weight set packed for one precision is refused by the other family of entry points
    s = tf16._struct(int(xb.shape[1]))
    out = torch.empty_like(pred)
    P = ctypes.c_void_p
    rc = _lib.load().quattro_tf_forward_bf16(ctypes.byref(s), P(xb.data_ptr()), P(pb.data_ptr()), B, P(out.data_ptr()), None)
    assert rc == _lib.ERR_BAD_ARG
    with pytest.raises(ValueError):
        TransformerILQR(12, 52, device=DEV, precision="fp8")


def test_predictor_shapes_beyond_the_fused_kernel_run_layer_wise_in_fp32():
    """A predictor with the reference constructor's default shape (d_model 64, 8 heads: transformer_ilqr.py:30) has no fused
    bf16 kernel; its forward runs through the fp32 kernels of the training step (MFMA GEMMs, MFMA attention, LayerNorm) —
    on the device, several launches — and equals the fp64 oracle to fp32 accuracy; gains mode and the active mask too."""
    import os
    from quattro_ilqr_amd import TransformerILQR, training
    n, m, N, P = 4, 1, 30, 10
    c = m * (1 + n)
    T = N + 1 - P
    params, buffers = training.init_params(n, c, 64, 8, 3, 128, 100, T, seed=4, device="cpu")
    W = {k: v.detach().numpy() for k, v in params.items()}
    W["pos_encoder.pe"] = buffers["pos_encoder.pe"].numpy()
    rng = np.random.default_rng(1)
    norm = dict(x_mean=rng.standard_normal(n), x_std=1.0 + rng.random(n), u_mean=rng.standard_normal(c), u_std=1.0 + rng.random(c))
    hp = dict(target_len=T, prompt_len=P, state_dim=n, control_dim=c, d_model=64, nhead=8, num_decoder_layers=3,
              dim_feedforward=128, dropout=0.1, max_seq_len=100)
    tf = TransformerILQR(n, c, device=DEV).load_arrays(W, norm, hp)
    assert not tf.fused_kernel_covers()
    B = 9
    x = rng.standard_normal((B, N + 1, n))
    pr = rng.standard_normal((B, P, c))
    want = np.array([o_tf.predict(W, norm, x[i], pr[i], 8, P) for i in range(B)])
    xb, pb = (torch.as_tensor(a.astype(np.float32), device=DEV).contiguous() for a in (x, pr))
    got = tf.predict_batch(xb, pb).double().cpu().numpy()
    assert rel_fro(got, want) < 2e-5
    assert rel_fro(tf.predict(x[0], pr[0]), want[0]) < 2e-5
    K = torch.full((B, N, m, n), -7.0, device=DEV)
    k = torch.full((B, N, m), -7.0, device=DEV)
    active = torch.ones(B, dtype=torch.int32, device=DEV)
    active[::2] = 0
    tf.predict_gains(xb, pb, K, k, active)
    rows = torch.as_tensor(got, device=DEV).float().view(B, T, m, 1 + n)
    assert torch.allclose(k[1::2, :T], rows[1::2, ..., 0], atol=1e-6) and torch.allclose(K[1::2, :T], rows[1::2, ..., 1:], atol=1e-6)
    assert bool((k[::2] == -7.0).all()) and bool((K[::2] == -7.0).all()) and bool((K[:, T:] == -7.0).all())
    with pytest.raises(NotImplementedError):
        tf.prepare(N + 1)                                   # graph capture needs the fused kernel


def test_hybrid_solve_with_the_shipped_checkpoint_against_the_pure_solve():
    """what the shipped quadrotor predictor buys on this GPU, pinned on the G8 start (the reference's own hybrid
    run) and on a batch of the bench's cold starts: the hybrid solve stops after FEWER iterations than the pure one (8 vs 11
    on the G8 start, 8.3 vs 14.6 on average) on a cost within a stated factor of the pure solve's (measured: G8 start 1.13x; batch
    median 1.16x, mean 1.37x, worse on 82 % of the starts and better on 17 % — bounds 1.5x / 1.5x / 2x).  The numbers behind the
    product default 'pure' (bench extras.predictor_payoff)."""
    import os
    import quattro_ilqr_amd as q
    from conftest import GOLDEN
    g = load_golden("hybrid_quadrotor.npz")
    md = q.quadrotor_model()
    N = 50
    tf = q.TransformerILQR(12, 52, device=DEV).load(os.path.join(GOLDEN, "tf_weights_quadrotor.npz"))
    off = np.asarray(g["state_offset"], dtype=np.float64)
    rng = np.random.default_rng(1234)
    xb = np.asarray(md.x_ref) + rng.uniform(-1, 1, (255, 12)) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    x0 = np.concatenate([np.asarray(g["x0"], dtype=np.float64)[None], xb], axis=0)
    pure = q.QuattroILQR(md, N, max_iter=100, tol=1e-3, device=DEV).solve(x0)
    Jp, itp = pure["cost"].cpu().numpy().copy(), pure["iters"].cpu().numpy().copy()
    hyb = q.QuattroILQR(md, N, max_iter=100, tol=1e-3, tf=tf, state_offset=off, device=DEV).solve(x0)
    Jh, ith = hyb["cost"].cpu().numpy(), hyb["iters"].cpu().numpy()
    ratio = Jh / Jp
    print(f"shipped predictor vs pure, G8 start: iterations {ith[0]} vs {itp[0]}, cost {Jh[0]:.4f} vs {Jp[0]:.4f} (x{ratio[0]:.3f}); "
          f"256 cold starts: iterations mean {ith.mean():.1f} vs {itp.mean():.1f}, cost ratio median {np.median(ratio):.3f} mean {ratio.mean():.3f} "
          f"max {ratio.max():.2f}")
    assert int(hyb["status"].abs().sum()) == 0 and np.all(np.isfinite(Jh))
    print(f"hybrid better than pure on {100 * np.mean(ratio < 0.99):.0f} % of the starts, worse on {100 * np.mean(ratio > 1.01):.0f} %")
    assert ratio[0] < 1.5
    assert np.median(ratio) < 1.5 and ratio.mean() < 2.0
    assert ith.mean() < itp.mean()
