"""The device training step (csrc/tf_train.hip through the C ABI) against fp32 torch autograd of the same network
(`training.forward`, which tests/test_training_*.py pin to the REFERENCE module's outputs on the shipped checkpoints):
loss, prediction and every parameter gradient, with and without dropout; Adam against torch.optim.Adam; fit() end to end.
Reference: quattro_ilqr_tf/transformer_ilqr.py:102-208 (fit), transformer_model.py:85-138 (the network)."""
import math
import sys

import numpy as np
import pytest

from conftest import PKG_DIR

pytestmark = pytest.mark.gpu
sys.path.insert(0, PKG_DIR)
DEV = "cuda:0"

SHAPES = {
    # state_dim, control_dim, d_model, nhead, layers, ff, n_state_tok, prompt_len, target_len
    "quadrotor": (12, 52, 128, 4, 3, 512, 51, 1, 49),     # the shipped quadrotor predictor (N = 50)
    "cartpole": (4, 5, 128, 4, 2, 256, 31, 5, 26),         # the cart-pole predictor of the examples (N = 30, P = 5)
    "small": (3, 7, 64, 2, 1, 96, 6, 2, 5),                # ragged tiles: nothing is a multiple of the GEMM tile
    "long": (4, 5, 64, 2, 2, 128, 64, 32, 32),             # 128 tokens: every attention tile full, the longest supported
    "default": (4, 5, 64, 8, 3, 128, 31, 10, 21),          # the reference constructor's defaults (transformer_ilqr.py:30): head dimension 8
    "hd16": (12, 52, 128, 8, 1, 256, 21, 3, 18),           # head dimension 16
    "d96": (4, 5, 96, 4, 2, 80, 9, 3, 7),                  # d_model not a multiple of the wave width (head dimension 24)
}


def _setup(shape, B, seed, dropout=0.0):
    import torch
    from quattro_ilqr_amd import train_hip, training
    n, c, d, H, layers, ff, NS, P, T = shape
    params, buffers = training.init_params(n, c, d, H, layers, ff, NS + P + T + 9, T, seed=seed, device=DEV)   # (pe longer than L)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, v in params.items():            # biases / LayerNorm vectors away from their 0 / 1 initial values
            if k.endswith("bias") or "norm" in k:
                v += 0.2 * torch.randn(v.shape, generator=g).to(DEV)
    tr = train_hip.HipTrainer(n, c, d, H, layers, ff, NS, P, T, dropout, buffers["pos_encoder.pe"].cpu().numpy(), DEV)
    tr.load_state_dict({k: v.detach() for k, v in params.items()})
    x = torch.randn((B, NS, n), generator=g).to(DEV)
    u = torch.randn((B, P, c), generator=g).to(DEV)
    y = torch.randn((B, T, c), generator=g).to(DEV)
    return tr, params, buffers, x, u, y


def _rel(a, b):
    a, b = a.double(), b.double()
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)


@pytest.mark.parametrize("name", list(SHAPES))
def test_loss_prediction_and_every_gradient_match_fp32_autograd(name):
    import torch
    import torch.nn.functional as F
    from quattro_ilqr_amd import training
    shape = SHAPES[name]
    tr, params, buffers, x, u, y = _setup(shape, B=6 if name != "small" else 5, seed=3)
    loss, pred = tr.forward_backward(x, u, y, training=True, want_pred=True)
    ref_pred = training.forward(params, buffers, x, u, shape[3])
    ref_loss = F.mse_loss(ref_pred, y)
    ref_loss.backward()
    assert _rel(pred, ref_pred.detach()) < 2e-5
    assert abs(float(loss.item()) - float(ref_loss.item())) < 2e-5 * abs(float(ref_loss.item()))
    worst = 0.0
    for k, v in params.items():
        got = tr.view(tr.grads, k)
        err = _rel(got, v.grad)
        worst = max(worst, err)
        assert err < 2e-4, (k, err)
    # the padding floats between blocks stay zero (Adam would otherwise move them)
    total = sum(int(np.prod(s)) for _, _, s in tr.shapes.values())
    assert abs(float(tr.grads.abs().sum()) - sum(float(tr.view(tr.grads, k).abs().sum()) for k in tr.shapes)) < 1e-3
    assert total <= tr.n_params


def _forward_with_masks(params, buffers, x, u, nhead, masks):
    """training.forward with explicit dropout factors per site (0: positions; per layer: attention weights, out-proj,
    ff hidden, ff output) — the same network, term for term (transformer_model.py:122-138)."""
    import torch
    import torch.nn.functional as F
    W = params
    T, d = W["target_embedding"].shape
    B = x.shape[0]
    h = torch.cat([F.linear(x, W["state_embed.weight"], W["state_embed.bias"]),
                   F.linear(u, W["control_embed.weight"], W["control_embed.bias"]),
                   W["target_embedding"].unsqueeze(0).expand(B, T, d)], dim=1)
    L = h.shape[1]
    h = (h + buffers["pos_encoder.pe"][:, :L]) * masks[0].view(B, L, d)
    hd = d // nhead
    causal = torch.triu(torch.ones(L, L, dtype=torch.bool, device=h.device), diagonal=1)
    n_layers = sum(1 for k in W if k.endswith("self_attn.in_proj_weight"))
    for i in range(n_layers):
        q = f"transformer_decoder.layers.{i}."
        qkv = F.linear(h, W[q + "self_attn.in_proj_weight"], W[q + "self_attn.in_proj_bias"])
        qh, kh, vh = (t.reshape(B, L, nhead, hd).transpose(1, 2) for t in qkv.split(d, dim=-1))
        s = (qh @ kh.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
        a = torch.softmax(s.masked_fill(causal, float("-inf")), dim=-1) * masks[1 + 4 * i].view(B, nhead, L, L)
        o = (a @ vh).transpose(1, 2).reshape(B, L, d)
        o = F.linear(o, W[q + "self_attn.out_proj.weight"], W[q + "self_attn.out_proj.bias"])
        h = F.layer_norm(h + o * masks[2 + 4 * i].view(B, L, d), (d,), W[q + "norm1.weight"], W[q + "norm1.bias"], 1e-5)
        f = torch.relu(F.linear(h, W[q + "linear1.weight"], W[q + "linear1.bias"])) * masks[3 + 4 * i].view(B, L, -1)
        f = F.linear(f, W[q + "linear2.weight"], W[q + "linear2.bias"])
        h = F.layer_norm(h + f * masks[4 + 4 * i].view(B, L, d), (d,), W[q + "norm2.weight"], W[q + "norm2.bias"], 1e-5)
    return F.linear(h[:, -T:, :], W["output_linear.weight"], W["output_linear.bias"])


@pytest.mark.parametrize("name", ["quadrotor", "small"])
def test_dropout_forward_and_backward_see_the_same_masks(name):
    """With dropout the step must differentiate the network it actually evaluated: the masks the kernels hash on the fly
    (dumped through quattro_tf_train_dropout_mask_f32) go into a torch forward with explicit masks; loss and gradients
    agree.  Keep rates are what p says, and a different seed gives a different loss."""
    import torch
    import torch.nn.functional as F
    shape = SHAPES[name]
    n, c, d, H, layers, ff, NS, P, T = shape
    p, seed, B = 0.1, 1234567, 4
    tr, params, buffers, x, u, y = _setup(shape, B=B, seed=5, dropout=p)
    L = NS + P + T
    sizes = {0: B * L * d}
    for l in range(layers):
        sizes.update({1 + 4 * l: B * H * L * L, 2 + 4 * l: B * L * d, 3 + 4 * l: B * L * ff, 4 + 4 * l: B * L * d})
    masks = {s: tr.dropout_mask(seed, p, s, nel) for s, nel in sizes.items()}
    for s, m in masks.items():
        kept = float((m > 0).float().mean())
        assert abs(kept - (1 - p)) < 0.02, (s, kept)
        assert float(m.max()) == pytest.approx(1.0 / (1.0 - p), rel=1e-6)
    assert float((masks[0] != masks[2]).float().mean()) > 0.1          # sites are independent streams
    loss, pred = tr.forward_backward(x, u, y, seed=seed, training=True, want_pred=True)
    ref_pred = _forward_with_masks(params, buffers, x, u, H, masks)
    ref_loss = F.mse_loss(ref_pred, y)
    ref_loss.backward()
    assert _rel(pred, ref_pred.detach()) < 2e-5
    for k, v in params.items():
        assert _rel(tr.view(tr.grads, k), v.grad) < 2e-4, k
    l1 = float(loss.item())
    l2 = float(tr.forward_backward(x, u, y, seed=seed + 1, training=True).item())
    l_eval, _ = tr.evaluate(x, u, y)
    assert l1 != l2 and float(l_eval.item()) != l1
    # evaluation ignores the dropout rate entirely
    from quattro_ilqr_amd import training
    with torch.no_grad():
        assert abs(float(l_eval.item()) - float(F.mse_loss(training.forward(params, buffers, x, u, H), y))) < 2e-5 * l1


def test_adam_matches_torch_optim_adam():
    import torch
    g = torch.Generator().manual_seed(0)
    from quattro_ilqr_amd import train_hip
    shape = SHAPES["small"]
    tr, params, buffers, x, u, y = _setup(shape, B=3, seed=9)
    tr.lr = 2e-3
    ref = [v.detach().clone().requires_grad_(True) for v in params.values()]
    opt = torch.optim.Adam(ref, lr=2e-3)
    for step in range(5):
        fake = torch.randn(tr.n_params, generator=g).to(DEV) * (10.0 ** (step - 2))
        tr.grads.copy_(fake)
        for k, r in zip(params, ref):
            r.grad = tr.view(fake, k).clone()
        tr.adam_step()
        opt.step()
    for k, r in zip(params, ref):
        assert _rel(tr.view(tr.params, k), r.detach()) < 1e-6, k


def test_fit_on_the_device_backend_trains_and_matches_the_torch_backend():
    """fit(backend="hip") — the whole loop of transformer_ilqr.py:142-208 on the hand-written step — learns, and from the
    same initial weights and batches follows the torch-autograd backend's loss curve (dropout 0: same arithmetic up to
    fp32 summation order)."""
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import datagen
    md = q.cartpole_model()
    N, P = 30, 5
    rng = np.random.default_rng(2)
    B = 96
    x0 = np.stack([rng.uniform(-0.5, 0.5, B), np.zeros(B), rng.uniform(-0.5, 0.5, B), np.zeros(B)], axis=1)
    log = datagen.collect(q.QuattroILQR(md, N, max_iter=6, tol=1e-1, device=DEV), x0)
    perm = rng.permutation(len(log))
    n_train = int(0.8 * len(log))
    hist = {}
    for backend in ("hip", "torch"):
        tf = q.TransformerILQR(4, 5, prompt_len=P, d_model=128, nhead=4, num_decoder_layers=2, dim_feedforward=256,
                               dropout=0.0, max_seq_len=80, device=DEV)
        tf.fit(log.select(perm[:n_train]), log.select(perm[n_train:]), num_epochs=12, batch_size=32, learning_rate=1e-3,
               patience=12, backend=backend)
        hist[backend] = (np.array(tf.train_loss_history), np.array(tf.test_loss_history), tf)
    th, eh, tf_hip = hist["hip"]
    tt, et, _ = hist["torch"]
    assert th[-1] < 0.3 * th[0] and eh[-1] < 0.5 * eh[0]
    assert np.allclose(th[:3], tt[:3], rtol=2e-3), (th[:3], tt[:3])
    assert np.allclose(th, tt, rtol=0.15), (th, tt)
    # the weights trained on the device run in the bf16 inference kernel
    import torch
    x_data, kK_data = datagen.create_dataset(log.x_seq, log.k_seq, log.K_seq, P)
    x_err = torch.as_tensor(x_data[:16], device=DEV).contiguous()
    prompt = torch.as_tensor(kK_data[:16, -P:], device=DEV).contiguous()
    pred = tf_hip.predict_batch(x_err, prompt) if hasattr(tf_hip, "predict_batch") else None
    if pred is not None:
        assert bool(torch.isfinite(pred).all())


def test_fit_with_the_reference_default_dropout_learns_on_the_device_backend():
    """dropout = 0.1 (the reference constructor's default, transformer_ilqr.py:30): training applies the hashed masks, the
    test-set evaluation does not; the loss still goes down and the histories have the reference's shape."""
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import datagen
    md = q.cartpole_model()
    N, P = 30, 5
    rng = np.random.default_rng(3)
    B = 64
    x0 = np.stack([rng.uniform(-0.5, 0.5, B), np.zeros(B), rng.uniform(-0.5, 0.5, B), np.zeros(B)], axis=1)
    log = datagen.collect(q.QuattroILQR(md, N, max_iter=5, tol=1e-1, device=DEV), x0)
    perm = rng.permutation(len(log))
    n_train = int(0.8 * len(log))
    tf = q.TransformerILQR(4, 5, prompt_len=P, d_model=128, nhead=4, num_decoder_layers=2, dim_feedforward=256,
                           dropout=0.1, max_seq_len=80, device=DEV)
    tf.fit(log.select(perm[:n_train]), log.select(perm[n_train:]), num_epochs=10, batch_size=24, learning_rate=1e-3,
           patience=10)
    assert tf.fit_backend == "hip"
    assert len(tf.train_loss_history) == 10 and len(tf.test_loss_history) == 10
    assert tf.train_loss_history[-1] < 0.5 * tf.train_loss_history[0]
    assert tf.test_loss_history[-1] < 0.6 * tf.test_loss_history[0]
    assert tf.test_loss_history[-1] < tf.train_loss_history[-1] * 1.5      # evaluation runs without dropout


def test_data_parallel_fit_on_the_device_backend_two_ranks_one_gpu(tmp_path):
    """The data-parallel path of fit() with the hand-written step: two ranks (sharing this box's one GPU, so the collective
    is gloo here — on a node it is RCCL over the flat gradient array) against the single-process run: same loss histories
    and weights to fp32 summation order; both ranks bit-identical to each other."""
    sys.path.insert(0, __import__("os").path.dirname(__file__))
    from test_training_cpu import _run_dp_workers
    one = _run_dp_workers(tmp_path, 1, DEV, "hip")[0]
    two = _run_dp_workers(tmp_path, 2, DEV, "hip")
    assert one["backend"] == "hip" and two[0]["backend"] == "hip"
    for r in two:
        assert np.allclose(r["train"], one["train"], rtol=5e-4), (r["train"], one["train"])
        assert np.allclose(r["test"], one["test"], rtol=5e-4)
        assert abs(r["w"] - one["w"]) < 5e-4 * abs(one["w"])
    assert two[0]["train"] == two[1]["train"] and two[0]["w"] == two[1]["w"]


def test_evaluate_chunks_large_sets():
    """fit() evaluates the whole test set, and _predict_fp32 a whole solver batch, in one evaluate() call;
    the forward-only step still carves the full training workspace (~2.8 MB per sequence) and launches dim3(H, batch)
    attention grids.  evaluate() therefore walks the set in chunks of EVAL_CHUNK sequences: predictions bit-identical to
    evaluating the pieces one by one, loss = the mean over the whole set (ragged last chunk weighted by its size)."""
    import torch
    import torch.nn.functional as F
    from quattro_ilqr_amd import train_hip, training
    shp = (4, 5, 64, 8, 2, 128, 31, 5, 25)           # the reference constructor's default width, cart-pole token counts
    prm, buf = training.init_params(*shp[:6], 100, shp[8], seed=3, device=DEV)
    tr = train_hip.HipTrainer(*shp, 0.0, buf["pos_encoder.pe"].cpu().numpy(), DEV)
    tr.load_state_dict({k: v.detach() for k, v in prm.items()})
    tr.EVAL_CHUNK = 96
    B = 96 * 2 + 41
    g = torch.Generator().manual_seed(5)
    x, u, y = (torch.randn(sz, generator=g).to(DEV) for sz in ((B, 31, 4), (B, 5, 5), (B, 25, 5)))
    loss, pred = tr.evaluate(x, u, y)
    loss = float(loss.item())
    ws_bytes = tr._ws.numel()
    for lo in range(0, B, 96):
        _, pp = tr.evaluate(x[lo:lo + 96].contiguous(), u[lo:lo + 96].contiguous())
        assert torch.equal(pp, pred[lo:lo + 96])
    assert tr._ws.numel() == ws_bytes                                    # the workspace never grew beyond one chunk's
    want = float(F.mse_loss(pred, y).item())
    assert abs(loss - want) <= 2e-6 * abs(want), (loss, want)
    with torch.no_grad():
        ref = training.forward(prm, buf, x, u, shp[3])
    assert float((pred - ref).abs().max()) < 2e-4
    _, p_only = tr.evaluate(x, u)                                         # no target: prediction only
    assert torch.equal(p_only, pred)
