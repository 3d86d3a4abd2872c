"""Oracle transformer forward from raw weight arrays (NumPy).  TEST INFRASTRUCTURE ONLY.

Restates the reference's predictor
  * quattro_ilqr_tf/transformer_model.py:122-138  TransformerPredictor.forward
  * quattro_ilqr_tf/transformer_model.py:77-80    PositionalEncoding.forward (dropout = identity in eval)
  * quattro_ilqr_tf/transformer_model.py:37-50    DataNormalizer
  * quattro_ilqr_tf/transformer_ilqr.py:311-325   TransformerILQR.predict
and the third-party blocks it is built from (torch.nn.TransformerEncoderLayer as configured there:
batch_first, post-LayerNorm, ReLU, eps 1e-5, additive -inf causal mask, softmax(QK^T/sqrt(hd))V).
torch itself is not used here; the restatement is pinned by tests/golden/tf_*.npz, which hold the
reference module's own outputs on the shipped checkpoints.

Weight dict keys are the reference state_dict names (see tests/golden/make_golden.py: export_weights).
"""
import numpy as np


def hyper_from_weights(w):
    d = w["state_embed.weight"].shape[0]
    n_layers = 0
    while f"transformer_decoder.layers.{n_layers}.linear1.weight" in w:
        n_layers += 1
    return dict(
        d_model=d, state_dim=w["state_embed.weight"].shape[1], control_dim=w["control_embed.weight"].shape[1],
        ff=w["transformer_decoder.layers.0.linear1.weight"].shape[0], n_layers=n_layers,
        target_len=w["target_embedding"].shape[0], max_seq_len=w["pos_encoder.pe"].shape[1])


def _layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def forward(w, x_norm, prompt_norm, nhead, dtype=np.float64, return_hidden=False):
    """x_norm (B, N+1, n), prompt_norm (B, P, c) -> (B, T, c) normalised prediction."""
    W = {k: np.asarray(v).astype(dtype) for k, v in w.items()}
    hp = hyper_from_weights(W)
    d, T = hp["d_model"], hp["target_len"]
    x_norm = np.asarray(x_norm, dtype=dtype)
    prompt_norm = np.asarray(prompt_norm, dtype=dtype)
    Bt = x_norm.shape[0]
    x_emb = x_norm @ W["state_embed.weight"].T + W["state_embed.bias"]
    u_emb = prompt_norm @ W["control_embed.weight"].T + W["control_embed.bias"]
    tgt = np.broadcast_to(W["target_embedding"], (Bt, T, d))
    h = np.concatenate([x_emb, u_emb, tgt], axis=1)
    Lseq = h.shape[1]
    h = h + W["pos_encoder.pe"][:, :Lseq]
    hidden = [h.copy()]
    hd = d // nhead
    causal = np.triu(np.ones((Lseq, Lseq), dtype=bool), 1)
    for li in range(hp["n_layers"]):
        p = f"transformer_decoder.layers.{li}."
        qkv = h @ W[p + "self_attn.in_proj_weight"].T + W[p + "self_attn.in_proj_bias"]
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        split = lambda a: a.reshape(Bt, Lseq, nhead, hd).transpose(0, 2, 1, 3)
        q, k, v = split(q), split(k), split(v)
        s = (q @ k.transpose(0, 1, 3, 2)) / np.sqrt(dtype(hd))
        s = np.where(causal, -np.inf, s)
        s = s - s.max(axis=-1, keepdims=True)
        e = np.exp(s)
        a = e / e.sum(axis=-1, keepdims=True)
        o = (a @ v).transpose(0, 2, 1, 3).reshape(Bt, Lseq, d)
        o = o @ W[p + "self_attn.out_proj.weight"].T + W[p + "self_attn.out_proj.bias"]
        h = _layer_norm(h + o, W[p + "norm1.weight"], W[p + "norm1.bias"])
        f = np.maximum(h @ W[p + "linear1.weight"].T + W[p + "linear1.bias"], 0)
        f = f @ W[p + "linear2.weight"].T + W[p + "linear2.bias"]
        h = _layer_norm(h + f, W[p + "norm2.weight"], W[p + "norm2.bias"])
        hidden.append(h.copy())
    out = h[:, -T:, :] @ W["output_linear.weight"].T + W["output_linear.bias"]
    if return_hidden:
        return out, hidden
    return out


def predict(w, norm, x_seq, kK_seq, nhead, prompt_len, dtype=np.float64):
    """Single-sample predict(): normalise -> last prompt_len rows -> forward -> de-normalise.
    x_seq (N+1, n), kK_seq (>=P, c) -> (T, c)."""
    x_n = (np.asarray(x_seq) - norm["x_mean"]) / norm["x_std"]
    u_n = (np.asarray(kK_seq) - norm["u_mean"]) / norm["u_std"]
    u_n = u_n[-prompt_len:, :]
    y = forward(w, x_n.astype(np.float32)[None], u_n.astype(np.float32)[None], nhead, dtype=dtype)[0]
    return y * norm["u_std"] + norm["u_mean"]
