"""Training of the gain predictor on ROCm: `fit` / `save` of the reference's TransformerILQR.

Reference mirrored: quattro_ilqr_tf/transformer_ilqr.py `fit` :102-208 (Adam, MSE on the normalised target rows, shuffled
mini-batches, optional test set with early stopping and best-state restore, loss histories) and `save` :213-255 (the
directory with tf_model.pt + tf_model_normalizer.npz that `load` :259-304 reads), around the architecture of
quattro_ilqr_tf/transformer_model.py:85-138 (post-LN encoder layers with ReLU, causal mask, learnable target tokens,
sinusoidal positions, dropout after the positional encoding, on the attention weights and after each sub-block).

`fit` runs on one of two backends: "hip" — every mini-batch is the hand-written fp32 forward / backward / Adam of
csrc/tf_train.hip (train_hip.HipTrainer, the default on a GPU) — or "torch", a functional restatement in plain torch ops
under autograd (the CPU path, the fallback for shapes the kernels do not cover, and the reference the device step is
tested against; its forward is pinned to the same golden outputs as the oracle, tests/test_training_cpu.py).
The trained parameters carry the reference's state_dict names, so the result goes straight into the HIP inference
kernel (`TransformerILQR.load_arrays`) and into checkpoints the reference itself can load.
"""
import datetime
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import datagen


def init_params(state_dim, control_dim, d_model, nhead, num_layers, dim_feedforward, max_seq_len, target_len, seed=0,
                device="cpu"):
    """Parameters with the reference module's names, shapes and default initialisers (nn.Linear: U(-1/sqrt(in), 1/sqrt(in))
    for weight and bias; MultiheadAttention: Xavier-uniform in_proj, zero biases; LayerNorm: ones / zeros; target tokens
    N(0, 0.02^2)).  `pos_encoder.pe` is the fixed sinusoidal buffer."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    d, ff, c = d_model, dim_feedforward, control_dim

    def uni(shape, bound):
        return (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound

    def linear(o, i):
        return uni((o, i), 1.0 / math.sqrt(i)), uni((o,), 1.0 / math.sqrt(i))

    p = {}
    p["target_embedding"] = torch.randn((target_len, d), generator=g) * 0.02
    p["state_embed.weight"], p["state_embed.bias"] = linear(d, state_dim)
    p["control_embed.weight"], p["control_embed.bias"] = linear(d, c)
    p["output_linear.weight"], p["output_linear.bias"] = linear(c, d)
    for i in range(num_layers):
        q = f"transformer_decoder.layers.{i}."
        p[q + "self_attn.in_proj_weight"] = uni((3 * d, d), math.sqrt(6.0 / (3 * d + d)))
        p[q + "self_attn.in_proj_bias"] = torch.zeros(3 * d)
        p[q + "self_attn.out_proj.weight"] = uni((d, d), 1.0 / math.sqrt(d))
        p[q + "self_attn.out_proj.bias"] = torch.zeros(d)
        p[q + "linear1.weight"], p[q + "linear1.bias"] = linear(ff, d)
        p[q + "linear2.weight"], p[q + "linear2.bias"] = linear(d, ff)
        for nm in ("norm1", "norm2"):
            p[q + nm + ".weight"], p[q + nm + ".bias"] = torch.ones(d), torch.zeros(d)
    params = {k: v.to(device=device, dtype=torch.float32).requires_grad_(True) for k, v in p.items()}
    pos = torch.arange(max_seq_len, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * (-math.log(10000.0) / d))
    pe = torch.zeros(max_seq_len, d)
    pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
    buffers = {"pos_encoder.pe": pe[None].to(device)}
    return params, buffers


def forward(params, buffers, x_norm, prompt_norm, nhead, dropout=0.0, training=False):
    """(B, N+1, n), (B, P, c) normalised -> (B, T, c) normalised prediction.  transformer_model.py:122-138."""
    W = params
    T, d = W["target_embedding"].shape
    B = x_norm.shape[0]
    drop = (lambda t: F.dropout(t, dropout, True)) if (training and dropout > 0.0) else (lambda t: t)
    h = torch.cat([F.linear(x_norm, W["state_embed.weight"], W["state_embed.bias"]),
                   F.linear(prompt_norm, W["control_embed.weight"], W["control_embed.bias"]),
                   W["target_embedding"].unsqueeze(0).expand(B, T, d)], dim=1)
    L = h.shape[1]
    h = drop(h + buffers["pos_encoder.pe"][:, :L])
    hd = d // nhead
    causal = torch.triu(torch.ones(L, L, dtype=torch.bool, device=h.device), diagonal=1)
    n_layers = sum(1 for k in W if k.endswith("self_attn.in_proj_weight"))
    for i in range(n_layers):
        q = f"transformer_decoder.layers.{i}."
        qkv = F.linear(h, W[q + "self_attn.in_proj_weight"], W[q + "self_attn.in_proj_bias"])
        qh, kh, vh = (t.reshape(B, L, nhead, hd).transpose(1, 2) for t in qkv.split(d, dim=-1))
        s = (qh @ kh.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
        a = drop(torch.softmax(s.masked_fill(causal, float("-inf")), dim=-1))
        o = (a @ vh).transpose(1, 2).reshape(B, L, d)
        o = F.linear(o, W[q + "self_attn.out_proj.weight"], W[q + "self_attn.out_proj.bias"])
        h = F.layer_norm(h + drop(o), (d,), W[q + "norm1.weight"], W[q + "norm1.bias"], 1e-5)
        f = drop(torch.relu(F.linear(h, W[q + "linear1.weight"], W[q + "linear1.bias"])))
        f = F.linear(f, W[q + "linear2.weight"], W[q + "linear2.bias"])
        h = F.layer_norm(h + drop(f), (d,), W[q + "norm2.weight"], W[q + "norm2.bias"], 1e-5)
    return F.linear(h[:, -T:, :], W["output_linear.weight"], W["output_linear.bias"])


def _as_arrays(data, prompt_len):
    """DataFrame with x_seq / k_seq / K_seq columns (the reference's input), an IterationLog, or (x_data, kK_data)."""
    if isinstance(data, tuple) and len(data) == 2:
        return np.asarray(data[0], dtype=np.float32), np.asarray(data[1], dtype=np.float32)
    if isinstance(data, datagen.IterationLog):
        return datagen.create_dataset(data.x_seq, data.k_seq, data.K_seq, prompt_len)
    cols = {c: [np.asarray(v, dtype=np.float32) for v in data[c]] for c in ("x_seq", "k_seq", "K_seq")}   # DataFrame / dict
    return datagen.create_dataset(np.stack(cols["x_seq"]), np.stack(cols["k_seq"]), np.stack(cols["K_seq"]), prompt_len)


def _dist_world(distributed):
    """(rank, world) of the data-parallel group, or (0, 1).  `distributed`: "auto" = use torch.distributed when a process
    group is initialised with more than one rank; True = require it; False = never."""
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if distributed is True and not on:
        raise RuntimeError("fit(distributed=True) needs an initialised torch.distributed process group with world_size > 1")
    if distributed is False or not on:
        return 0, 1
    return dist.get_rank(), dist.get_world_size()


def fit(tf, data, test_data=None, num_epochs=50, batch_size=16, learning_rate=1e-3, patience=5, seed=0, verbose=False,
        backend="auto", distributed="auto"):
    """Train `tf` (a quattro_ilqr_amd.TransformerILQR built with the architecture hyper-parameters) in place and stage the
    result for the HIP inference kernel.  Returns tf; sets train_loss_history / test_loss_history like the reference.

    backend "hip": every mini-batch is one `quattro_tf_train_step_f32` (hand-written forward, loss, backward) and one
    `quattro_tf_adam_f32` (train_hip.HipTrainer); "torch": the functional restatement above under autograd (rocBLAS);
    "auto": "hip" on a GPU when the shape has kernels (head dimension <= 32, d_model <= 512, sequence <= 128 tokens),
    else "torch".  Same initial weights, same shuffles, same early stopping either way.

    Data parallel (not in the reference, which trains in one process): with a torch.distributed process group of W ranks —
    one per GPU, RCCL — every rank holds the data set, takes every W-th sequence of each mini-batch, and the gradients are
    summed with ONE all-reduce per step (the flat gradient array of the device backend; per tensor on the torch backend),
    weighted so that the step equals the single-process step on the whole mini-batch.  Parameters start identical (same
    seed) and stay identical (same reduced gradient on every rank); every rank evaluates the test set, so early stopping
    takes the same decision everywhere."""
    dev = tf.device
    P = tf.prompt_len
    x_data, kK_data = _as_arrays(data, P)
    if x_data.shape[0] == 0:
        raise ValueError("no sequences longer than prompt_len in the training data")
    T = x_data.shape[1]
    tf.target_len = T - P                                               # transformer_ilqr.py:106
    norm = datagen.fit_normalizer(x_data, kK_data)
    xn, up, ut = datagen.training_slices(x_data, kK_data, norm, P)
    xn_t, up_t, ut_t = (torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev) for a in (xn, up, ut))
    test = None
    if test_data is not None:
        xt, kt = _as_arrays(test_data, P)
        test = tuple(torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
                     for a in datagen.training_slices(xt, kt, norm, P))
    params, buffers = init_params(tf.state_dim, tf.control_dim, tf.d_model, tf.nhead, tf.num_decoder_layers,
                                  tf.dim_feedforward, tf.max_seq_len, tf.target_len, seed=seed, device=dev)
    if backend not in ("auto", "hip", "torch"):
        raise ValueError(f"backend must be 'auto', 'hip' or 'torch' (got {backend!r})")
    shape = (tf.state_dim, tf.control_dim, tf.d_model, tf.nhead, tf.num_decoder_layers, tf.dim_feedforward,
             xn_t.shape[1], P, tf.target_len)
    if backend != "torch":
        from . import train_hip
        on_gpu = torch.device(dev).type == "cuda"
        ok = on_gpu and train_hip.supported(*shape, dropout=tf.dropout)
        if backend == "hip" and not ok:
            raise NotImplementedError("fit(backend='hip'): needs a GPU and a predictor shape the training kernels cover "
                                      "(head dimension <= 32, d_model <= 512, sequence <= 128 tokens)")
        backend = "hip" if ok else "torch"
    if backend == "hip":
        trainer = train_hip.HipTrainer(*shape, tf.dropout, buffers["pos_encoder.pe"].cpu().numpy(), dev, lr=learning_rate)
        trainer.load_state_dict({k: v.detach() for k, v in params.items()})
    else:
        opt = torch.optim.Adam(list(params.values()), lr=learning_rate)
    rank, world = _dist_world(distributed)
    if world > 1:
        import torch.distributed as dist
    gen = torch.Generator(device="cpu").manual_seed(seed + 1)
    n = xn_t.shape[0]
    best, best_state, stale = float("inf"), None, 0
    tf.train_loss_history, tf.test_loss_history, tf.num_epochs = [], [], num_epochs
    tf.fit_backend = backend
    step = 0
    for epoch in range(num_epochs):
        perm = torch.randperm(n, generator=gen).to(dev)
        total = torch.zeros((), dtype=torch.float64, device=dev)
        for i in range(0, n, batch_size):
            idx_all = perm[i:i + batch_size]
            idx = idx_all[rank::world]                       # this rank's share of the mini-batch (all of it when world == 1)
            share = idx.numel() / idx_all.numel()            # weight of the local mean in the mini-batch mean
            step += 1
            if backend == "hip":
                if idx.numel() > 0:
                    loss = trainer.forward_backward(xn_t[idx].contiguous(), up_t[idx].contiguous(), ut_t[idx].contiguous(),
                                                    seed=(seed << 32) + step * world + rank, training=True)
                    batch_loss = loss[0].double() * idx.numel()
                else:
                    trainer.grads.zero_()
                    batch_loss = torch.zeros((), dtype=torch.float64, device=dev)
                if world > 1:
                    trainer.grads.mul_(share)
                    dist.all_reduce(trainer.grads)
                trainer.adam_step()
            else:
                opt.zero_grad(set_to_none=True)
                if idx.numel() > 0:
                    pred = forward(params, buffers, xn_t[idx], up_t[idx], tf.nhead, tf.dropout, training=True)
                    loss = F.mse_loss(pred, ut_t[idx])
                    loss.backward()
                    batch_loss = loss.detach().double() * idx.numel()
                else:
                    batch_loss = torch.zeros((), dtype=torch.float64, device=dev)
                if world > 1:
                    for v in params.values():
                        g = v.grad.mul_(share) if v.grad is not None else torch.zeros_like(v)
                        dist.all_reduce(g)
                        v.grad = g
                opt.step()
            total += batch_loss
        if world > 1:
            dist.all_reduce(total)
        tf.train_loss_history.append(float(total.item()) / n)
        if test is not None:
            if backend == "hip":
                tl = float(trainer.evaluate(test[0], test[1], test[2])[0].item())
            else:
                with torch.no_grad():
                    tl = float(F.mse_loss(forward(params, buffers, test[0], test[1], tf.nhead), test[2]).item())
            tf.test_loss_history.append(tl)
            if tl < best:
                best, stale = tl, 0
                best_state = (trainer.params.clone() if backend == "hip"
                              else {k: v.detach().clone() for k, v in params.items()})
            else:
                stale += 1
            if verbose:
                print(f"Epoch {epoch + 1}/{num_epochs}, Train Loss: {tf.train_loss_history[-1]:.6f}, Test Loss: {tl:.6f}")
            if stale >= patience:
                if verbose:
                    print(f"Early stopping triggered at epoch {epoch + 1}.")
                if backend == "hip":
                    trainer.params.copy_(best_state)
                else:
                    params = {k: v.requires_grad_(True) for k, v in best_state.items()}
                break
        elif verbose:
            print(f"Epoch {epoch + 1}/{num_epochs}, Train Loss: {tf.train_loss_history[-1]:.6f}")
    final = trainer.state_dict() if backend == "hip" else params
    weights = {k: v.detach().float().cpu().numpy() for k, v in final.items()}
    weights["pos_encoder.pe"] = buffers["pos_encoder.pe"].cpu().numpy()
    hp = dict(target_len=tf.target_len, prompt_len=P, state_dim=tf.state_dim, control_dim=tf.control_dim,
              d_model=tf.d_model, nhead=tf.nhead, num_decoder_layers=tf.num_decoder_layers,
              dim_feedforward=tf.dim_feedforward, dropout=tf.dropout, max_seq_len=tf.max_seq_len)
    return tf.load_arrays(weights, norm, hp)


def _is_plain_linear(name, sd):
    """Modules of the reference network that are exactly nn.Linear — what quantize_dynamic(model, {nn.Linear}) converts:
    the two embeddings, the output layer and every layer's linear1 / linear2.  (MultiheadAttention keeps its in_proj as
    a bare parameter and its out_proj is a NonDynamicallyQuantizableLinear: both stay fp32.)"""
    leaf = name.rsplit(".", 1)[-1]
    return (name + ".weight") in sd and (name + ".bias") in sd and leaf in ("state_embed", "control_embed", "output_linear",
                                                                            "linear1", "linear2")


def int8_state_dict(sd):
    """fp32 state dict -> the state dict the reference stores with quant_mode == "int8" (transformer_ilqr.py:225-226, :235):
    every plain nn.Linear replaced by torch's dynamically quantised Linear (per-tensor symmetric int8 weight; entries
    `<name>.scale`, `.zero_point`, `._packed_params.dtype`, `._packed_params._packed_params` = (qint8 weight, bias)), made
    with the same torch call on a one-layer module per Linear."""
    import warnings
    import torch.nn as nn
    out = {}
    names = sorted({k.rsplit(".", 1)[0] for k in sd if k.endswith(".weight")})
    lin = [n for n in names if _is_plain_linear(n, sd)]
    skip = {n + sfx for n in lin for sfx in (".weight", ".bias")}
    for k, v in sd.items():
        if k not in skip:
            out[k] = v
    for n in lin:
        w, b = sd[n + ".weight"].float(), sd[n + ".bias"].float()
        holder = nn.Sequential(nn.Linear(w.shape[1], w.shape[0]))
        with torch.no_grad():
            holder[0].weight.copy_(w)
            holder[0].bias.copy_(b)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = torch.ao.quantization.quantize_dynamic(holder, {nn.Linear}, dtype=torch.qint8)
        for k, v in q.state_dict().items():
            out[n + k[1:]] = v                                  # "0.scale" -> "<name>.scale"
    return out


def dequantize_state_dict(sd):
    """The inverse for loading: (qint8 weight, bias) pairs back to fp32 `.weight` / `.bias`; everything else unchanged.  The
    weights then sit on the int8 grid (scale x integer); activations are NOT quantised on the device (the reference's
    int8 path quantises them per call on the CPU) — see DESIGN.md."""
    out = {}
    for k, v in sd.items():
        if k.endswith("._packed_params._packed_params"):
            n = k[: -len("._packed_params._packed_params")]
            qw, b = v
            out[n + ".weight"] = qw.dequantize().float()
            out[n + ".bias"] = (torch.zeros(qw.shape[0]) if b is None else b.detach().float())
        elif k.endswith(("._packed_params.dtype", ".scale", ".zero_point")) and (
                k.rsplit(".", 1)[0] + "._packed_params._packed_params" in sd or k.endswith("._packed_params.dtype")):
            continue
        else:
            out[k] = v
    return out


def save(tf, base_name, root="."):
    """The reference's checkpoint directory (transformer_ilqr.py:213-255): tf_model.pt (state dict; fp16 tensors when
    quant_mode == "float16") + tf_model_normalizer.npz (normaliser and hyper-parameters).  Returns the directory."""
    if tf._w is None:
        raise RuntimeError("nothing to save: fit() or load() first")
    sd = {k: torch.as_tensor(v) for k, v in tf._w.items()}
    n_params = sum(v.numel() for k, v in sd.items() if k != "pos_encoder.pe")
    param_str = f"{n_params / 1e6:.1f}M" if n_params >= 1e6 else (f"{n_params / 1e3:.1f}k" if n_params >= 1e3 else str(n_params))
    if tf.quant_mode == "float16":
        sd = {k: v.half() for k, v in sd.items()}
    elif tf.quant_mode == "int8":
        sd = int8_state_dict(sd)
    epochs = getattr(tf, "num_epochs", 0)
    hyper = (f"decoder_dec{tf.num_decoder_layers}_dmodel{tf.d_model}_nhead{tf.nhead}_ff{tf.dim_feedforward}"
             f"_drop{tf.dropout}_epoch{epochs}_promptlen{tf.prompt_len}")
    stamp = datetime.datetime.now().strftime("%Y%m%d_%H%M")
    path = os.path.join(root, f"{stamp}_{base_name}_{hyper}_{param_str}")
    os.makedirs(path, exist_ok=True)
    torch.save(sd, os.path.join(path, "tf_model.pt"))
    np.savez(os.path.join(path, "tf_model_normalizer.npz"),
             x_mean=tf._norm["x_mean"], x_std=tf._norm["x_std"], u_mean=tf._norm["u_mean"], u_std=tf._norm["u_std"],
             target_len=tf.target_len, prompt_len=tf.prompt_len, state_dim=tf.state_dim, control_dim=tf.control_dim,
             d_model=tf.d_model, nhead=tf.nhead, num_decoder_layers=tf.num_decoder_layers,
             dim_feedforward=tf.dim_feedforward, dropout=tf.dropout, max_seq_len=tf.max_seq_len, num_epochs=epochs,
             quant_mode=tf.quant_mode)
    return path
