"""Transformer gain predictor on the GPU: host mirror of the reference's TransformerILQR wrapper (load / predict).

Reference mirrored: quattro_ilqr_tf/transformer_ilqr.py (class TransformerILQR :24, load :259-304, predict :311-325)
around quattro_ilqr_tf/transformer_model.py (TransformerPredictor :85-138, PositionalEncoding :55-80,
DataNormalizer :15-50).  Training (`fit`, `save`, `_create_dataset`) lives in training.py / datagen.py and is exposed
here with the reference's method names.

The forward itself is one HIP kernel (csrc/tf_stream.hip, bf16 MFMA, fp32 accumulation) behind
`quattro_tf_forward_bf16` / `quattro_tf_gains_bf16`; this file stages the weights on the device, has the library pack them
into the kernel's fragment stream (`quattro_tf_pack_stream_bf16`, once per set of weights) and checks shapes.  Checkpoints are read with
loaders that execute nothing from the file (`torch.load(weights_only=True)`, `numpy.load` without pickle).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import check

_HP_KEYS = ("target_len", "prompt_len", "state_dim", "control_dim", "d_model", "nhead", "num_decoder_layers",
            "dim_feedforward", "dropout", "max_seq_len")


class TransformerILQR:
    """Same constructor arguments and `load` / `predict` / `prompt_len` surface as the reference's TransformerILQR;
    adds `predict_batch` (device tensors in, device tensor out) for the batched solver."""

    def __init__(self, state_dim, control_dim, prompt_len=10, d_model=64, nhead=8, num_decoder_layers=3,
                 dim_feedforward=128, dropout=0.1, max_seq_len=100, quant_mode="none", device="cuda:0", precision="bf16"):
        self.state_dim, self.control_dim, self.prompt_len = state_dim, control_dim, prompt_len
        self.d_model, self.nhead, self.num_decoder_layers = d_model, nhead, num_decoder_layers
        self.dim_feedforward, self.dropout, self.max_seq_len = dim_feedforward, dropout, max_seq_len
        self.quant_mode = quant_mode
        # operand type of the device kernel's MFMAs: "bf16" (the north star's) or "fp16" (what the shipped checkpoints
        # store and the reference's predict() computes in, transformer_ilqr.py:317-319); not in the reference's signature
        if precision not in ("bf16", "fp16"):
            raise ValueError(f"precision must be 'bf16' or 'fp16' (got {precision!r})")
        self.precision = precision
        self.target_len = None
        self.device = torch.device(device)
        self._w = None            # host fp32 arrays, reference state_dict names
        self._norm = None
        self._dev = None          # device tensors (kept alive: the C struct holds raw pointers)
        self._tok_bias = {}
        self._struct_cache = {}
        self._fp32 = {}           # n_state_tok -> train_hip.HipTrainer (forward only): shapes the fused kernel does not cover

    # ------------------------------------------------------------------------------------------ loading
    def load(self, model_path):
        """`model_path`: a reference checkpoint directory (tf_model.pt + tf_model_normalizer.npz) or one .npz exported
        by tests/golden/make_golden.py (`tf_weights_<model>.npz`)."""
        if os.path.isdir(model_path):
            data = np.load(os.path.join(model_path, "tf_model_normalizer.npz"), allow_pickle=False)
            sd = torch.load(os.path.join(model_path, "tf_model.pt"), map_location="cpu", weights_only=True)
            if any(k.endswith("._packed_params._packed_params") for k in sd):     # quant_mode "int8" (transformer_ilqr.py:296-297)
                from . import training
                sd = training.dequantize_state_dict(sd)
            weights = {k: v.float().numpy() for k, v in sd.items()}
            norm = {k: np.asarray(data[k], dtype=np.float64) for k in ("x_mean", "x_std", "u_mean", "u_std")}
            hp = {k: data[k].item() for k in _HP_KEYS}
            self.quant_mode = str(data["quant_mode"])
            if "num_epochs" in data.files:
                hp["num_epochs"] = data["num_epochs"].item()
        else:
            z = np.load(model_path, allow_pickle=False)
            weights = {k: z[k].astype(np.float32) for k in z.files if not k.startswith(("norm.", "hp."))}
            norm = {k[5:]: z[k].astype(np.float64) for k in z.files if k.startswith("norm.")}
            hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
        return self.load_arrays(weights, norm, hp)

    def load_arrays(self, weights, norm, hp):
        self.target_len, self.prompt_len = int(hp["target_len"]), int(hp["prompt_len"])
        self.state_dim, self.control_dim = int(hp["state_dim"]), int(hp["control_dim"])
        self.d_model, self.nhead = int(hp["d_model"]), int(hp["nhead"])
        self.num_decoder_layers, self.dim_feedforward = int(hp["num_decoder_layers"]), int(hp["dim_feedforward"])
        self.max_seq_len = int(hp["max_seq_len"])
        self.dropout = float(hp.get("dropout", self.dropout))            # restored like transformer_ilqr.py:283-286
        self.num_epochs = int(hp.get("num_epochs", getattr(self, "num_epochs", 0)))
        self._w = {k: np.asarray(v, dtype=np.float32) for k, v in weights.items()}
        self._norm = {k: np.asarray(v, dtype=np.float64) for k, v in norm.items()}
        if self._w["state_embed.weight"].shape != (self.d_model, self.state_dim):
            raise ValueError("state_embed.weight does not match the hyper-parameters")
        self._stage()
        return self

    @classmethod
    def random_init(cls, state_dim, control_dim, prompt_len, target_len, d_model=128, nhead=4, num_decoder_layers=3,
                    dim_feedforward=512, max_seq_len=110, seed=0, device="cuda:0", precision="bf16"):
        """Random weights of the named architecture (synthetic benchmarking: no checkpoint travels to the GPU box)."""
        g = np.random.default_rng(seed)
        d, ff, c = d_model, dim_feedforward, control_dim
        lin = lambda o, i: (g.uniform(-1, 1, (o, i)) / np.sqrt(i)).astype(np.float32)
        vec = lambda o, s=0.02: (s * g.standard_normal(o)).astype(np.float32)
        w = {"target_embedding": (0.02 * g.standard_normal((target_len, d))).astype(np.float32),
             "state_embed.weight": lin(d, state_dim), "state_embed.bias": vec(d),
             "control_embed.weight": lin(d, c), "control_embed.bias": vec(d),
             "output_linear.weight": lin(c, d), "output_linear.bias": vec(c)}
        pos = np.arange(max_seq_len, dtype=np.float32)[:, None]
        div = np.exp(np.arange(0, d, 2, dtype=np.float32) * (-np.log(10000.0) / d))
        pe = np.zeros((max_seq_len, d), dtype=np.float32)
        pe[:, 0::2], pe[:, 1::2] = np.sin(pos * div), np.cos(pos * div)
        w["pos_encoder.pe"] = pe[None]
        for i in range(num_decoder_layers):
            p = f"transformer_decoder.layers.{i}."
            w[p + "self_attn.in_proj_weight"], w[p + "self_attn.in_proj_bias"] = lin(3 * d, d), vec(3 * d)
            w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"] = lin(d, d), vec(d)
            w[p + "linear1.weight"], w[p + "linear1.bias"] = lin(ff, d), vec(ff)
            w[p + "linear2.weight"], w[p + "linear2.bias"] = lin(d, ff), vec(d)
            for nm in ("norm1", "norm2"):
                w[p + nm + ".weight"], w[p + nm + ".bias"] = (1.0 + vec(d, 0.05)), vec(d)
        norm = dict(x_mean=np.zeros(state_dim), x_std=np.ones(state_dim), u_mean=np.zeros(c), u_std=np.ones(c))
        hp = dict(target_len=target_len, prompt_len=prompt_len, state_dim=state_dim, control_dim=c, d_model=d,
                  nhead=nhead, num_decoder_layers=num_decoder_layers, dim_feedforward=ff, dropout=0.0,
                  max_seq_len=max_seq_len)
        return cls(state_dim, c, device=device, precision=precision).load_arrays(w, norm, hp)

    # ------------------------------------------------------------------------------------------ training (reference names)
    def _create_dataset(self, df):
        """transformer_ilqr.py:70-92: DataFrame (or dict of columns) with x_seq, k_seq, K_seq -> (x_data, kK_data)."""
        from . import training
        return training._as_arrays(df, self.prompt_len)

    def fit(self, df, test_df=None, num_epochs=50, batch_size=16, learning_rate=1e-3, patience=5, **kw):
        """transformer_ilqr.py:102-208 on the GPU (backend="hip": the hand-written training step of csrc/tf_train.hip, the
        default there; "torch": autograd); the trained weights are staged for the HIP inference kernel.  `df` may also be
        a datagen.IterationLog or an (x_data, kK_data) pair."""
        from . import training
        return training.fit(self, df, test_df, num_epochs, batch_size, learning_rate, patience, **kw)

    def save(self, base_name, root="."):
        """transformer_ilqr.py:213-255: writes <root>/<timestamp>_<base_name>_<hyper-parameters>_<size>/ and returns it."""
        from . import training
        return training.save(self, base_name, root)

    # ------------------------------------------------------------------------------------------ device staging
    def _stage(self):
        dev, w = self.device, self._w
        f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)
        t16 = torch.float16 if self.precision == "fp16" else torch.bfloat16
        b16 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev).to(t16).contiguous()
        d = {"x_mean": f32(self._norm["x_mean"]), "x_std": f32(self._norm["x_std"]),
             "u_mean": f32(self._norm["u_mean"]), "u_std": f32(self._norm["u_std"]),
             "state_b": f32(w["state_embed.bias"]),
             "ctrl_w": f32(w["control_embed.weight"]), "ctrl_b": f32(w["control_embed.bias"]),
             "b_out": f32(w["output_linear.bias"])}
        if not self.fused_kernel_covers():          # the layer-wise fp32 path reads the fp32 weights themselves (_fp32_forward)
            self._dev = d
            self._tok_bias, self._struct_cache, self._fp32, self._streams = {}, {}, {}, None
            return
        wst = np.zeros((self.d_model, 16), dtype=np.float32)            # one 16-deep MFMA k-step, columns >= n_x zero
        wst[:, : self.state_dim] = w["state_embed.weight"]
        d["w_state"] = b16(wst)
        wout = np.zeros((64, self.d_model), dtype=np.float32)
        wout[: self.control_dim] = w["output_linear.weight"]
        d["w_out"] = b16(wout)
        for i in range(self.num_decoder_layers):
            p = f"transformer_decoder.layers.{i}."
            d[f"w_qkv{i}"], d[f"b_qkv{i}"] = b16(w[p + "self_attn.in_proj_weight"]), f32(w[p + "self_attn.in_proj_bias"])
            d[f"w_o{i}"], d[f"b_o{i}"] = b16(w[p + "self_attn.out_proj.weight"]), f32(w[p + "self_attn.out_proj.bias"])
            d[f"w_1{i}"], d[f"b_1{i}"] = b16(w[p + "linear1.weight"]), f32(w[p + "linear1.bias"])
            d[f"w_2{i}"], d[f"b_2{i}"] = b16(w[p + "linear2.weight"]), f32(w[p + "linear2.bias"])
            d[f"ln1_g{i}"], d[f"ln1_b{i}"] = f32(w[p + "norm1.weight"]), f32(w[p + "norm1.bias"])
            d[f"ln2_g{i}"], d[f"ln2_b{i}"] = f32(w[p + "norm2.weight"]), f32(w[p + "norm2.bias"])
        self._dev = d
        self._tok_bias = {}
        self._struct_cache = {}
        self._fp32 = {}
        self._streams = None          # (w_stream, p_stream): built by the library from the arrays above on first use

    def shifted_mean(self, x_shift, out=None):
        """x_mean + x_shift as an fp32 device tensor (n,): the normalisation mean that lets the kernel be fed raw states
        x instead of x_err = x - x_shift (x_shift = x_ref - state_offset, quattro_ilqr_tf.py:504).  With `out` the value
        is copied into that (fixed-address) tensor — what a solver that replays a captured graph needs."""
        v = (self._norm["x_mean"] + np.asarray(x_shift, dtype=np.float64).reshape(-1)).astype(np.float32)
        t = torch.as_tensor(v, device=self.device)
        if out is None:
            return t
        out.copy_(t)
        return out

    def prepare(self, n_state_tok, x_mean=None):
        """Build (and cache) everything a forward over `n_state_tok` state tokens needs — the token-bias table upload
        and the C struct — so that no host-to-device copy happens later inside a stream capture."""
        if not self.fused_kernel_covers():
            raise NotImplementedError("graph capture (use_graph=True) needs the fused predictor kernel (d_model 128, 4 heads); "
                                      "this predictor runs through the layer-wise fp32 kernels, eagerly")
        self._struct(n_state_tok, x_mean=x_mean)

    def _struct(self, n_state_tok, x_shift=None, x_mean=None):
        """C struct for sequences with `n_state_tok` state tokens (= horizon + 1); cached.
        x_mean  : device tensor (n,) to normalise with instead of the checkpoint's mean (see shifted_mean); the struct
                  holds its ADDRESS, so the owner may rewrite its contents between launches (graph replays included).
        x_shift : (n,) NumPy convenience form of the same for one-shot calls: a shifted mean is created and cached by
                  value (a handful of entries at most; the oldest is dropped)."""
        if x_mean is not None:
            if x_shift is not None:
                raise ValueError("pass x_shift or x_mean, not both")
            if (not x_mean.is_cuda or x_mean.dtype != torch.float32 or tuple(x_mean.shape) != (self.state_dim,)
                    or not x_mean.is_contiguous()):
                raise ValueError(f"x_mean must be a contiguous fp32 GPU tensor of shape ({self.state_dim},)")
            key = (n_state_tok, "ptr", x_mean.data_ptr())
        else:
            key = (n_state_tok, None if x_shift is None else np.asarray(x_shift, dtype=np.float32).tobytes())
        hit = self._struct_cache.get(key)
        if hit is not None:
            return hit[0]
        s = self._build_struct(n_state_tok)
        keep = x_mean
        if x_shift is not None:
            keep = self.shifted_mean(x_shift)
        if keep is not None:
            s.x_mean = keep.data_ptr()
        while len(self._struct_cache) >= 16:             # bounded: one entry per (horizon, owner buffer / shift value)
            self._struct_cache.pop(next(iter(self._struct_cache)))
        self._struct_cache[key] = (s, keep)              # `keep` holds the mean tensor alive
        return s

    def _build_struct(self, n_state_tok):
        L = n_state_tok + self.prompt_len + self.target_len
        if L > self.max_seq_len:
            raise IndexError(f"sequence of {L} tokens exceeds max_seq_len {self.max_seq_len} of the positional encoding")
        if L > 128:
            raise NotImplementedError(f"sequences of {L} > 128 tokens have no device kernel")
        if n_state_tok not in self._tok_bias:
            # positional encoding + what every token of a kind adds regardless of its input: the embedding bias of
            # state / prompt tokens, the target embedding (transformer_model.py:125-131); transposed [d][128] and
            # zero-padded, the layout the kernel's token-per-lane tiles read with full-width loads
            tb = self._w["pos_encoder.pe"][0, :L].astype(np.float32).copy()
            tb[:n_state_tok] += self._w["state_embed.bias"]
            tb[n_state_tok:n_state_tok + self.prompt_len] += self._w["control_embed.bias"]
            tb[L - self.target_len:] += self._w["target_embedding"]
            tbt = np.zeros((self.d_model, 128), dtype=np.float32)
            tbt[:, :L] = tb.T
            self._tok_bias[n_state_tok] = torch.as_tensor(tbt, device=self.device).contiguous()
        s = _lib.TfWeights()
        s.n_x, s.c_dim, s.d_model, s.n_head = self.state_dim, self.control_dim, self.d_model, self.nhead
        s.d_ff, s.n_layers = self.dim_feedforward, self.num_decoder_layers
        s.n_state_tok, s.prompt_len, s.target_len = n_state_tok, self.prompt_len, self.target_len
        s.precision = _lib.TF_PRECISION_F16 if self.precision == "fp16" else _lib.TF_PRECISION_BF16
        d = self._dev
        for name in ("x_mean", "x_std", "u_mean", "u_std", "w_state", "state_b", "ctrl_w", "ctrl_b", "w_out", "b_out"):
            setattr(s, name, d[name].data_ptr())
        s.tok_bias_t = self._tok_bias[n_state_tok].data_ptr()
        for i in range(self.num_decoder_layers):
            for name in ("w_qkv", "b_qkv", "w_o", "b_o", "w_1", "b_1", "w_2", "b_2", "ln1_g", "ln1_b", "ln2_g", "ln2_b"):
                getattr(s, name)[i] = d[f"{name}{i}"].data_ptr()
        if self._streams is None:
            lib = _lib.load()
            ne, nf = lib.quattro_tf_stream_elems(ctypes.byref(s)), lib.quattro_tf_param_floats(ctypes.byref(s))
            if ne == 0 or nf == 0:
                raise NotImplementedError(
                    f"no device kernel for this predictor shape (d_model {self.d_model}, nhead {self.nhead}, "
                    f"dim_feedforward {self.dim_feedforward}, control_dim {self.control_dim}): supported are d_model 128, "
                    "4 heads, dim_feedforward a multiple of 64 up to 1024, control_dim <= 64")
            ws = torch.empty((ne,), dtype=torch.bfloat16, device=self.device)
            ps = torch.empty((nf,), dtype=torch.float32, device=self.device)
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            check(self._entry("pack_stream")(ctypes.byref(s), ctypes.c_void_p(ws.data_ptr()),
                                             ctypes.c_void_p(ps.data_ptr()), stream), "quattro_tf_pack_stream")
            self._streams = (ws, ps)
        s.w_stream, s.p_stream = self._streams[0].data_ptr(), self._streams[1].data_ptr()
        return s

    # ------------------------------------------------------------------------------------------ shapes beyond the fused kernel
    def fused_kernel_covers(self):
        """The one-launch bf16 / fp16 kernel (csrc/tf_stream.hip) is built for d_model 128, 4 heads (both shipped
        checkpoints); other predictors the reference can construct — its default is d_model 64, 8 heads — run their
        forward through the layer-wise fp32 kernels of the training step (csrc/tf_train.hip: MFMA GEMMs, MFMA attention,
        LayerNorm), still on the device, several launches instead of one, no graph capture."""
        return (self.d_model == 128 and self.nhead == 4 and self.dim_feedforward % 64 == 0 and 64 <= self.dim_feedforward <= 1024
                and self.control_dim <= 64 and self.state_dim <= _lib.MAX_NX)

    def _fp32_forward(self, n_state_tok):
        hit = self._fp32.get(n_state_tok)
        if hit is not None:
            return hit
        from . import train_hip
        shape = (self.state_dim, self.control_dim, self.d_model, self.nhead, self.num_decoder_layers, self.dim_feedforward,
                 n_state_tok, self.prompt_len, self.target_len)
        L = n_state_tok + self.prompt_len + self.target_len
        if L > self.max_seq_len:
            raise IndexError(f"sequence of {L} tokens exceeds max_seq_len {self.max_seq_len} of the positional encoding")
        if not train_hip.supported(*shape):
            raise NotImplementedError(
                f"no device kernel for this predictor shape (d_model {self.d_model}, nhead {self.nhead}, {L} tokens): the fused "
                "kernel takes d_model 128 / 4 heads, the layer-wise fp32 path head dimensions up to 32, d_model up to 512 and "
                "at most 128 tokens")
        tr = train_hip.HipTrainer(*shape, 0.0, self._w["pos_encoder.pe"], self.device)
        tr.load_state_dict({k: torch.as_tensor(v) for k, v in self._w.items() if k in tr.shapes})
        while len(self._fp32) >= 4:
            self._fp32.pop(next(iter(self._fp32)))
        self._fp32[n_state_tok] = tr
        return tr

    def _predict_fp32(self, x, prompt, x_mean=None):
        """(B, N+1, n) raw states or state errors, (B, P, c) raw prompt -> (B, T, c) de-normalised prediction, fp32."""
        d = self._dev
        mean = d["x_mean"] if x_mean is None else x_mean
        xn = ((x - mean) / d["x_std"]).contiguous()
        un = ((prompt - d["u_mean"]) / d["u_std"]).contiguous()
        _, pred = self._fp32_forward(int(x.shape[1])).evaluate(xn, un)
        return pred * d["u_std"] + d["u_mean"]

    def _entry(self, what):
        """quattro_tf_<what>_bf16 or quattro_tf_<what>_f16, by self.precision."""
        return getattr(_lib.load(), f"quattro_tf_{what}_{'f16' if self.precision == 'fp16' else 'bf16'}")

    # ------------------------------------------------------------------------------------------ inference
    def predict_batch(self, x_err, prompt):
        """x_err (B, N+1, n) and prompt (B, P, c): fp32 device tensors (raw, un-normalised) -> (B, T, c) fp32 device tensor."""
        if self._dev is None:
            raise RuntimeError("no weights loaded: call load() / load_arrays() first")
        if x_err.dim() != 3 or x_err.shape[2] != self.state_dim:
            raise ValueError(f"x_err must be (B, N+1, {self.state_dim})")
        B = x_err.shape[0]
        if tuple(prompt.shape) != (B, self.prompt_len, self.control_dim):
            raise ValueError(f"prompt must be ({B}, {self.prompt_len}, {self.control_dim}), got {tuple(prompt.shape)}")
        for t, nm in ((x_err, "x_err"), (prompt, "prompt")):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError(f"{nm} must be a contiguous fp32 GPU tensor")
        if not self.fused_kernel_covers():
            return self._predict_fp32(x_err, prompt)
        s = self._struct(int(x_err.shape[1]))
        pred = torch.empty((B, self.target_len, self.control_dim), dtype=torch.float32, device=x_err.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._entry("forward")(ctypes.byref(s), ctypes.c_void_p(x_err.data_ptr()),
                                     ctypes.c_void_p(prompt.data_ptr()), B,
                                     ctypes.c_void_p(pred.data_ptr()), stream), "quattro_tf_forward")
        return pred

    def predict_gains(self, x_err, prompt, K, k, active=None, x_shift=None, x_mean=None):
        """Like predict_batch, but the prediction is unpacked by the kernel straight into the gain stacks K (B,N,m,n) and
        k (B,N,m) (rows t < min(T, N)); trajectories with active[b] == 0 are left untouched.  prompt=None: the prompt is
        read from the last P rows of K, k themselves (the swept tail).  With x_shift (n,) NumPy, or
        x_mean = shifted_mean(x_shift) as a device tensor, the first argument is the raw state sequence x and the kernel
        forms x - x_shift itself (see _struct)."""
        if self._dev is None:
            raise RuntimeError("no weights loaded: call load() / load_arrays() first")
        B, N, m, n = K.shape
        if self.control_dim != m * (1 + n):
            raise ValueError(f"control_dim {self.control_dim} is not m (1 + n) for gains of shape ({m}, {n})")
        if prompt is None:
            # the prompt rows [k | K.flat] are read by the kernel from rows N - P .. N - 1 of K, k (the swept tail, written
            # there by ops.linearize_sweep(in_place=True)): no packed prompt array at all
            if not self.fused_kernel_covers() or self.target_len + self.prompt_len > N:
                P = self.prompt_len
                prompt = torch.cat([k[:, N - P:], K[:, N - P:].reshape(B, P, -1)], dim=-1).contiguous()
        if tuple(x_err.shape[:1]) != (B,) or x_err.shape[2] != self.state_dim or \
                (prompt is not None and tuple(prompt.shape) != (B, self.prompt_len, self.control_dim)):
            raise ValueError("x_err / prompt do not match the gain stacks")
        for t, nm in ((x_err, "x_err"), (prompt, "prompt"), (K, "K"), (k, "k")):
            if t is None:
                continue
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError(f"{nm} must be a contiguous fp32 GPU tensor")
        if tuple(k.shape) != (B, N, m):
            raise ValueError("k must be (B, N, m)")
        if active is not None and (active.dtype != torch.int32 or tuple(active.shape) != (B,) or not active.is_cuda):
            raise ValueError("active must be an int32 GPU tensor of shape (B,)")
        if not self.fused_kernel_covers():
            if x_shift is not None and x_mean is not None:
                raise ValueError("pass x_shift or x_mean, not both")
            if x_shift is not None:
                x_mean = self.shifted_mean(x_shift)
            rows = self._predict_fp32(x_err, prompt, x_mean).view(B, self.target_len, m, 1 + n)[:, :min(self.target_len, N)]
            Tn = rows.shape[1]
            if active is None:
                k[:, :Tn] = rows[..., 0]
                K[:, :Tn] = rows[..., 1:]
            else:
                live = active != 0
                k[:, :Tn] = torch.where(live[:, None, None], rows[..., 0], k[:, :Tn])
                K[:, :Tn] = torch.where(live[:, None, None, None], rows[..., 1:], K[:, :Tn])
            return
        s = self._struct(int(x_err.shape[1]), x_shift, x_mean)
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = ctypes.c_void_p
        check(self._entry("gains")(ctypes.byref(s), P(x_err.data_ptr()), P(prompt.data_ptr()) if prompt is not None else None, B, N, n, m,
                                   P(K.data_ptr()), P(k.data_ptr()),
                                   P(active.data_ptr()) if active is not None else None, stream),
              "quattro_tf_gains")

    def predict(self, x_seq, kK_seq):
        """Reference signature: x_seq (N+1, n), kK_seq (>= P, c) NumPy -> (T, c) NumPy; the prompt is the last P rows."""
        x = torch.as_tensor(np.ascontiguousarray(np.asarray(x_seq, dtype=np.float32)[None]), device=self.device)
        pr = np.asarray(kK_seq, dtype=np.float32)[-self.prompt_len:, :]
        p = torch.as_tensor(np.ascontiguousarray(pr[None]), device=self.device)
        return self.predict_batch(x, p)[0].double().cpu().numpy()
