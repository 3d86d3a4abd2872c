"""Device training step of the gain predictor over the C ABI (`quattro_tf_train_step_f32`, `quattro_tf_adam_f32`).

What `training.fit(..., backend="hip")` runs per mini-batch instead of torch autograd: the hand-written forward with
saved activations, the MSE loss, the backward and the Adam update of csrc/tf_train.hip.  torch is plumbing here — it
owns the flat parameter / gradient / moment arrays and the workspace.  Mirrors the loop body of
quattro_ilqr_tf/transformer_ilqr.py:150-172.

The flat parameter array holds every block of the reference module's state dict in its own shape and layout
(`state_dict()` / `load_state_dict()` slice it by the reference names), so weights trained here go straight into
checkpoints, the HIP inference kernel and the torch restatement in `training.forward`.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check
from .ops import _ptr, _stream


def _desc(state_dim, control_dim, d_model, nhead, n_layers, d_ff, n_state_tok, prompt_len, target_len, dropout):
    d = _lib.TfTrainDesc()
    d.state_dim, d.control_dim, d.d_model, d.nhead, d.n_layers, d.d_ff = state_dim, control_dim, d_model, nhead, n_layers, d_ff
    d.n_state_tok, d.prompt_len, d.target_len, d.dropout = n_state_tok, prompt_len, target_len, float(dropout)
    return d


def supported(state_dim, control_dim, d_model, nhead, n_layers, d_ff, n_state_tok, prompt_len, target_len, dropout=0.0):
    """True when the device training step has kernels for this shape (head dimension <= 32, d_model <= 512, sequence of
    at most 128 tokens)."""
    d = _desc(state_dim, control_dim, d_model, nhead, n_layers, d_ff, n_state_tok, prompt_len, target_len, dropout)
    return _lib.load().quattro_tf_train_param_count(ctypes.byref(d)) > 0


class HipTrainer:
    """Flat fp32 parameters + Adam state on `device`, and the per-mini-batch step."""

    def __init__(self, state_dim, control_dim, d_model, nhead, n_layers, d_ff, n_state_tok, prompt_len, target_len,
                 dropout, pe, device, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.lib = _lib.load()
        self.desc = _desc(state_dim, control_dim, d_model, nhead, n_layers, d_ff, n_state_tok, prompt_len, target_len,
                          dropout)
        self.n_params = int(self.lib.quattro_tf_train_param_count(ctypes.byref(self.desc)))
        if self.n_params == 0:
            raise NotImplementedError("quattro_tf_train_step_f32 has no kernels for this predictor shape")
        self.device = torch.device(device)
        self.L = n_state_tok + prompt_len + target_len
        self.shapes = self._shapes()
        self.offsets = {}
        for name, (which, layer, shape) in self.shapes.items():
            off = int(self.lib.quattro_tf_train_param_offset(ctypes.byref(self.desc), which, layer))
            if off < 0:
                raise RuntimeError(f"no offset for {name}")
            self.offsets[name] = off
        z = lambda: torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
        self.params, self.grads, self.m, self.v = z(), z(), z(), z()
        pe = torch.as_tensor(np.asarray(pe, dtype=np.float32)).reshape(-1, d_model)
        if pe.shape[0] < self.L:
            raise ValueError("positional table shorter than the sequence")
        self.pe = pe[: self.L].contiguous().to(self.device)
        self.lr, self.betas, self.eps, self.t = float(lr), betas, float(eps), 0
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._ws, self._ws_batch = None, 0

    def _shapes(self):
        D = self.desc
        d, ff, c, n, T = D.d_model, D.d_ff, D.control_dim, D.state_dim, D.target_len
        glob = [(T, d), (d, n), (d,), (d, c), (d,), (c, d), (c,)]
        lay = [(3 * d, d), (3 * d,), (d, d), (d,), (ff, d), (ff,), (d, ff), (d,), (d,), (d,), (d,), (d,)]
        out = {}
        for i, (nm, sh) in enumerate(zip(_lib.TF_TRAIN_GLOBAL, glob)):
            out[nm] = (i, 0, sh)
        for l in range(D.n_layers):
            for i, (nm, sh) in enumerate(zip(_lib.TF_TRAIN_LAYER, lay)):
                out[f"transformer_decoder.layers.{l}.{nm}"] = (len(glob) + i, l, sh)
        return out

    # ------------------------------------------------------------------ state dict <-> flat array
    def view(self, flat, name):
        _, _, shape = self.shapes[name]
        off = self.offsets[name]
        return flat[off: off + int(np.prod(shape))].view(*shape)

    def load_state_dict(self, sd):
        for name in self.shapes:
            self.view(self.params, name).copy_(torch.as_tensor(sd[name], dtype=torch.float32).detach().to(self.device))
        return self

    def state_dict(self, flat=None):
        flat = self.params if flat is None else flat
        return {name: self.view(flat, name).clone() for name in self.shapes}

    # ------------------------------------------------------------------ steps
    def _workspace(self, batch):
        if self._ws is None or batch > self._ws_batch:
            nbytes = int(self.lib.quattro_tf_train_workspace_bytes(ctypes.byref(self.desc), batch))
            if nbytes == 0:
                raise NotImplementedError("quattro_tf_train_workspace_bytes: unsupported shape")
            self._ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
        base = self._ws.data_ptr()
        aligned = (base + 255) // 256 * 256
        return ctypes.c_void_p(aligned), self._ws.numel() - (aligned - base)

    def _check_batch(self, x_norm, prompt_norm, target_norm):
        D = self.desc
        B = x_norm.shape[0]
        for t, shape, nm in ((x_norm, (B, D.n_state_tok, D.state_dim), "x_norm"),
                             (prompt_norm, (B, D.prompt_len, D.control_dim), "prompt_norm"),
                             (target_norm, (B, D.target_len, D.control_dim), "target_norm")):
            if t is None:
                continue
            if tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise ValueError(f"{nm} must be a contiguous float32 GPU tensor of shape {shape} (got {tuple(t.shape)}, "
                                 f"{t.dtype}, {t.device})")
        return B

    def forward_backward(self, x_norm, prompt_norm, target_norm, seed=0, training=True, want_pred=False):
        """Loss and gradients of one mini-batch (gradients land in self.grads).  Returns the loss as a device scalar
        tensor (and the normalised prediction when asked)."""
        B = self._check_batch(x_norm, prompt_norm, target_norm)
        ws, ws_bytes = self._workspace(B)
        pred = (torch.empty((B, self.desc.target_len, self.desc.control_dim), dtype=torch.float32, device=self.device)
                if want_pred else None)
        check(self.lib.quattro_tf_train_step_f32(ctypes.byref(self.desc), _ptr(self.params), _ptr(self.grads), ws, ws_bytes,
                                                 _ptr(x_norm), _ptr(prompt_norm), _ptr(target_norm), _ptr(self.pe), B,
                                                 ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), 1 if training else 0,
                                                 _ptr(self.loss), _ptr(pred), _stream()), "quattro_tf_train_step_f32")
        return (self.loss, pred) if want_pred else self.loss

    EVAL_CHUNK = 1024      # sequences per forward-only call: bounds the workspace (~2.8 MB per sequence of the shipped predictor:
                           # saved activations + gradient temporaries are carved even when no gradient is asked for) and keeps
                           # the attention grids (dim3(H, batch)) inside gridDim.y <= 65535

    def evaluate(self, x_norm, prompt_norm, target_norm=None):
        """Forward without dropout: (loss or None, normalised prediction).  Any batch size: evaluated in chunks of
        EVAL_CHUNK sequences, the loss as the sequence-weighted mean of the chunks' means (= the mean over the whole set)."""
        B = self._check_batch(x_norm, prompt_norm, target_norm)
        pred = torch.empty((B, self.desc.target_len, self.desc.control_dim), dtype=torch.float32, device=self.device)
        total = None
        for lo in range(0, B, self.EVAL_CHUNK):
            hi = min(B, lo + self.EVAL_CHUNK)
            ws, ws_bytes = self._workspace(hi - lo)
            tgt = None if target_norm is None else target_norm[lo:hi]
            check(self.lib.quattro_tf_train_step_f32(ctypes.byref(self.desc), _ptr(self.params), None, ws, ws_bytes,
                                                     _ptr(x_norm[lo:hi]), _ptr(prompt_norm[lo:hi]), _ptr(tgt), _ptr(self.pe),
                                                     hi - lo, ctypes.c_uint64(0), 0, _ptr(self.loss) if tgt is not None else None,
                                                     _ptr(pred[lo:hi]), _stream()), "quattro_tf_train_step_f32")
            if tgt is not None:
                part = self.loss * ((hi - lo) / B)
                total = part if total is None else total + part
        if total is not None and B > self.EVAL_CHUNK:
            return total, pred
        return (self.loss if target_norm is not None else None), pred

    def adam_step(self):
        self.t += 1
        check(self.lib.quattro_tf_adam_f32(_ptr(self.params), _ptr(self.grads), _ptr(self.m), _ptr(self.v), self.n_params,
                                           self.lr, self.betas[0], self.betas[1], self.eps, self.t, _stream()),
              "quattro_tf_adam_f32")

    def dropout_mask(self, seed, p, site, n):
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        check(self.lib.quattro_tf_train_dropout_mask_f32(ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), float(p), int(site), n,
                                                         _ptr(out), _stream()), "quattro_tf_train_dropout_mask_f32")
        return out
