"""Tensor-level operators over the C ABI.  torch is plumbing here: it owns device memory and the stream; every
computation is a kernel of libquattro_hip.so.  All tensors must be contiguous CUDA(=HIP) tensors on one device.

Shape checks happen here, on the host, before anything is launched: a hand-written kernel that is handed a
buffer smaller than its grid assumes faults the GPU.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check

ALPHAS = (1.0, 0.5, 0.25, 0.1, 0.05, 0.01)      # reference line search, quattro_ilqr_tf.py:440,552
QUU_REG = 1e-6                                    # quattro_ilqr_tf.py:304


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _req(t, shape, dtype, name):
    if t is None:
        raise ValueError(f"{name} is required")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device})")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype} (got {t.dtype})")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)} (got {tuple(t.shape)})")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _alphas(alphas):
    if not 0 < len(alphas) <= _lib.MAX_ALPHAS:
        raise ValueError(f"1..{_lib.MAX_ALPHAS} line-search step sizes supported")
    return (ctypes.c_float * len(alphas))(*alphas), len(alphas)


def record_stride(n, m, layout, lib=None):
    s = (lib or _lib.load()).quattro_record_stride(n, m, layout)
    if s == 0:
        raise NotImplementedError(f"no record layout {layout} for (n, m) = ({n}, {m})")
    return s


def record_header(n, m, layout, lib=None):
    return (lib or _lib.load()).quattro_record_header(n, m, layout)


def preferred_layout(n, m, lib=None):
    return (lib or _lib.load()).quattro_preferred_layout(n, m)


def model_layout(model):
    """The layout linearize + sweep are fastest with for this model (TILE16C for the Euler quadrotor: constants of the
    problem live once in a header record, 304 B per step through HBM instead of 1,664 B)."""
    p = model.c_params()
    lay = _lib.load_for(model).quattro_model_layout(ctypes.byref(p))
    if lay < 0:
        raise NotImplementedError(f"no device kernels for model {model.name}")
    return lay


def alloc_records(n, m, layout, B, S, device, lib=None):
    """Record buffer for B x S steps: (B, S, stride) for the plain layouts, a flat header + B*S*stride buffer for TILE16C.
    `lib` (here and below): the library of a user model (its (n, m) exist only there); None = libquattro_hip.so."""
    stride, header = record_stride(n, m, layout, lib), record_header(n, m, layout, lib)
    if header == 0:
        return torch.empty((B, S, stride), dtype=torch.float32, device=device)
    return torch.empty((header + B * S * stride,), dtype=torch.float32, device=device)


def _check_records(rec, n, m, layout, B, S=None, lib=None):
    """-> S.  Shape / size check of a record buffer against (B, S) for either kind of layout."""
    stride, header = record_stride(n, m, layout, lib), record_header(n, m, layout, lib)
    if rec.dtype != torch.float32 or not rec.is_contiguous() or rec.device.type != "cuda":
        raise ValueError("rec must be a contiguous float32 tensor on the GPU")
    body = rec.numel() - header
    if S is None:
        S = body // (B * stride) if B > 0 else 0
    if S <= 0 or body != B * S * stride:
        raise ValueError(f"rec holds {rec.numel()} floats, expected {header} + {B} x {S} x {stride}")
    return S


def pack_derivs(A, Bm, lx, lu, lxx, luu, lux, layout=None, lib=None):
    """Separate row-major blocks (B, S, ...) -> records (B, S, stride)."""
    Bt, S, n, _ = A.shape
    m = Bm.shape[3]
    layout = preferred_layout(n, m, lib) if layout is None else layout
    stride = record_stride(n, m, layout, lib)
    f32 = torch.float32
    _req(A, (Bt, S, n, n), f32, "A"); _req(Bm, (Bt, S, n, m), f32, "B"); _req(lx, (Bt, S, n), f32, "lx")
    _req(lu, (Bt, S, m), f32, "lu"); _req(lxx, (Bt, S, n, n), f32, "lxx"); _req(luu, (Bt, S, m, m), f32, "luu")
    _req(lux, (Bt, S, m, n), f32, "lux")
    rec = torch.zeros((Bt, S, stride), dtype=f32, device=A.device)
    check((lib or _lib.load()).quattro_pack_derivs_f32(_ptr(A), _ptr(Bm), _ptr(lx), _ptr(lu), _ptr(lxx), _ptr(luu), _ptr(lux),
                                              Bt, S, n, m, layout, _ptr(rec), _stream()), "quattro_pack_derivs_f32")
    return rec, layout


def unpack_derivs(rec, B, n, m, layout, lib=None):
    """Records (any layout) -> dict of row-major blocks A (B,S,n,n), B (B,S,n,m), lx, lu, lxx, luu, lux — the arrays the
    reference's _compute_dynamics_jacobians / _compute_cost_derivatives return, for every (b, t) of the buffer."""
    S = _check_records(rec, n, m, layout, B, lib=lib)
    f32, dev = torch.float32, rec.device
    out = dict(A=torch.empty((B, S, n, n), dtype=f32, device=dev), B=torch.empty((B, S, n, m), dtype=f32, device=dev),
               lx=torch.empty((B, S, n), dtype=f32, device=dev), lu=torch.empty((B, S, m), dtype=f32, device=dev),
               lxx=torch.empty((B, S, n, n), dtype=f32, device=dev), luu=torch.empty((B, S, m, m), dtype=f32, device=dev),
               lux=torch.empty((B, S, m, n), dtype=f32, device=dev))
    check((lib or _lib.load()).quattro_unpack_derivs_f32(_ptr(rec), B, S, n, m, layout, _ptr(out["A"]), _ptr(out["B"]),
                                                _ptr(out["lx"]), _ptr(out["lu"]), _ptr(out["lxx"]), _ptr(out["luu"]),
                                                _ptr(out["lux"]), _stream()), "quattro_unpack_derivs_f32")
    return out


def riccati_sweep(rec, VxN, VxxN, n, m, layout, reg=QUU_REG, K=None, k=None, status=None, active=None, repair=False,
                  lib=None):
    """Backward sweep over the S steps held in `rec`.  Returns K (B,S,m,n), k (B,S,m), status (B,).

    repair=True (one host synchronisation): trajectories the quadrotor-shaped kernel flags TRAJ_ILLCOND — an indefinite or
    badly scaled Q_uu + reg I, which it eliminates without pivoting — are swept again by the generic kernel, which pivots
    like the reference's np.linalg.inv (quattro_ilqr_tf.py:306); their bit is cleared when the second sweep is clean."""
    Bt = VxN.shape[0]
    f32 = torch.float32
    S = _check_records(rec, n, m, layout, Bt, lib=lib)
    _req(VxN, (Bt, n), f32, "VxN"); _req(VxxN, (Bt, n, n), f32, "VxxN")
    K = torch.empty((Bt, S, m, n), dtype=f32, device=rec.device) if K is None else _req(K, (Bt, S, m, n), f32, "K")
    k = torch.empty((Bt, S, m), dtype=f32, device=rec.device) if k is None else _req(k, (Bt, S, m), f32, "k")
    status = (torch.zeros((Bt,), dtype=torch.int32, device=rec.device) if status is None
              else _req(status, (Bt,), torch.int32, "status"))
    if active is not None:
        _req(active, (Bt,), torch.int32, "active")
    # N and t_start only enter the kernel as S = N - t_start
    check((lib or _lib.load()).quattro_riccati_sweep_f32(_ptr(rec), _ptr(VxN), _ptr(VxxN), Bt, S, 0, n, m, layout, reg,
                                                _ptr(K), _ptr(k), _ptr(status), _ptr(active), _stream()),
          "quattro_riccati_sweep_f32")
    if repair and layout == _lib.LAYOUT_ROWMAJOR_TILE:   # the same records through the pivoting kernel: no repacking
        flagged = torch.nonzero((status & _lib.TRAJ_ILLCOND) != 0).reshape(-1)
        if flagged.numel() > 0:
            stride = record_stride(n, m, layout, lib)
            rec2 = rec.reshape(Bt, S, stride).index_select(0, flagged).contiguous()
            K2, k2, st2 = riccati_sweep(rec2, VxN.index_select(0, flagged).contiguous(),
                                        VxxN.index_select(0, flagged).contiguous(), n, m, _lib.LAYOUT_ROWMAJOR, reg, lib=lib)
            K.index_copy_(0, flagged, K2)
            k.index_copy_(0, flagged, k2)
            status.index_copy_(0, flagged, st2)
    if repair and layout == _lib.LAYOUT_TILE16:          # (TILE16C / TILE16R come from the built-in convex cost: never flagged)
        flagged = torch.nonzero((status & _lib.TRAJ_ILLCOND) != 0).reshape(-1)
        if flagged.numel() > 0:
            blocks = unpack_derivs(rec, Bt, n, m, layout)
            sel = {name: t.index_select(0, flagged).contiguous() for name, t in blocks.items()}
            rec2, lay2 = pack_derivs(sel["A"], sel["B"], sel["lx"], sel["lu"], sel["lxx"], sel["luu"], sel["lux"],
                                     layout=_lib.LAYOUT_ROWMAJOR)
            K2, k2, st2 = riccati_sweep(rec2, VxN.index_select(0, flagged).contiguous(),
                                        VxxN.index_select(0, flagged).contiguous(), n, m, lay2, reg)
            K.index_copy_(0, flagged, K2)
            k.index_copy_(0, flagged, k2)
            status.index_copy_(0, flagged, st2)
    return K, k, status


def linearize(model, x, u, t_start=0, layout=None, rec=None, VxN=None, VxxN=None):
    """Records for t in [t_start, N) and the terminal pair, about the nominal (x, u)."""
    Bt, N, m = u.shape
    n = x.shape[2]
    if (n, m) != (model.n, model.m):
        raise ValueError(f"trajectory dims ({n}, {m}) do not match model {model.name} ({model.n}, {model.m})")
    if not 0 <= t_start < N:
        raise ValueError("t_start must be in [0, N)")
    layout = model_layout(model) if layout is None else layout
    f32 = torch.float32
    _req(x, (Bt, N + 1, n), f32, "x"); _req(u, (Bt, N, m), f32, "u")
    S = N - t_start
    lib = _lib.load_for(model)
    if rec is None:
        rec = alloc_records(n, m, layout, Bt, S, x.device, lib)
    else:
        _check_records(rec, n, m, layout, Bt, S, lib)
    VxN = torch.empty((Bt, n), dtype=f32, device=x.device) if VxN is None else _req(VxN, (Bt, n), f32, "VxN")
    VxxN = torch.empty((Bt, n, n), dtype=f32, device=x.device) if VxxN is None else _req(VxxN, (Bt, n, n), f32, "VxxN")
    p = model.c_params()
    check(lib.quattro_linearize_f32(ctypes.byref(p), _ptr(x), _ptr(u), Bt, N, t_start, layout, _ptr(rec),
                                            _ptr(VxN), _ptr(VxxN), None, _stream()), "quattro_linearize_f32")
    return rec, VxN, VxxN, layout


def model_fuses_sweep(model):
    """True where linearize_sweep() — the sweep kernel linearising its own trajectory — is the model's fastest backward pass."""
    p = model.c_params()
    return _lib.load_for(model).quattro_model_fuses_sweep(ctypes.byref(p)) == 1


def model_can_fuse_sweep(model):
    """True where linearize_sweep() works at all (also the RK4 quadrotor, whose record path is faster as separate launches)."""
    p = model.c_params()
    return _lib.load_for(model).quattro_model_fuses_sweep(ctypes.byref(p)) > 0


def linearize_sweep_scratch_bytes(model, B, N, t_start=0):
    """Device scratch linearize_sweep needs for this model (0 except for the RK4 quadrotor: 528 B per step)."""
    p = model.c_params()
    return int(_lib.load_for(model).quattro_linearize_sweep_scratch_bytes(ctypes.byref(p), B, N, t_start))


def linearize_sweep(model, x, u, t_start=0, reg=QUU_REG, K=None, k=None, status=None, active=None, scratch=None,
                    in_place=False):
    """Linearisation + backward sweep in one launch, no record buffer (Euler quadrotor: bit-identical to linearize +
    riccati_sweep).  Returns K (B,S,m,n), k (B,S,m), status (B,) for the S = N - t_start steps from t_start.
    in_place=True: K (B,N,m,n) and k (B,N,m) are the FULL gain stacks and the swept steps are written at rows t_start .. N-1
    (quattro_linearize_sweep_rows_f32): what a hybrid iteration wants — the predictor fills the rows below.
    `scratch`: uint8 device buffer of >= linearize_sweep_scratch_bytes(...) where that is not 0 (allocated here if None)."""
    Bt, N, m = u.shape
    n = x.shape[2]
    if (n, m) != (model.n, model.m):
        raise ValueError(f"trajectory dims ({n}, {m}) do not match model {model.name} ({model.n}, {model.m})")
    if not 0 <= t_start < N:
        raise ValueError("t_start must be in [0, N)")
    f32 = torch.float32
    _req(x, (Bt, N + 1, n), f32, "x"); _req(u, (Bt, N, m), f32, "u")
    S = N if in_place else N - t_start
    if in_place and (K is None or k is None):
        raise ValueError("in_place needs the full gain stacks K (B,N,m,n) and k (B,N,m)")
    K = torch.empty((Bt, S, m, n), dtype=f32, device=x.device) if K is None else _req(K, (Bt, S, m, n), f32, "K")
    k = torch.empty((Bt, S, m), dtype=f32, device=x.device) if k is None else _req(k, (Bt, S, m), f32, "k")
    status = (torch.zeros((Bt,), dtype=torch.int32, device=x.device) if status is None
              else _req(status, (Bt,), torch.int32, "status"))
    if active is not None:
        _req(active, (Bt,), torch.int32, "active")
    p = model.c_params()
    need = linearize_sweep_scratch_bytes(model, Bt, N, t_start)
    if need and scratch is None:
        scratch = torch.empty((need,), dtype=torch.uint8, device=x.device)
    nbytes = 0 if scratch is None else scratch.numel() * scratch.element_size()
    check(_lib.load_for(model).quattro_linearize_sweep_rows_f32(ctypes.byref(p), _ptr(x), _ptr(u), Bt, N, t_start, reg, _ptr(K),
                                                                _ptr(k), N if in_place else 0, _ptr(status), _ptr(active),
                                                                _ptr(scratch), nbytes, _stream()),
          "quattro_linearize_sweep_rows_f32")
    return K, k, status


def simulate(model, x0, u, x=None, cost=None):
    """Open-loop rollout from x0 (B,n) under u (B,N,m): x (B,N+1,n), cost (B,) fp64."""
    Bt, N, m = u.shape
    n = model.n
    f32 = torch.float32
    _req(x0, (Bt, n), f32, "x0"); _req(u, (Bt, N, model.m), f32, "u")
    x = torch.empty((Bt, N + 1, n), dtype=f32, device=u.device) if x is None else _req(x, (Bt, N + 1, n), f32, "x")
    cost = torch.empty((Bt,), dtype=torch.float64, device=u.device) if cost is None else _req(cost, (Bt,), torch.float64, "cost")
    p = model.c_params()
    check(_lib.load_for(model).quattro_simulate_f32(ctypes.byref(p), _ptr(x0), _ptr(u), Bt, N, _ptr(x), _ptr(cost), _stream()),
          "quattro_simulate_f32")
    return x, cost


def total_cost(model, x, u):
    """sum_t L(x_t,u_t) + Lf(x_N) for given sequences: cost (B,) fp64."""
    Bt, N, m = u.shape
    f32 = torch.float32
    _req(x, (Bt, N + 1, model.n), f32, "x"); _req(u, (Bt, N, model.m), f32, "u")
    cost = torch.empty((Bt,), dtype=torch.float64, device=u.device)
    p = model.c_params()
    check(_lib.load_for(model).quattro_total_cost_f32(ctypes.byref(p), _ptr(x), _ptr(u), Bt, N, _ptr(cost), _stream()),
          "quattro_total_cost_f32")
    return cost


def rollout(model, x_nom, u_nom, K, k, alphas=ALPHAS, want_traj=False, active=None):
    """Closed-loop forward passes for every alpha: cost (n_alpha, B) fp64 [, x_new (n_alpha,B,N+1,n), u_new]."""
    Bt, N, m = u_nom.shape
    n = model.n
    f32 = torch.float32
    _req(x_nom, (Bt, N + 1, n), f32, "x_nom"); _req(u_nom, (Bt, N, model.m), f32, "u_nom")
    _req(K, (Bt, N, m, n), f32, "K"); _req(k, (Bt, N, m), f32, "k")
    arr, na = _alphas(alphas)
    cost = torch.empty((na, Bt), dtype=torch.float64, device=u_nom.device)
    x_new = torch.empty((na, Bt, N + 1, n), dtype=f32, device=u_nom.device) if want_traj else None
    u_new = torch.empty((na, Bt, N, m), dtype=f32, device=u_nom.device) if want_traj else None
    if active is not None:
        _req(active, (Bt,), torch.int32, "active")
    p = model.c_params()
    check(_lib.load_for(model).quattro_rollout_f32(ctypes.byref(p), _ptr(x_nom), _ptr(u_nom), _ptr(K), _ptr(k), arr, na, Bt, N,
                                          _ptr(x_new), _ptr(u_new), _ptr(cost), _ptr(active), _stream()),
          "quattro_rollout_f32")
    return (cost, x_new, u_new) if want_traj else cost


def linesearch_scratch_bytes(model, B, N):
    return int(_lib.load_for(model).quattro_linesearch_scratch_bytes(model.n, model.m, B, N))


def linesearch_scratch(model, B, N, device):
    """Fresh device scratch for one fused line search (candidate trajectories).  Not cached: torch's caching allocator
    makes the allocation cheap and stream-ordered, and a module-level cache could free a buffer whose address a
    captured graph (or a launch in flight on another stream) still holds.  Long-lived callers (QuattroILQR) own
    their scratch and pass it as `scratch=`."""
    return torch.empty((linesearch_scratch_bytes(model, B, N),), dtype=torch.uint8, device=device)


def linesearch(model, x_nom, u_nom, K, k, cost, tol, alphas=ALPHAS, alpha_idx=None, active=None, iters=None,
               scratch=None):
    """Fused line search; commits the first accepted alpha into x_nom/u_nom/cost IN PLACE.  Returns alpha_idx (B,)."""
    Bt, N, m = u_nom.shape
    n = model.n
    f32 = torch.float32
    _req(x_nom, (Bt, N + 1, n), f32, "x_nom"); _req(u_nom, (Bt, N, model.m), f32, "u_nom")
    _req(K, (Bt, N, m, n), f32, "K"); _req(k, (Bt, N, m), f32, "k"); _req(cost, (Bt,), torch.float64, "cost")
    arr, na = _alphas(alphas)
    alpha_idx = (torch.empty((Bt,), dtype=torch.int32, device=u_nom.device) if alpha_idx is None
                 else _req(alpha_idx, (Bt,), torch.int32, "alpha_idx"))
    if active is not None:
        _req(active, (Bt,), torch.int32, "active")
    if iters is not None:
        _req(iters, (Bt,), torch.int32, "iters")
    if scratch is None:
        scratch = linesearch_scratch(model, Bt, N, u_nom.device)
    p = model.c_params()
    check(_lib.load_for(model).quattro_linesearch_f32(ctypes.byref(p), _ptr(x_nom), _ptr(u_nom), _ptr(K), _ptr(k), arr, na, Bt,
                                             N, float(tol), _ptr(cost), _ptr(alpha_idx), _ptr(active), _ptr(iters),
                                             _ptr(scratch), scratch.numel() * scratch.element_size(), _stream()),
          "quattro_linesearch_f32")
    return alpha_idx


def workspace(model, B, N, device):
    """Device workspace of the fused iteration (records, V_x(N), V_xx(N), candidate trajectories); uint8, 256-aligned
    (torch's caching allocator hands out 512-byte aligned blocks)."""
    p = model.c_params()
    nbytes = _lib.load_for(model).quattro_model_workspace_bytes(ctypes.byref(p), B, N)
    if nbytes == 0:
        raise NotImplementedError(f"no device kernel for n={model.n}, m={model.m}")
    return torch.empty((nbytes,), dtype=torch.uint8, device=device)


def ilqr_iterate(model, x_nom, u_nom, K, k, cost, tol, workspace, alphas=ALPHAS, reg=QUU_REG, alpha_idx=None,
                 active=None, iters=None, status=None):
    """One whole pure-iLQR iteration (linearize + sweep + line search/commit) from ONE C call; everything in place."""
    Bt, N, m = u_nom.shape
    n = model.n
    f32, i32 = torch.float32, torch.int32
    _req(x_nom, (Bt, N + 1, n), f32, "x_nom"); _req(u_nom, (Bt, N, model.m), f32, "u_nom")
    _req(K, (Bt, N, m, n), f32, "K"); _req(k, (Bt, N, m), f32, "k"); _req(cost, (Bt,), torch.float64, "cost")
    _req(alpha_idx, (Bt,), i32, "alpha_idx"); _req(active, (Bt,), i32, "active")
    if iters is not None:
        _req(iters, (Bt,), i32, "iters")
    if status is not None:
        _req(status, (Bt,), i32, "status")
    arr, na = _alphas(alphas)
    p = model.c_params()
    check(_lib.load_for(model).quattro_ilqr_iterate_f32(ctypes.byref(p), _ptr(x_nom), _ptr(u_nom), Bt, N, float(reg), arr, na,
                                               float(tol), _ptr(K), _ptr(k), _ptr(cost), _ptr(alpha_idx), _ptr(active),
                                               _ptr(iters), _ptr(status), _ptr(workspace),
                                               workspace.numel() * workspace.element_size(), _stream()),
          "quattro_ilqr_iterate_f32")
    return alpha_idx


def model_has_device_loop(model):
    """True where the whole solve (and the MPC loop around it) as ONE persistent launch is the model's fastest form
    (csrc/solve_quad.hip, csrc/solve_cartpole.hip)."""
    p = model.c_params()
    return _lib.load_for(model).quattro_model_has_device_loop(ctypes.byref(p)) == 1


def model_can_device_loop(model):
    """True where a persistent kernel exists at all — also a user-compiled model's (csrc/solve_user.hip: one wave per
    trajectory through the generic device bodies; no host involvement, but slower than enqueued iterations)."""
    p = model.c_params()
    return _lib.load_for(model).quattro_model_has_device_loop(ctypes.byref(p)) > 0


class SolveLog:
    """Device-side per-iteration log ring of a solve (include/quattro_hip.h: quattro_solve_log) and its decoding.

    One record per (trajectory, iteration): a 64-byte header (four s_memrealtime stamps, cost before / after, accepted step,
    iteration number) followed by x_seq, u_seq (traj=True) and K, k (gains=True) — the contents of the reference's log dict
    (quattro_ilqr_tf.py:453-466) and the samples of its *_time lists (:16-42).  Iteration i of trajectory b is record
    b * capacity + (i % capacity); `rows(b, count)` downloads and decodes the first `count` of them."""

    TICK = 1e-8            # s_memrealtime counts at 100 MHz

    def __init__(self, model, N, B, capacity, device, traj=True, gains=True):
        self.lib = _lib.load_for(model)
        self.n, self.m, self.N, self.B, self.capacity = model.n, model.m, int(N), int(B), int(capacity)
        self.flags = (_lib.LOG_TRAJ if traj else 0) | (_lib.LOG_GAINS if gains else 0)
        self.rec_bytes = int(self.lib.quattro_solve_log_record_bytes(self.n, self.m, self.N, self.flags))
        off = lambda f: int(self.lib.quattro_solve_log_offset(self.n, self.m, self.N, self.flags, f))
        self.off = dict(x=off(_lib.LOG_FIELD_X), u=off(_lib.LOG_FIELD_U), K=off(_lib.LOG_FIELD_K), k=off(_lib.LOG_FIELD_KFF))
        if self.rec_bytes < _lib.LOG_HEADER_BYTES or self.capacity <= 0:
            raise ValueError("bad log geometry")
        self.buf = torch.zeros((self.B * self.capacity * self.rec_bytes,), dtype=torch.uint8, device=device)
        self.c = _lib.SolveLogC(self.buf.data_ptr(), self.capacity, self.flags)
        self._pin = None

    def byref(self):
        return ctypes.byref(self.c)

    def decode(self, raw, count):
        """raw: uint8 array of `count` consecutive records -> dict of arrays (first axis = record)."""
        rec = np.asarray(raw, dtype=np.uint8).reshape(count, self.rec_bytes)
        n, m, N = self.n, self.m, self.N
        take = lambda a, b, dt: np.ascontiguousarray(rec[:, a:b]).view(dt)
        out = dict(stamps=take(0, 32, np.uint64), cost=take(32, 48, np.float64), alpha_idx=take(48, 52, np.int32)[:, 0],
                   iteration=take(52, 56, np.int32)[:, 0])
        if self.flags & _lib.LOG_TRAJ:
            out["x"] = take(self.off["x"], self.off["x"] + 4 * (N + 1) * n, np.float32).reshape(count, N + 1, n)
            out["u"] = take(self.off["u"], self.off["u"] + 4 * N * m, np.float32).reshape(count, N, m)
        if self.flags & _lib.LOG_GAINS:
            out["K"] = take(self.off["K"], self.off["K"] + 4 * N * m * n, np.float32).reshape(count, N, m, n)
            out["k"] = take(self.off["k"], self.off["k"] + 4 * N * m, np.float32).reshape(count, N, m)
        return out

    def stage(self, b, count):
        """Enqueue the download of the first `count` records of trajectory b into a pinned buffer (no synchronisation)."""
        count = min(int(count), self.capacity)
        if self._pin is None or self._pin.shape[0] < count * self.rec_bytes:
            self._pin = torch.empty((max(count, 4) * self.rec_bytes,), dtype=torch.uint8).pin_memory()
        a = b * self.capacity * self.rec_bytes
        self._pin[:count * self.rec_bytes].copy_(self.buf[a:a + count * self.rec_bytes], non_blocking=True)
        self._staged = (b, count)

    def staged(self, count):
        """Decode `count` records of the last stage() (the stream must have been synchronised since)."""
        b, have = self._staged
        if count > have:
            raise ValueError("more records asked for than were staged")
        return self.decode(self._pin.numpy()[:count * self.rec_bytes], count)

    def rows(self, b, count):
        """Download and decode the first `count` (<= capacity) records of trajectory b (synchronises the stream)."""
        count = min(int(count), self.capacity)
        if count <= 0:
            return self.decode(np.zeros((0,), dtype=np.uint8), 0)
        a = b * self.capacity * self.rec_bytes
        return self.decode(self.buf[a:a + count * self.rec_bytes].cpu().numpy(), count)


def solve_log_record(model, log, phase, x_nom, u_nom, K, k, cost, alpha_idx, active, iters, force=False):
    """One phase of the log of a loop the caller enqueues kernel by kernel (quattro_solve_log_record_f32)."""
    Bt, N, m = u_nom.shape
    check(_lib.load_for(model).quattro_solve_log_record_f32(log.byref(), int(phase), _ptr(x_nom), _ptr(u_nom), _ptr(K), _ptr(k),
                                                            _ptr(cost), _ptr(alpha_idx), _ptr(active), _ptr(iters), Bt, N,
                                                            model.n, m, int(bool(force)), _stream()),
          "quattro_solve_log_record_f32")


def ilqr_solve(model, x_nom, u_nom, K, k, cost, tol, max_iter, workspace, alphas=ALPHAS, reg=QUU_REG, x0=None,
               alpha_idx=None, active=None, iters=None, status=None, fixed_iters=False, reset=False, log=None,
               persistent=False, enqueue=False):
    """The whole solve from ONE C call with no host involvement: up to max_iter iterations, every trajectory stopping on
    its own test; x0 given = roll the nominal out from it first.  Everything in place (quattro_ilqr_solve_logged_f32).
    reset: the call sets active / iters / alpha_idx / status itself; log: a SolveLog ring filled by the device;
    persistent: take a persistent kernel that exists but is not the model's fastest form (a user model's); enqueue: never
    the persistent kernel."""
    Bt, N, m = u_nom.shape
    n = model.n
    f32, i32 = torch.float32, torch.int32
    _req(x_nom, (Bt, N + 1, n), f32, "x_nom"); _req(u_nom, (Bt, N, model.m), f32, "u_nom")
    _req(K, (Bt, N, m, n), f32, "K"); _req(k, (Bt, N, m), f32, "k"); _req(cost, (Bt,), torch.float64, "cost")
    _req(alpha_idx, (Bt,), i32, "alpha_idx"); _req(active, (Bt,), i32, "active"); _req(iters, (Bt,), i32, "iters")
    if status is not None:
        _req(status, (Bt,), i32, "status")
    flags = (_lib.SOLVE_FIXED_ITERS if fixed_iters else 0) | (_lib.SOLVE_RESET if reset else 0) | \
        (_lib.SOLVE_PERSISTENT if persistent else 0) | (_lib.SOLVE_ENQUEUE if enqueue else 0)
    if x0 is not None:
        _req(x0, (Bt, n), f32, "x0")
        flags |= _lib.SOLVE_SIMULATE
    if log is not None and (log.B != Bt or log.N != N or (log.n, log.m) != (n, m)):
        raise ValueError("log ring was built for another problem size")
    arr, na = _alphas(alphas)
    p = model.c_params()
    check(_lib.load_for(model).quattro_ilqr_solve_logged_f32(
        ctypes.byref(p), _ptr(x0), _ptr(x_nom), _ptr(u_nom), Bt, N, float(reg), arr, na, float(tol), int(max_iter), flags,
        _ptr(K), _ptr(k), _ptr(cost), _ptr(alpha_idx), _ptr(active), _ptr(iters), _ptr(status), _ptr(workspace),
        workspace.numel() * workspace.element_size(), None if log is None else log.byref(), _stream()),
        "quattro_ilqr_solve_logged_f32")


class PreparedSolve:
    """ilqr_solve with everything that does not change between solves of one QuattroILQR checked and converted ONCE: shapes,
    dtypes, device pointers, the alpha array, the parameter struct.  A call is then one ctypes call (single-trajectory
    solves are a few tens of microseconds of device time: a dozen shape checks per call were a third of the host's share)."""

    def __init__(self, model, x_nom, u_nom, K, k, cost, workspace, alphas, reg, x0, alpha_idx, active, iters, status):
        Bt, N, m = u_nom.shape
        n = model.n
        f32, i32 = torch.float32, torch.int32
        _req(x_nom, (Bt, N + 1, n), f32, "x_nom"); _req(u_nom, (Bt, N, model.m), f32, "u_nom")
        _req(K, (Bt, N, m, n), f32, "K"); _req(k, (Bt, N, m), f32, "k"); _req(cost, (Bt,), torch.float64, "cost")
        _req(alpha_idx, (Bt,), i32, "alpha_idx"); _req(active, (Bt,), i32, "active"); _req(iters, (Bt,), i32, "iters")
        _req(status, (Bt,), i32, "status"); _req(x0, (Bt, n), f32, "x0")
        self.keep = (x_nom, u_nom, K, k, cost, workspace, x0, alpha_idx, active, iters, status)     # the pointers stay valid
        self.arr, self.na = _alphas(alphas)
        self.p = model.c_params()
        self.fn = _lib.load_for(model).quattro_ilqr_solve_logged_f32
        self.dims = (Bt, N, n, m)
        self.head = (ctypes.byref(self.p), _ptr(x0), _ptr(x_nom), _ptr(u_nom), Bt, N, float(reg), self.arr, self.na)
        self.tail = (_ptr(K), _ptr(k), _ptr(cost), _ptr(alpha_idx), _ptr(active), _ptr(iters), _ptr(status), _ptr(workspace),
                     workspace.numel() * workspace.element_size())

    def __call__(self, tol, max_iter, fixed_iters=False, log=None, persistent=False, stream=None):
        flags = _lib.SOLVE_SIMULATE | _lib.SOLVE_RESET | (_lib.SOLVE_FIXED_ITERS if fixed_iters else 0) | \
            (_lib.SOLVE_PERSISTENT if persistent else 0)
        if log is not None and (log.B, log.N, log.n, log.m) != self.dims:
            raise ValueError("log ring was built for another problem size")
        check(self.fn(*self.head, float(tol), int(max_iter), flags, *self.tail, None if log is None else log.byref(),
                      _stream() if stream is None else stream), "quattro_ilqr_solve_logged_f32")


def mpc_run(model, x_cur, x_nom, u_nom, K, k, cost, tol, max_iter, n_steps, workspace, traj_x, traj_u, traj_iters,
            disturbance=None, alphas=ALPHAS, reg=QUU_REG, alpha_idx=None, active=None, iters=None, status=None):
    """B controllers x n_steps control steps (solve -> apply u_0 -> shift the warm start) in ONE launch
    (quattro_mpc_run_f32); x_cur and u_nom advance in place, the closed-loop record goes to traj_x / traj_u / traj_iters."""
    Bt, N, m = u_nom.shape
    n = model.n
    f32, i32 = torch.float32, torch.int32
    _req(x_cur, (Bt, n), f32, "x_cur"); _req(x_nom, (Bt, N + 1, n), f32, "x_nom"); _req(u_nom, (Bt, N, model.m), f32, "u_nom")
    _req(K, (Bt, N, m, n), f32, "K"); _req(k, (Bt, N, m), f32, "k"); _req(cost, (Bt,), torch.float64, "cost")
    _req(alpha_idx, (Bt,), i32, "alpha_idx"); _req(active, (Bt,), i32, "active"); _req(iters, (Bt,), i32, "iters")
    _req(traj_x, (Bt, n_steps + 1, n), f32, "traj_x"); _req(traj_u, (Bt, n_steps, m), f32, "traj_u")
    _req(traj_iters, (Bt, n_steps), i32, "traj_iters")
    if status is not None:
        _req(status, (Bt,), i32, "status")
    if disturbance is not None:
        _req(disturbance, (n_steps, Bt, n), f32, "disturbance")
    arr, na = _alphas(alphas)
    p = model.c_params()
    check(_lib.load_for(model).quattro_mpc_run_f32(ctypes.byref(p), _ptr(x_cur), _ptr(x_nom), _ptr(u_nom), Bt, N, float(reg), arr, na,
                                          float(tol), int(max_iter), int(n_steps), _ptr(traj_x), _ptr(traj_u),
                                          _ptr(traj_iters), _ptr(disturbance), _ptr(K), _ptr(k), _ptr(cost), _ptr(alpha_idx),
                                          _ptr(active), _ptr(iters), _ptr(status), _ptr(workspace),
                                          workspace.numel() * workspace.element_size(), _stream()), "quattro_mpc_run_f32")
