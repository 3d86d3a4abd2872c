"""ctypes binding of libquattro_hip.so (C ABI: include/quattro_hip.h).

The library is built in-tree by `quattro-transformer-ilqr_amd/csrc/Makefile` (hipcc, gfx950) and must sit next to
this file.  There is no fallback: if it is missing or does not export a declared symbol, importing the ops fails.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_long, c_size_t, c_uint64, c_void_p

MAX_NX, MAX_NU, MAX_ALPHAS = 16, 8, 8

QUATTRO_OK = 0
ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_LAUNCH, ERR_WORKSPACE = -1, -2, -3, -4
TRAJ_NONFINITE, TRAJ_SINGULAR, TRAJ_ILLCOND = 1, 2, 4
MODEL_CARTPOLE, MODEL_QUADROTOR, MODEL_USER = 1, 2, 3
INTEGRATOR_EULER, INTEGRATOR_RK4 = 0, 1
LAYOUT_ROWMAJOR, LAYOUT_TILE16, LAYOUT_TILE16C, LAYOUT_TILE16R, LAYOUT_ROWMAJOR_TILE = 0, 1, 2, 3, 4
SOLVE_SIMULATE, SOLVE_FIXED_ITERS, SOLVE_RESET, SOLVE_ENQUEUE, SOLVE_PERSISTENT = 1, 2, 4, 8, 16
LOG_TRAJ, LOG_GAINS = 1, 2
LOG_FIELD_X, LOG_FIELD_U, LOG_FIELD_K, LOG_FIELD_KFF = 0, 1, 2, 3
LOG_PHASE_BEGIN, LOG_PHASE_BACKWARD_DONE, LOG_PHASE_GAINS_DONE, LOG_PHASE_END = 0, 1, 2, 3
LOG_HEADER_BYTES = 64


class ModelParams(ctypes.Structure):
    """Mirror of `quattro_model_params` (include/quattro_hip.h)."""
    _fields_ = [
        ("model_id", c_int32), ("integrator", c_int32), ("n", c_int32), ("m", c_int32),
        ("dt", c_float), ("barrier_alpha", c_float), ("barrier_beta", c_float), ("reserved0", c_float),
        ("phys", c_float * 8),
        ("q", c_float * MAX_NX), ("qf", c_float * MAX_NX), ("x_ref", c_float * MAX_NX),
        ("r", c_float * MAX_NU),
    ]


class SolveLogC(ctypes.Structure):
    """Mirror of `quattro_solve_log` (include/quattro_hip.h)."""
    _fields_ = [("records", c_void_p), ("capacity", c_int32), ("flags", c_int32)]


TF_MAX_LAYERS = 8
TF_PRECISION_BF16, TF_PRECISION_F16 = 0, 1


class TfTrainDesc(ctypes.Structure):
    """Mirror of `quattro_tf_train_desc` (include/quattro_hip.h)."""
    _fields_ = [(n, c_int32) for n in ("state_dim", "control_dim", "d_model", "nhead", "n_layers", "d_ff", "n_state_tok",
                                       "prompt_len", "target_len")] + [("dropout", c_float)]


# blocks of the flat training parameter array (QUATTRO_TF_P_*): reference state_dict name (per-layer names take the
# `transformer_decoder.layers.<l>.` prefix), in the order of the header
TF_TRAIN_GLOBAL = ["target_embedding", "state_embed.weight", "state_embed.bias", "control_embed.weight",
                   "control_embed.bias", "output_linear.weight", "output_linear.bias"]
TF_TRAIN_LAYER = ["self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                  "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias",
                  "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"]


class TfWeights(ctypes.Structure):
    """Mirror of `quattro_tf_weights` (include/quattro_hip.h): dims + device pointers."""
    _fields_ = (
        [(n, c_int32) for n in ("n_x", "c_dim", "d_model", "n_head", "d_ff", "n_layers", "n_state_tok", "prompt_len",
                                "target_len", "precision")]
        + [(n, c_void_p) for n in ("x_mean", "x_std", "u_mean", "u_std", "w_state", "state_b", "ctrl_w", "ctrl_b",
                                   "tok_bias")]
        + [(n, c_void_p * TF_MAX_LAYERS) for n in ("w_qkv", "b_qkv", "w_o", "b_o", "w_1", "b_1", "w_2", "b_2",
                                                   "ln1_g", "ln1_b", "ln2_g", "ln2_b")]
        + [("w_out", c_void_p), ("b_out", c_void_p), ("tok_bias_t", c_void_p), ("w_stream", c_void_p),
           ("p_stream", c_void_p)])


LIB_NAME = "libquattro_hip.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)
# diagnostics only (scripts/ab_*.sh): another build of the same library, to time two variants of a kernel on ONE GPU box
# in one call (boxes differ by several per cent, so numbers from different calls cannot rank variants)
LIB_PATH = os.environ.get("QUATTRO_HIP_LIB", LIB_PATH)

_P = c_void_p  # device pointers travel as integers from tensor.data_ptr()

# name -> (restype, argtypes); kept in one table so tests can check every declared symbol is exported
SIGNATURES = {
    "quattro_version": (c_int, []),
    "quattro_status_string": (c_char_p, [c_int]),
    "quattro_record_stride": (c_int, [c_int, c_int, c_int]),
    "quattro_record_header": (c_int, [c_int, c_int, c_int]),
    "quattro_preferred_layout": (c_int, [c_int, c_int]),
    "quattro_model_layout": (c_int, [POINTER(ModelParams)]),
    "quattro_pack_derivs_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "quattro_unpack_derivs_f32": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "quattro_riccati_sweep_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, _P,
                                          _P, _P]),
    "quattro_linearize_f32": (c_int, [POINTER(ModelParams), _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "quattro_model_fuses_sweep": (c_int, [POINTER(ModelParams)]),
    "quattro_linearize_sweep_scratch_bytes": (c_size_t, [POINTER(ModelParams), c_int, c_int, c_int]),
    "quattro_linearize_sweep_f32": (c_int, [POINTER(ModelParams), _P, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P,
                                            c_size_t, _P]),
    "quattro_linearize_sweep_rows_f32": (c_int, [POINTER(ModelParams), _P, _P, c_int, c_int, c_int, c_float, _P, _P, c_int, _P, _P,
                                                 _P, c_size_t, _P]),
    "quattro_simulate_f32": (c_int, [POINTER(ModelParams), _P, _P, c_int, c_int, _P, _P, _P]),
    "quattro_total_cost_f32": (c_int, [POINTER(ModelParams), _P, _P, c_int, c_int, _P, _P]),
    "quattro_rollout_f32": (c_int, [POINTER(ModelParams), _P, _P, _P, _P, POINTER(c_float), c_int, c_int, c_int, _P,
                                    _P, _P, _P, _P]),
    "quattro_linesearch_scratch_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "quattro_linesearch_f32": (c_int, [POINTER(ModelParams), _P, _P, _P, _P, POINTER(c_float), c_int, c_int, c_int,
                                       c_double, _P, _P, _P, _P, _P, c_size_t, _P]),
    "quattro_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "quattro_model_workspace_bytes": (c_size_t, [POINTER(ModelParams), c_int, c_int]),
    "quattro_ilqr_iterate_f32": (c_int, [POINTER(ModelParams), _P, _P, c_int, c_int, c_float, POINTER(c_float), c_int,
                                         c_double, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "quattro_model_has_device_loop": (c_int, [POINTER(ModelParams)]),
    "quattro_ilqr_solve_f32": (c_int, [POINTER(ModelParams), _P, _P, _P, c_int, c_int, c_float, POINTER(c_float), c_int,
                                       c_double, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "quattro_solve_log_record_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "quattro_solve_log_offset": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "quattro_ilqr_solve_logged_f32": (c_int, [POINTER(ModelParams), _P, _P, _P, c_int, c_int, c_float, POINTER(c_float), c_int,
                                              c_double, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t,
                                              POINTER(SolveLogC), _P]),
    "quattro_solve_log_record_f32": (c_int, [POINTER(SolveLogC), c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int,
                                             c_int, c_int, _P]),
    "quattro_mpc_run_f32": (c_int, [POINTER(ModelParams), _P, _P, _P, c_int, c_int, c_float, POINTER(c_float), c_int,
                                    c_double, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "quattro_tf_stream_elems": (c_size_t, [POINTER(TfWeights)]),
    "quattro_tf_param_floats": (c_size_t, [POINTER(TfWeights)]),
    "quattro_tf_pack_stream_bf16": (c_int, [POINTER(TfWeights), _P, _P, _P]),
    "quattro_tf_forward_bf16": (c_int, [POINTER(TfWeights), _P, _P, c_int, _P, _P]),
    "quattro_tf_gains_bf16": (c_int, [POINTER(TfWeights), _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "quattro_tf_pack_stream_f16": (c_int, [POINTER(TfWeights), _P, _P, _P]),
    "quattro_tf_forward_f16": (c_int, [POINTER(TfWeights), _P, _P, c_int, _P, _P]),
    "quattro_tf_gains_f16": (c_int, [POINTER(TfWeights), _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "quattro_tf_train_param_count": (c_size_t, [POINTER(TfTrainDesc)]),
    "quattro_tf_train_param_offset": (c_long, [POINTER(TfTrainDesc), c_int, c_int]),
    "quattro_tf_train_workspace_bytes": (c_size_t, [POINTER(TfTrainDesc), c_int]),
    "quattro_tf_train_step_f32": (c_int, [POINTER(TfTrainDesc), _P, _P, _P, c_size_t, _P, _P, _P, _P, c_int, c_uint64,
                                          c_int, _P, _P, _P]),
    "quattro_tf_adam_f32": (c_int, [_P, _P, _P, _P, c_size_t, c_float, c_float, c_float, c_float, c_int, _P]),
    "quattro_tf_train_dropout_mask_f32": (c_int, [c_uint64, c_float, c_int, c_size_t, _P, _P]),
}

_lib = None
_user_libs = {}


class QuattroError(RuntimeError):
    pass


def _bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


def load(path=None):
    """dlopen the library (once) and attach the prototypes.  `path`: a user-model library (user_model.compile_model) — the
    same C ABI with the caller's problem compiled in as QUATTRO_MODEL_USER; it resolves everything it does not define
    itself from libquattro_hip.so, which is loaded first."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QuattroError(
                f"{LIB_NAME} not found at {LIB_PATH}: build it with `python __graft_entry__.py` or "
                "`make -C quattro-transformer-ilqr_amd/csrc`.  There is no CPU fallback for the iLQR hot path.")
        _lib = _bind(LIB_PATH)
    if not path:
        return _lib
    lib = _user_libs.get(path)
    if lib is None:
        if not os.path.exists(path):
            raise QuattroError(f"user-model library {path} not found (rebuild it with user_model.compile_model)")
        lib = _user_libs[path] = _bind(path)
    return lib


def load_for(model):
    """The library that holds `model`'s kernels: libquattro_hip.so for the built-in problems, its own for a user model."""
    return load(getattr(model, "lib_path", None) or None)


def check(status, what):
    if status != QUATTRO_OK:
        msg = load().quattro_status_string(status).decode()
        if status == ERR_UNSUPPORTED:
            raise NotImplementedError(f"{what}: {msg}")
        if status == ERR_BAD_ARG:
            raise ValueError(f"{what}: {msg}")
        raise QuattroError(f"{what}: {msg} (status {status})")
