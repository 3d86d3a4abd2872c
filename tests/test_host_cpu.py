"""CPU-side checks of the product's host logic and of the C-ABI library itself (no kernel is launched)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden

import __graft_entry__ as entry
from oracle import ilqr as o_ilqr


@pytest.fixture(scope="module")
def lib():
    from quattro_ilqr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        entry.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from quattro_ilqr_amd import _lib
    declared = entry.declared_symbols()
    assert len(declared) >= 12
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/quattro_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared       # the Python binding covers the whole header
    assert lib.quattro_version() == 100


def test_model_params_struct_matches_c_header(tmp_path, lib):
    from quattro_ilqr_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "quattro_hip.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu", sizeof(quattro_model_params), '
                   'offsetof(quattro_model_params, phys), offsetof(quattro_model_params, q), '
                   'offsetof(quattro_model_params, x_ref), offsetof(quattro_model_params, r));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    P = _lib.ModelParams
    assert [int(v) for v in out] == [ctypes.sizeof(P), P.phys.offset, P.q.offset, P.x_ref.offset, P.r.offset]


def test_record_layout_queries(lib):
    from quattro_ilqr_amd import _lib
    assert lib.quattro_record_stride(12, 4, _lib.LAYOUT_ROWMAJOR) == 416
    assert lib.quattro_record_stride(12, 4, _lib.LAYOUT_TILE16) == 416      # 2n^2+2nm+m^2+n+m, no padding
    assert lib.quattro_record_stride(4, 1, _lib.LAYOUT_ROWMAJOR) == 48      # 46 padded to 16-byte multiple
    assert lib.quattro_record_stride(4, 1, _lib.LAYOUT_TILE16) == 0
    assert lib.quattro_record_stride(7, 3, _lib.LAYOUT_ROWMAJOR) == 0
    assert lib.quattro_preferred_layout(12, 4) == _lib.LAYOUT_TILE16
    assert lib.quattro_preferred_layout(4, 1) == _lib.LAYOUT_ROWMAJOR


def test_workspace_query_and_fused_driver_argument_checks(lib):
    """quattro_workspace_bytes covers records + terminal derivatives + line-search scratch (each 256-byte rounded);
    the fused driver refuses a missing, misaligned or short workspace before launching anything."""
    from quattro_ilqr_amd import _lib
    def up(b):
        return (b + 255) // 256 * 256
    for n, m, B, N in ((12, 4, 4096, 50), (4, 1, 1024, 50), (4, 1, 3, 7)):
        stride = lib.quattro_record_stride(n, m, lib.quattro_preferred_layout(n, m))
        want = up(4 * B * N * stride) + up(4 * B * n) + up(4 * B * n * n) + up(lib.quattro_linesearch_scratch_bytes(n, m, B, N))
        assert lib.quattro_workspace_bytes(n, m, B, N) == want
    assert lib.quattro_workspace_bytes(5, 2, 8, 8) == 0 and lib.quattro_workspace_bytes(12, 4, 0, 8) == 0
    p = _lib.ModelParams()
    p.model_id, p.n, p.m = _lib.MODEL_QUADROTOR, 12, 4
    null, one, al = ctypes.c_void_p(0), ctypes.c_void_p(1), ctypes.c_void_p(256)
    arr = (ctypes.c_float * 6)(1, .5, .25, .1, .05, .01)
    p.integrator = _lib.INTEGRATOR_EULER if hasattr(_lib, "INTEGRATOR_EULER") else 0
    need = lib.quattro_model_workspace_bytes(ctypes.byref(p), 2, 10)
    # Euler quadrotor -> TILE16C records: a 416-float header + 76 floats per step
    assert need == up(4 * (416 + 2 * 10 * 76)) + up(4 * 2 * 12) + up(4 * 2 * 144) + up(lib.quattro_linesearch_scratch_bytes(12, 4, 2, 10))
    assert need < lib.quattro_workspace_bytes(12, 4, 2, 10)
    assert lib.quattro_record_header(12, 4, _lib.LAYOUT_TILE16C) == 416 and lib.quattro_record_stride(12, 4, _lib.LAYOUT_TILE16C) == 76
    assert lib.quattro_model_layout(ctypes.byref(p)) == _lib.LAYOUT_TILE16C
    def call(ws, nbytes, x=one, na=6):
        return lib.quattro_ilqr_iterate_f32(ctypes.byref(p), x, one, 2, 10, 1e-6, arr, na, 1e-3, one, one, one, one, one,
                                            null, null, ws, nbytes, null)
    assert call(null, need) == _lib.ERR_WORKSPACE
    assert call(ctypes.c_void_p(264), need) == _lib.ERR_WORKSPACE        # not 256-byte aligned
    assert call(al, need - 1) == _lib.ERR_WORKSPACE
    assert call(al, need, x=null) == _lib.ERR_BAD_ARG
    assert call(al, need, na=9) == _lib.ERR_BAD_ARG


def test_bad_arguments_are_rejected_before_any_launch(lib):
    from quattro_ilqr_amd import _lib
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(1)     # never dereferenced: the checks below fail first
    # null pointer
    assert lib.quattro_riccati_sweep_f32(null, one, one, 1, 10, 0, 4, 1, 0, 1e-6, one, one, null, null, null) == _lib.ERR_BAD_ARG
    # t_start outside [0, N)
    assert lib.quattro_riccati_sweep_f32(one, one, one, 1, 10, 10, 4, 1, 0, 1e-6, one, one, null, null, null) == _lib.ERR_BAD_ARG
    # unknown (n, m)
    assert lib.quattro_riccati_sweep_f32(one, one, one, 1, 10, 0, 5, 2, 0, 1e-6, one, one, null, null, null) == _lib.ERR_UNSUPPORTED
    p = _lib.ModelParams()
    p.model_id, p.n, p.m = 99, 4, 1
    assert lib.quattro_simulate_f32(ctypes.byref(p), one, one, 1, 10, one, null, null) == _lib.ERR_UNSUPPORTED
    p.model_id, p.n, p.m = _lib.MODEL_QUADROTOR, 4, 1          # dims that do not belong to the model
    assert lib.quattro_simulate_f32(ctypes.byref(p), one, one, 1, 10, one, null, null) == _lib.ERR_UNSUPPORTED
    arr = (ctypes.c_float * 9)(*([1.0] * 9))
    p.n, p.m = 12, 4
    assert lib.quattro_rollout_f32(ctypes.byref(p), one, one, one, one, arr, 9, 1, 10, null, null, one, null, null) == _lib.ERR_BAD_ARG
    assert lib.quattro_status_string(_lib.ERR_UNSUPPORTED).decode().startswith("unsupported")
    # empty batches / horizons are argument errors, not launches of empty grids
    p.model_id, p.n, p.m = _lib.MODEL_QUADROTOR, 12, 4
    assert lib.quattro_simulate_f32(ctypes.byref(p), one, one, 0, 10, one, null, null) == _lib.ERR_BAD_ARG
    assert lib.quattro_simulate_f32(ctypes.byref(p), one, one, 4, 0, one, null, null) == _lib.ERR_BAD_ARG
    assert lib.quattro_riccati_sweep_f32(one, one, one, 0, 10, 0, 12, 4, 1, 1e-6, one, one, null, null, null) == _lib.ERR_BAD_ARG
    assert lib.quattro_linearize_f32(ctypes.byref(p), one, one, 4, 10, 0, 7, one, one, one, null, null) == _lib.ERR_UNSUPPORTED   # unknown layout
    assert lib.quattro_linearize_f32(ctypes.byref(p), one, one, 4, 10, 0, 1, one, one, null, null, null) == _lib.ERR_BAD_ARG      # V_x without V_xx
    p.integrator = _lib.INTEGRATOR_RK4
    assert lib.quattro_linearize_f32(ctypes.byref(p), one, one, 4, 10, 0, _lib.LAYOUT_TILE16C, one, one, one, null, null) == _lib.ERR_UNSUPPORTED
    assert lib.quattro_model_layout(ctypes.byref(p)) == _lib.LAYOUT_TILE16R
    assert lib.quattro_model_fuses_sweep(ctypes.byref(p)) == 2          # available, but the record path is faster stand-alone
    assert lib.quattro_model_has_device_loop(ctypes.byref(p)) == 1
    assert lib.quattro_linearize_sweep_scratch_bytes(ctypes.byref(p), 4, 10, 2) == 4 * 8 * 132 * 4
    # the RK4 quadrotor's fused sweep needs its coefficient scratch: refused without / with too little of it
    assert lib.quattro_linearize_sweep_f32(ctypes.byref(p), one, one, 4, 10, 0, 1e-6, one, one, null, null, null, 0, null) == _lib.ERR_WORKSPACE
    assert lib.quattro_linearize_sweep_f32(ctypes.byref(p), one, one, 4, 10, 0, 1e-6, one, one, null, null, one, 64, null) == _lib.ERR_WORKSPACE
    p.integrator = _lib.INTEGRATOR_EULER
    assert lib.quattro_model_fuses_sweep(ctypes.byref(p)) == 1
    assert lib.quattro_linearize_sweep_scratch_bytes(ctypes.byref(p), 4, 10, 0) == 0
    assert lib.quattro_linearize_f32(ctypes.byref(p), one, one, 4, 10, 0, _lib.LAYOUT_TILE16R, one, one, one, null, null) == _lib.ERR_UNSUPPORTED
    assert lib.quattro_linearize_sweep_f32(ctypes.byref(p), null, one, 4, 10, 0, 1e-6, one, one, null, null, null, 0, null) == _lib.ERR_BAD_ARG
    # the device-resident loops: argument checks before any launch
    arr6 = (ctypes.c_float * 6)(1.0, 0.5, 0.25, 0.1, 0.05, 0.01)
    assert lib.quattro_ilqr_solve_f32(ctypes.byref(p), null, one, one, 4, 10, 1e-6, arr6, 6, 1e-3, 5, 1, one, one, one, one, one, one, null,
                                      one, 1 << 30, null) == _lib.ERR_BAD_ARG          # SIMULATE flag without x0
    assert lib.quattro_ilqr_solve_f32(ctypes.byref(p), one, one, one, 4, 10, 1e-6, arr6, 6, 1e-3, 5, 0, one, one, one, one, one, one, null,
                                      null, 0, null) == _lib.ERR_WORKSPACE
    assert lib.quattro_mpc_run_f32(ctypes.byref(p), one, one, one, 4, 10, 1e-6, arr6, 6, 1e-3, 5, 0, one, one, one, null, one, one, one, one,
                                   one, one, null, one, 1 << 30, null) == _lib.ERR_BAD_ARG        # n_steps = 0
    p.integrator = 5
    assert lib.quattro_simulate_f32(ctypes.byref(p), one, one, 4, 10, one, null, null) == _lib.ERR_UNSUPPORTED
    assert lib.quattro_tf_gains_bf16(None, one, one, 1, 10, 12, 4, one, one, null, null) == _lib.ERR_BAD_ARG
    w = _lib.TfWeights(); w.c_dim = 51
    assert lib.quattro_tf_gains_bf16(ctypes.byref(w), one, one, 1, 10, 12, 4, one, one, null, null) == _lib.ERR_BAD_ARG   # c != m (1 + n)


def test_ops_validate_tensors_on_the_host():
    torch = pytest.importorskip("torch")
    from quattro_ilqr_amd import models, ops
    md = models.quadrotor_model()
    x0 = torch.zeros((2, 12))
    u = torch.zeros((2, 5, 4))
    with pytest.raises(ValueError, match="GPU"):
        ops.simulate(md, x0, u)
    with pytest.raises(ValueError):
        ops.linearize(models.cartpole_model(), torch.zeros((2, 6, 12)), u)
    with pytest.raises(ValueError, match="Unknown integration method"):
        models.quadrotor_model(integrator="heun").c_params()


def test_device_model_struct_contents():
    from quattro_ilqr_amd import _lib, models
    p = models.quadrotor_model(dt=0.02, integrator="rk4").c_params()
    assert (p.model_id, p.integrator, p.n, p.m) == (_lib.MODEL_QUADROTOR, _lib.INTEGRATOR_RK4, 12, 4)
    assert abs(p.dt - 0.02) < 1e-9 and p.barrier_alpha == 1000.0 and p.barrier_beta == 10.0
    assert list(p.q)[:12] == [10, 10, 50, 1, 1, 1, 10, 10, 50, 1, 1, 1] and abs(p.x_ref[2] - 0.5) < 1e-9
    assert [round(v, 5) for v in list(p.phys)[:7]] == [1.0, 0.02, 0.02, 0.04, 0.1, 9.81, 0.01]
    c = models.cartpole_model().c_params()
    assert (c.n, c.m, c.barrier_alpha) == (4, 1, 0.0) and abs(c.r[0] - 0.001) < 1e-9


def test_ilqr_tf_constructor_contract():
    """ValueError for tf_window >= horizon (reference :96-97); no silent CPU path for arbitrary callables."""
    pytest.importorskip("torch")
    from quattro_ilqr_amd import iLQR_TF, models
    u0 = [np.zeros(1) for _ in range(10)]
    md = models.cartpole_model()
    with pytest.raises(ValueError, match="tf_window must be less than the horizon"):
        iLQR_TF(None, None, None, np.zeros(4), u0, 10, tf_window=10, model=md)
    with pytest.raises(NotImplementedError, match="no CPU fallback"):
        iLQR_TF(lambda x, u: x, lambda x, u: 0.0, lambda x: 0.0, np.zeros(4), u0, 10, tf_window=5)
    il = iLQR_TF(None, None, None, np.zeros(4), u0, 10, model=md, tol=1e-1, tf_window=5)
    assert il.get_time() == ([], [], [])
    off = il.get_state_offset()
    off[0] = 3.0
    assert il.get_state_offset()[0] == 0.0          # copy semantics (:612)
    il.set_state_offset(off)
    assert il.get_state_offset()[0] == 3.0

    class TF:
        prompt_len = 3

        def predict(self, x, p):
            return None
    il = iLQR_TF(None, None, None, np.zeros(4), u0, 10, model=md, tf=TF(), tf_window=5)
    assert il.tf_window == 3 and len(il.get_time()) == 4            # :118-119, :597-600


def test_prompt_pack_and_unpack_layouts_match_the_oracle():
    torch = pytest.importorskip("torch")
    from quattro_ilqr_amd.solver import _pack_prompt, _unpack_prediction
    g = load_golden("hybrid_quadrotor.npz")
    k_seg, K_seg = g["k_seg"][0], g["K_seg"][0]                    # (P, 4), (P, 4, 12)
    want = o_ilqr.pack_prompt(list(k_seg), list(K_seg))
    got = _pack_prompt(torch.as_tensor(k_seg)[None], torch.as_tensor(K_seg)[None])[0].numpy()
    assert np.array_equal(got, want) and np.array_equal(got, g["prompt"][0].astype(got.dtype)) or np.allclose(got, g["prompt"][0], rtol=1e-7)
    pred = g["prediction"][0]                                       # (49, 52)
    wk, wK = o_ilqr.unpack_prediction(pred, 4, 12)
    gk, gK = _unpack_prediction(torch.as_tensor(pred)[None], 4, 12)
    assert np.array_equal(gk[0].numpy(), wk) and np.array_equal(gK[0].numpy(), wK)


def test_cartpole_lqr_law_and_switcher_match_the_reference():
    """CartPoleMPC's LQR-only mode and ControllerSwitcher (host arithmetic, no GPU): DARE gain, the double sign of the
    applied LQR control, and the blending weights, against values produced by the reference (G11)."""
    import quattro_ilqr_amd as q
    from conftest import load_golden
    g = load_golden("lqr_cartpole.npz")
    mpc = q.CartPoleMPC(horizon=30, dt=0.01, integration_method="euler", lqr_only=True, device="cpu")
    assert mpc.ilqr is None
    A_d, B_d = mpc.linearized_dynamics(0.01)
    assert np.allclose(A_d, g["A_d"], rtol=0, atol=1e-15) and np.allclose(B_d, g["B_d"], rtol=0, atol=1e-15)
    assert np.array_equal(mpc.Q_lqr, g["Q_lqr"]) and np.array_equal(mpc.R_lqr, g["R_lqr"])
    for x, u_ref, step_ref in zip(g["xs"], g["u_lqr"], g["u_step"]):
        assert np.allclose(mpc.compute_linear_lqr_control(x), u_ref, rtol=1e-10)
        xs, u = mpc.control_step(x)
        assert xs == [] and np.allclose(u, step_ref, rtol=1e-10)          # = MINUS the LQR law (cartpole_mpc.py:322)
    sw = q.ControllerSwitcher(epsilon_low=0.5, epsilon_high=1.5)
    for e, w_ref in zip(g["errs"], g["w"]):
        sw.update_error(e)
        assert abs(sw.get_blending_weight(0.01) - w_ref) < 1e-15
    assert len(sw.error_history) == 3 and sw.compute_acceleration_norm(0.01) > 0


def test_training_parameter_layout_covers_the_reference_state_dict(tmp_path, lib):
    """The flat parameter array of quattro_tf_train_step_f32: one block per entry of the reference module's state dict
    (transformer_model.py:85-120), in its own shape, 16-byte aligned, non-overlapping; struct layout as in the header;
    shapes without kernels report 0 parameters.  Host-only queries: no GPU needed."""
    from quattro_ilqr_amd import _lib
    src = tmp_path / "sz2.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "quattro_hip.h"\n'
                   'int main(){printf("%zu %zu %d", sizeof(quattro_tf_train_desc), offsetof(quattro_tf_train_desc, dropout), '
                   'QUATTRO_TF_P_COUNT);return 0;}\n')
    exe = tmp_path / "sz2"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    size, off_drop, n_blocks = (int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True,
                                                                text=True).stdout.split())
    D = _lib.TfTrainDesc
    assert (size, off_drop) == (ctypes.sizeof(D), D.dropout.offset)
    assert n_blocks == len(_lib.TF_TRAIN_GLOBAL) + len(_lib.TF_TRAIN_LAYER)

    def desc(n, c, d, H, layers, ff, NS, P, T):
        x = D()
        (x.state_dim, x.control_dim, x.d_model, x.nhead, x.n_layers, x.d_ff, x.n_state_tok, x.prompt_len, x.target_len,
         x.dropout) = n, c, d, H, layers, ff, NS, P, T, 0.1
        return x

    for shape in ((12, 52, 128, 4, 3, 512, 51, 1, 49), (4, 5, 128, 4, 2, 256, 31, 5, 26), (3, 7, 64, 2, 1, 96, 6, 2, 5)):
        n, c, d, H, layers, ff, NS, P, T = shape
        x = desc(*shape)
        total = lib.quattro_tf_train_param_count(ctypes.byref(x))
        glob = [T * d, d * n, d, d * c, d, c * d, c]
        lay = [3 * d * d, 3 * d, d * d, d, ff * d, ff, d * ff, d, d, d, d, d]
        blocks = [(lib.quattro_tf_train_param_offset(ctypes.byref(x), i, 0), sz) for i, sz in enumerate(glob)]
        for l in range(layers):
            blocks += [(lib.quattro_tf_train_param_offset(ctypes.byref(x), len(glob) + i, l), sz) for i, sz in enumerate(lay)]
        assert all(o >= 0 and o % 4 == 0 for o, _ in blocks)
        blocks.sort()
        for (o0, s0), (o1, _) in zip(blocks, blocks[1:]):
            assert o0 + s0 <= o1
        assert blocks[-1][0] + blocks[-1][1] <= total < sum(s for _, s in blocks) + 4 * len(blocks)
        assert lib.quattro_tf_train_param_offset(ctypes.byref(x), n_blocks, 0) == -1
        assert lib.quattro_tf_train_param_offset(ctypes.byref(x), len(glob), layers) == -1
        assert lib.quattro_tf_train_workspace_bytes(ctypes.byref(x), 8) > 0
    # head dimensions up to 32 train on the device (the reference's default constructor: d_model 64, nhead 8), any d_model
    # up to 512; head dimension 64, d_model 1024, a d_model the heads do not divide do not
    assert lib.quattro_tf_train_param_count(ctypes.byref(desc(4, 5, 64, 8, 3, 256, 31, 10, 21))) > 0
    assert lib.quattro_tf_train_param_count(ctypes.byref(desc(4, 5, 96, 3, 3, 256, 31, 10, 21))) > 0
    assert lib.quattro_tf_train_param_count(ctypes.byref(desc(4, 5, 128, 2, 3, 256, 31, 10, 21))) == 0
    assert lib.quattro_tf_train_param_count(ctypes.byref(desc(4, 5, 1024, 32, 3, 256, 31, 10, 21))) == 0
    assert lib.quattro_tf_train_param_count(ctypes.byref(desc(4, 5, 100, 3, 3, 256, 31, 10, 21))) == 0
    # argument checks of the step itself happen before any launch
    x = desc(12, 52, 128, 4, 3, 512, 51, 1, 49)
    assert lib.quattro_tf_train_step_f32(ctypes.byref(x), None, None, None, 0, None, None, None, None, 4, 0, 1, None, None,
                                         None) == _lib.ERR_BAD_ARG
    assert lib.quattro_tf_adam_f32(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 1, None) == _lib.ERR_BAD_ARG


def test_user_model_library_builds_and_exports_the_whole_abi():
    """compile_model (hipcc cross-compiles without a GPU): the per-model library exports every symbol of include/quattro_hip.h,
    knows its own (n, m) — which libquattro_hip.so does not — and refuses dims outside the header's limits.  No compute calls."""
    import ctypes
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import _lib
    md = q.compile_model("pendulum_cpu_test", 2, 1, dt=0.02, integrator="rk4", phys=(9.81, 1.0),
                         rate="xd[0] = x[1];  xd[1] = -(P[0] / P[1]) * sin(x[0]) + u[0];",
                         q=(1.0, 0.1), r=(0.01,), qf=(50.0, 5.0), x_ref=(3.14159, 0.0))
    assert isinstance(md, q.DeviceModel) and md.model_id == _lib.MODEL_USER and os.path.exists(md.lib_path)
    lib = _lib.load_for(md)
    for name in _lib.SIGNATURES:
        assert hasattr(lib, name)
    p = md.c_params()
    assert lib.quattro_record_stride(2, 1, _lib.LAYOUT_ROWMAJOR) > 0 and _lib.load().quattro_record_stride(2, 1, 0) == 0
    assert lib.quattro_model_layout(ctypes.byref(p)) == _lib.LAYOUT_ROWMAJOR
    assert _lib.load().quattro_model_layout(ctypes.byref(p)) == -1
    assert lib.quattro_model_workspace_bytes(ctypes.byref(p), 16, 20) > 0
    assert lib.quattro_model_has_device_loop(ctypes.byref(p)) == 2 and lib.quattro_model_fuses_sweep(ctypes.byref(p)) == 0
    # the built-in models still answer from the same library (it resolves their kernels from libquattro_hip.so)
    pq = q.quadrotor_model().c_params()
    assert lib.quattro_model_layout(ctypes.byref(pq)) == _lib.LAYOUT_TILE16C
    again = q.compile_model("pendulum_cpu_test", 2, 1, dt=0.02, integrator="rk4", phys=(9.81, 1.0),
                            rate="xd[0] = x[1];  xd[1] = -(P[0] / P[1]) * sin(x[0]) + u[0];",
                            q=(1.0, 0.1), r=(0.01,), qf=(50.0, 5.0), x_ref=(3.14159, 0.0))
    assert again.lib_path == md.lib_path                                    # cached by content
    with pytest.raises(ValueError):
        q.compile_model("too_big", 17, 2, rate="xd[0] = x[0];")
    with pytest.raises(_lib.QuattroError):
        q.compile_model("broken", 2, 1, rate="xd[0] = undefined_symbol;")


# ------------------------------------------------------------------------------------------------ reference problem objects
REFERENCE = "/root/reference"


def _oracle_device_eval(md, xs, us, device):
    """Stand-in for solver._device_eval where there is no GPU (the build container): the fp64 oracle of the same model."""
    from oracle import models as o_models
    mk = o_models.quadrotor_spec if md.name == "quadrotor" else o_models.cartpole_spec
    integ = {"euler": o_models.INTEGRATOR_EULER, "rk4": o_models.INTEGRATOR_RK4}[md.integrator]
    spec = mk(dt=md.dt, integrator=integ, x_ref=np.asarray(md.x_ref))
    spec.Q, spec.R, spec.Qf = np.diag(md.q), np.diag(md.r), np.diag(md.qf)       # the model under test, not the defaults
    spec.barrier_alpha, spec.barrier_beta = md.barrier_alpha, md.barrier_beta
    f = np.array([spec.f(x, u) for x, u in zip(xs, us)])
    return f, np.array([spec.L(x, u) for x, u in zip(xs, us)]), np.array([spec.Lf(x) for x in xs])


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference checkout (build container only)")
@pytest.mark.parametrize("which", ["quadrotor", "cartpole"])
def test_the_references_own_mpc_objects_bind_to_the_dropin(which, monkeypatch):
    """SURVEY 8(b) 'never an error': the reference's QuadrotorMPC / CartPoleMPC (examples/*/…_mpc.py), constructed UNCHANGED
    with this package's iLQR_TF in place of theirs, bind — their bound methods are recognised by the owner's attribute set and
    verified by probing the three callables against the device model (here evaluated by the oracle: no GPU in this container).
    A tampered cost (attributes say one thing, the callable computes another) must be refused."""
    pytest.importorskip("torch")
    import importlib
    from quattro_ilqr_amd import solver
    for p in (REFERENCE, os.path.join(REFERENCE, "examples", which)):
        monkeypatch.syspath_prepend(p)
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    monkeypatch.setattr(solver, "_device_eval", _oracle_device_eval)
    ref_mod = importlib.import_module(f"{which}_mpc")
    monkeypatch.setattr(ref_mod, "iLQR_TF", solver.iLQR_TF)            # the drop-in: same constructor call, our class
    cls = ref_mod.QuadrotorMPC if which == "quadrotor" else ref_mod.CartPoleMPC
    kw = dict(horizon=20, dt=0.01, integration_method="euler")
    mpc = cls(**kw) if which == "quadrotor" else cls(ilqr_only=True, **kw)
    assert isinstance(mpc.ilqr, solver.iLQR_TF)
    md = mpc.ilqr._model()
    import quattro_ilqr_amd as q
    mirror = (q.QuadrotorMPC if which == "quadrotor" else q.CartPoleMPC).__new__(q.QuadrotorMPC if which == "quadrotor" else q.CartPoleMPC)
    mirror.dt, mirror.integration_method, mirror.x_ref = 0.01, "euler", mpc.x_ref
    mirror.Q, mirror.R, mirror.Qf = mpc.Q, mpc.R, mpc.Qf
    if which == "quadrotor":
        mirror.alpha, mirror.beta = mpc.alpha, mpc.beta
    assert md == mirror.device_model()                                  # the same device model the mirror classes describe
    assert mpc.ilqr._model() is md                                      # recognised and probed once
    if which == "cartpole":
        assert mpc.ilqr.tol == 1e-1
    # a changed attribute is picked up (and probed again); RK4 binds too
    mpc.integration_method = "rk4"
    assert mpc.ilqr._model().integrator == "rk4"
    # a callable that does not compute what the attributes claim is refused, with the way out in the message
    mpc.Q = mpc.Q * 1.0
    import types
    real = cls.running_cost
    mpc.ilqr.L = types.MethodType(lambda self, x, u: 1.01 * real(self, x, u), mpc)     # still bound to the same object
    mpc.Qf = 2.0 * mpc.Qf                                               # (forces a new probe)
    with pytest.raises(NotImplementedError, match="compile_model"):
        mpc.ilqr._model()


def test_objects_that_only_look_like_a_reference_problem_are_not_recognised():
    pytest.importorskip("torch")
    from quattro_ilqr_amd import solver

    class Dyn:
        m, Ix, Iy, Iz, arm, g = 1.0, 0.02, 0.02, 0.04, 0.1, 9.81

    class P:
        dynamics = Dyn()
        x_ref, dt, integration_method = np.zeros(12), 0.01, "euler"
        Q, R, Qf = np.eye(12), np.eye(4), np.eye(12)
        alpha, beta = 1.0, 1.0
    assert solver.recognise_problem_object(P()).name == "quadrotor"
    bad = P(); bad.Q = np.eye(12) + 0.1                                 # not diagonal: outside the built-in cost family
    assert solver.recognise_problem_object(bad) is None
    bad = P(); bad.integration_method = "midpoint"
    assert solver.recognise_problem_object(bad) is None
    bad = P(); bad.R = np.eye(3)
    assert solver.recognise_problem_object(bad) is None
    assert solver.recognise_problem_object(object()) is None
