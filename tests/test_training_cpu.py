"""Training side of the gain predictor, CPU: the training-time forward is pinned to the reference
module's fp32 outputs on the shipped checkpoints (G7 fixtures), and fit / save / load are exercised end to end on a
small synthetic training set."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, PKG_DIR, load_golden, rel_fro

sys.path.insert(0, PKG_DIR)
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("model", ["cartpole", "quadrotor"])
def test_training_forward_matches_the_reference_module_in_fp32(model):
    from quattro_ilqr_amd import training
    z = load_golden(f"tf_weights_{model}.npz")
    g = load_golden(f"tf_{model}.npz")
    W = {k: torch.tensor(z[k].astype(np.float32)) for k in z.files if not k.startswith(("norm.", "hp."))}
    norm = {k[5:]: z[k].astype(np.float64) for k in z.files if k.startswith("norm.")}
    nhead, P = int(z["hp.nhead"]), int(z["hp.prompt_len"])
    buffers = {"pos_encoder.pe": W.pop("pos_encoder.pe")}
    xn = torch.tensor(((g["x_err"] - norm["x_mean"]) / norm["x_std"]).astype(np.float32))
    un = torch.tensor(((g["prompt"] - norm["u_mean"]) / norm["u_std"]).astype(np.float32))[:, -P:]
    with torch.no_grad():
        out = training.forward(W, buffers, xn, un, nhead).double().numpy()
    pred = out * norm["u_std"] + norm["u_mean"]
    assert rel_fro(pred, g["pred_fp32"]) < 1e-5
    # dropout is the identity in evaluation mode and active in training mode
    with torch.no_grad():
        torch.manual_seed(0)
        tr = training.forward(W, buffers, xn, un, nhead, dropout=0.1, training=True).double().numpy()
        ev = training.forward(W, buffers, xn, un, nhead, dropout=0.1, training=False).double().numpy()
    assert np.array_equal(ev, out) and not np.allclose(tr, out)


def _toy_logs(E, N, n, m, seed):
    """Gains that are a smooth function of the states, so a small model can fit them."""
    r = np.random.default_rng(seed)
    x = np.cumsum(0.1 * r.standard_normal((E, N + 1, n)), axis=1)
    Wk = r.standard_normal((n, m)) * 0.5
    k = np.tanh(x[:, :N] @ Wk)
    K = np.einsum("etm,n->etmn", k, np.linspace(0.5, 1.5, n)) + 0.3
    return x.astype(np.float32), k.astype(np.float32), K.astype(np.float32)


def test_fit_save_load_round_trip(tmp_path):
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import datagen, training
    n, m, N, P = 4, 1, 12, 3
    xa, ka, Ka = _toy_logs(128, N, n, m, 0)
    x, k, K = xa[:96], ka[:96], Ka[:96]
    xt, kt, Kt = xa[96:], ka[96:], Ka[96:]                                # held-out sequences of the same process
    data = datagen.create_dataset(x, k, K, P)
    test = datagen.create_dataset(xt, kt, Kt, P)
    tf = q.TransformerILQR(n, m * (1 + n), prompt_len=P, d_model=128, nhead=4, num_decoder_layers=1, dim_feedforward=128,
                           dropout=0.0, max_seq_len=40, device="cpu")
    tf.fit(data, test, num_epochs=12, batch_size=16, learning_rate=2e-3, patience=3)
    assert tf.target_len == (N + 1) - P                                  # transformer_ilqr.py:106 (x_seq has N + 1 rows)
    assert len(tf.train_loss_history) == len(tf.test_loss_history) <= 12
    assert tf.train_loss_history[-1] < 0.5 * tf.train_loss_history[0]
    assert min(tf.test_loss_history) < 0.6 * tf.test_loss_history[0]
    assert tf._w["transformer_decoder.layers.0.linear1.weight"].shape == (128, 128)
    # the same data given as the reference's table of columns
    cols = {"x_seq": list(x), "k_seq": [list(r) for r in k], "K_seq": [list(r) for r in K]}
    xd, kd = tf._create_dataset(cols)
    assert np.array_equal(xd, data[0]) and np.array_equal(kd, data[1])
    # save -> the reference's directory format -> load
    path = tf.save("toy", root=str(tmp_path))
    assert os.path.basename(path).endswith("toy_decoder_dec1_dmodel128_nhead4_ff128_drop0.0_epoch12_promptlen3_%s" % os.path.basename(path).split("_")[-1])
    assert sorted(os.listdir(path)) == ["tf_model.pt", "tf_model_normalizer.npz"]
    meta = np.load(os.path.join(path, "tf_model_normalizer.npz"), allow_pickle=False)
    assert set(meta.files) == {"x_mean", "x_std", "u_mean", "u_std", "target_len", "prompt_len", "state_dim", "control_dim",
                               "d_model", "nhead", "num_decoder_layers", "dim_feedforward", "dropout", "max_seq_len",
                               "num_epochs", "quant_mode"}
    back = q.TransformerILQR(n, m * (1 + n), device="cpu").load(path)
    assert back.target_len == tf.target_len and back.prompt_len == P and back.dim_feedforward == 128
    for name, w in tf._w.items():
        assert np.array_equal(back._w[name], w), name
    for name in ("x_mean", "x_std", "u_mean", "u_std"):
        assert np.array_equal(back._norm[name], tf._norm[name])
    # early stopping restores the best state: with patience 1 and a huge learning rate the test loss goes up at once
    tf2 = q.TransformerILQR(n, m * (1 + n), prompt_len=P, d_model=128, nhead=4, num_decoder_layers=1, dim_feedforward=128,
                            dropout=0.0, max_seq_len=40, device="cpu")
    tf2.fit(data, test, num_epochs=30, batch_size=16, learning_rate=0.5, patience=1)
    assert len(tf2.test_loss_history) < 30


@pytest.mark.parametrize("model", ["cartpole", "quadrotor"])
def test_int8_checkpoint_format_and_round_trip(model, tmp_path):
    """quant_mode="int8" (SURVEY §8f rank 4): save() writes the state dict the reference writes — pinned to G12, made with
    the reference's own module and its own quantize_dynamic call: identical key set, scales, zero points and int8 weights
    (CRC) on the shipped checkpoint — and load() reads it back (weights-only loader) as fp32 weights on the int8 grid.
    The reference's int8 MODEL cannot predict under this torch (fixture `predict_error`), so there are no reference
    predictions to compare with: the dequantised model is checked against the fp32 one instead."""
    import zlib
    from quattro_ilqr_amd import TransformerILQR, training
    from oracle import transformer as o_tf
    z = load_golden(f"tf_weights_{model}.npz")
    g8 = load_golden(f"tf_int8_{model}.npz")
    g = load_golden(f"tf_{model}.npz")
    assert g8["pred_int8"].size == 0 and b"AttributeError" in bytes(g8["predict_error"])
    tf = TransformerILQR(int(z["hp.state_dim"]), int(z["hp.control_dim"]), quant_mode="int8", device="cpu")
    tf._stage = lambda: None                                # CPU container: no device staging
    tf.load(os.path.join(GOLDEN, f"tf_weights_{model}.npz"))
    tf.quant_mode = "int8"
    path = tf.save("int8", root=str(tmp_path))
    sd = torch.load(os.path.join(path, "tf_model.pt"), map_location="cpu", weights_only=True)
    assert sorted(sd.keys()) == [k.decode() for k in g8["keys"]]
    for name, scale, zp, crc in zip(g8["layers"], g8["scales"], g8["zero_points"], g8["crcs"]):
        qw, b = sd[name.decode() + "._packed_params._packed_params"]
        assert qw.dtype == torch.qint8 and qw.q_zero_point() == int(zp) and abs(qw.q_scale() - float(scale)) <= 1e-12
        assert zlib.crc32(qw.int_repr().numpy().tobytes()) == int(crc), name
        assert torch.equal(b.detach().float(), torch.tensor(z[name.decode() + ".bias"].astype(np.float32)))
    data = np.load(os.path.join(path, "tf_model_normalizer.npz"), allow_pickle=False)
    assert str(data["quant_mode"]) == "int8"
    back = TransformerILQR(1, 1, device="cpu")
    back._stage = lambda: None
    back.load(path)
    assert back.quant_mode == "int8" and back.dropout == float(z["hp.dropout"]) and back.num_epochs == tf.num_epochs
    for name in g8["layers"]:
        qw, _ = sd[name.decode() + "._packed_params._packed_params"]
        assert np.array_equal(back._w[name.decode() + ".weight"], qw.dequantize().numpy())
    for k in back._w:                                       # everything that is not a plain nn.Linear is untouched
        if not any(k.startswith(n.decode() + ".") for n in g8["layers"]):
            assert np.array_equal(back._w[k], tf._w[k]), k
    norm = {k[5:]: z[k].astype(np.float64) for k in z.files if k.startswith("norm.")}
    pred = np.array([o_tf.predict(back._w, norm, g["x_err"][i], g["prompt"][i], int(z["hp.nhead"]), int(z["hp.prompt_len"]))
                     for i in range(4)])
    assert rel_fro(pred, g["pred_fp32"][:4]) < 5e-2        # int8 weights (9 of 21 matrices): measured 1-2e-2


def test_load_save_load_round_trip_keeps_every_field(tmp_path):
    """dropout and num_epochs are restored by load() (transformer_ilqr.py:283-286), so a loaded model saves
    under the same directory name pattern and with the same normaliser file as the one it came from."""
    from quattro_ilqr_amd import TransformerILQR
    tf = TransformerILQR(4, 5, device="cpu")
    tf._stage = lambda: None
    tf.load(os.path.join(GOLDEN, "tf_weights_cartpole.npz"))
    tf.num_epochs, tf.quant_mode = 37, "float16"
    p1 = tf.save("rt", root=str(tmp_path))
    assert "_drop0.1_epoch37_" in os.path.basename(p1)
    t2 = TransformerILQR(1, 1, dropout=0.5, device="cpu")
    t2._stage = lambda: None
    t2.load(p1)
    assert (t2.dropout, t2.num_epochs, t2.quant_mode) == (0.1, 37, "float16")
    p2 = t2.save("rt2", root=str(tmp_path))
    assert os.path.basename(p1).split("_rt_")[1] == os.path.basename(p2).split("_rt2_")[1]
    a, b = (np.load(os.path.join(p, "tf_model_normalizer.npz"), allow_pickle=False) for p in (p1, p2))
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    s1, s2 = (torch.load(os.path.join(p, "tf_model.pt"), map_location="cpu", weights_only=True) for p in (p1, p2))
    assert sorted(s1) == sorted(s2) and all(torch.equal(s1[k], s2[k]) for k in s1)


DP_WORKER = """
import os, sys, json
sys.path[:0] = [{root!r}, {pkg!r}, {tests!r}]
import numpy as np, torch, torch.distributed as dist
import quattro_ilqr_amd as q
from quattro_ilqr_amd import datagen
from test_training_cpu import _toy_logs
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
n, m, N, P = 4, 1, 12, 3
xa, ka, Ka = _toy_logs(70, N, n, m, 0)                 # 54 training sequences: mini-batches of 16, the last one of 6
data = datagen.create_dataset(xa[:54], ka[:54], Ka[:54], P)
test = datagen.create_dataset(xa[54:], ka[54:], Ka[54:], P)
dev = {device!r}
tf = q.TransformerILQR(n, m * (1 + n), prompt_len=P, d_model=128, nhead=4, num_decoder_layers=1, dim_feedforward=128,
                       dropout=0.0, max_seq_len=40, device=dev)
tf.fit(data, test, num_epochs=4, batch_size=16, learning_rate=2e-3, patience=4, backend={backend!r})
out = dict(train=tf.train_loss_history, test=tf.test_loss_history, backend=tf.fit_backend,
           w=float(np.abs(tf._w["output_linear.weight"]).sum()), w1=float(tf._w["transformer_decoder.layers.0.linear1.weight"][3, 5]))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
print("RESULT" + json.dumps(out))
"""


def _run_dp_workers(tmp_path, world, device, backend):
    """world processes of DP_WORKER (gloo rendezvous on 127.0.0.1); returns the per-rank result dicts."""
    import json
    import socket
    import subprocess
    from conftest import ROOT
    script = tmp_path / f"dp_worker_{world}.py"
    script.write_text(DP_WORKER.format(root=ROOT, pkg=PKG_DIR, tests=os.path.join(ROOT, "tests"), device=device, backend=backend))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-3000:]
        outs.append(json.loads(next(l for l in so.splitlines() if l.startswith("RESULT"))[6:]))
    return outs


def test_data_parallel_fit_equals_the_single_process_fit_world2_gloo(tmp_path):
    """fit() under a two-rank process group (gloo on CPU; RCCL when the ranks are GPUs): each rank takes every second
    sequence of each mini-batch (16, 16, 16 and a ragged 6), one weighted all-reduce per step — loss histories and trained
    weights equal the single-process run on the whole mini-batches to fp32 summation order, on both ranks."""
    one = _run_dp_workers(tmp_path, 1, "cpu", "torch")[0]
    two = _run_dp_workers(tmp_path, 2, "cpu", "torch")
    assert one["backend"] == "torch"
    for r in two:
        assert np.allclose(r["train"], one["train"], rtol=2e-4), (r["train"], one["train"])
        assert np.allclose(r["test"], one["test"], rtol=2e-4)
        assert abs(r["w"] - one["w"]) < 2e-4 * abs(one["w"]) and abs(r["w1"] - one["w1"]) < 1e-4 * max(1.0, abs(one["w1"]))
    assert two[0]["train"] == two[1]["train"] and two[0]["w"] == two[1]["w"]         # the ranks stay bit-identical
