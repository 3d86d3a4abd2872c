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
