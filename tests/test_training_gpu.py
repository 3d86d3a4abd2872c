"""generate -> train -> deploy on one GPU: the batched solver's logs train a predictor with
TransformerILQR.fit (the device training step of csrc/tf_train.hip by default), and the trained weights run in the HIP inference kernel and in the hybrid
solver."""
import sys

import numpy as np
import pytest

from conftest import PKG_DIR, rel_fro

pytestmark = pytest.mark.gpu
sys.path.insert(0, PKG_DIR)
DEV = "cuda:0"


@pytest.mark.parametrize("model", ["cartpole", "quadrotor"])
def test_training_forward_on_rocm_matches_the_reference_module(model):
    """The training-time forward (torch ops on ROCm: rocBLAS GEMMs) is pinned to the REFERENCE module's fp32 predictions
    on the shipped checkpoints (G7) on the GPU too — so comparing the HIP inference kernel with it on freshly trained
    weights (below) is a comparison with the reference's arithmetic, not with ourselves."""
    import torch
    from conftest import load_golden
    from quattro_ilqr_amd import training
    z = load_golden(f"tf_weights_{model}.npz")
    g = load_golden(f"tf_{model}.npz")
    W = {k: torch.tensor(z[k].astype(np.float32), device=DEV) for k in z.files if not k.startswith(("norm.", "hp."))}
    norm = {k[5:]: z[k].astype(np.float64) for k in z.files if k.startswith("norm.")}
    nhead, P = int(z["hp.nhead"]), int(z["hp.prompt_len"])
    buffers = {"pos_encoder.pe": W.pop("pos_encoder.pe")}
    xn = torch.tensor(((g["x_err"] - norm["x_mean"]) / norm["x_std"]).astype(np.float32), device=DEV)
    un = torch.tensor(((g["prompt"] - norm["u_mean"]) / norm["u_std"]).astype(np.float32), device=DEV)[:, -P:]
    with torch.no_grad():
        out = training.forward(W, buffers, xn, un, nhead).double().cpu().numpy()
    assert rel_fro(out * norm["u_std"] + norm["u_mean"], g["pred_fp32"]) < 1e-5


def test_collect_fit_predict_and_hybrid_solve():
    import torch
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import datagen, training
    md = q.cartpole_model()
    N, P = 30, 5
    rng = np.random.default_rng(2)
    B = 96
    x0 = np.stack([rng.uniform(-0.5, 0.5, B), np.zeros(B), rng.uniform(-0.5, 0.5, B), np.zeros(B)], axis=1)
    log = datagen.collect(q.QuattroILQR(md, N, max_iter=6, tol=1e-1, device=DEV), x0)
    assert len(log) > B
    n_train = int(0.8 * len(log))
    perm = rng.permutation(len(log))
    tf = q.TransformerILQR(4, 5, prompt_len=P, d_model=128, nhead=4, num_decoder_layers=2, dim_feedforward=256,
                           dropout=0.0, max_seq_len=80, device=DEV)
    tf.fit(log.select(perm[:n_train]), log.select(perm[n_train:]), num_epochs=25, batch_size=32, learning_rate=1e-3,
           patience=25)
    assert tf.target_len == N + 1 - P
    assert tf.train_loss_history[-1] < 0.3 * tf.train_loss_history[0]
    assert tf.test_loss_history[-1] < 0.5 * tf.test_loss_history[0]
    # the HIP kernel (bf16 MFMA) reproduces the training-time forward (fp32 torch) on the trained weights
    x_data, kK_data = datagen.create_dataset(log.x_seq, log.k_seq, log.K_seq, P)
    x_err = torch.as_tensor(x_data[:32], device=DEV).contiguous()
    prompt = torch.as_tensor(kK_data[:32, -P:], device=DEV).contiguous()
    pred = tf.predict_batch(x_err, prompt).double().cpu().numpy()
    W = {k: torch.as_tensor(v, device=DEV) for k, v in tf._w.items()}
    buf = {"pos_encoder.pe": W.pop("pos_encoder.pe")}
    nm = {k: torch.as_tensor(v, dtype=torch.float32, device=DEV) for k, v in tf._norm.items()}
    with torch.no_grad():
        ref = training.forward(W, buf, (x_err - nm["x_mean"]) / nm["x_std"], (prompt - nm["u_mean"]) / nm["u_std"], tf.nhead)
        ref = (ref * nm["u_std"] + nm["u_mean"]).double().cpu().numpy()
    assert rel_fro(pred, ref) < 2e-2                        # bf16 operands, fp32 accumulation
    # and the hybrid solver runs with it: gains for t < N - P from the predictor, the last P steps from the sweep; the
    # predictor's stack is one row longer than needed (fitted on N + 1 state rows) and the tail is ignored like the
    # reference does
    hyb = q.QuattroILQR(md, N, max_iter=4, tol=1e-1, tf=tf, device=DEV)
    out = hyb.solve(x0[:16])
    assert bool(torch.isfinite(out["cost"]).all()) and int((out["iters"] >= 1).sum()) == 16
    x_nom, cost0 = q.ops.simulate(md, torch.as_tensor(x0[:16], dtype=torch.float32, device=DEV),
                                  torch.zeros((16, N, 1), device=DEV))
    assert bool((out["cost"] <= cost0).all())               # accepted steps never increase the cost
