"""Diagnostic (GPU box): which part of the predictor kernel disagrees with the fp64 oracle?  Ablates groups of
parameters on both sides and prints the relative error of each variant."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
from oracle import transformer as o_tf
from quattro_ilqr_amd import TransformerILQR

model = sys.argv[1] if len(sys.argv) > 1 else "quadrotor"
z = np.load(os.path.join(ROOT, "tests", "golden", f"tf_weights_{model}.npz"))
g = np.load(os.path.join(ROOT, "tests", "golden", f"tf_{model}.npz"))
W0 = {k: z[k].astype(np.float32) for k in z.files if not k.startswith(("norm.", "hp."))}
norm = {k[5:]: z[k].astype(np.float64) for k in z.files if k.startswith("norm.")}
hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
n, c = hp["state_dim"], hp["control_dim"]

def run(W, tag, x=None, pr=None):
    x = g["x_err"][0] if x is None else x
    pr = g["prompt"][0] if pr is None else pr
    tf = TransformerILQR(n, c, device="cuda:0").load_arrays(W, norm, hp)
    got = tf.predict(x, pr)
    want = o_tf.predict(W, norm, x, pr, hp["nhead"], hp["prompt_len"])
    e = np.linalg.norm(got - want) / np.linalg.norm(want)
    rows = np.linalg.norm(got - want, axis=1) / (np.linalg.norm(want, axis=1) + 1e-30)
    print(f"{tag:28s} rel err {e:.3e}   worst rows {np.argsort(rows)[-3:]} {np.sort(rows)[-3:]}")

def zero(W, pat, sl=None):
    W = dict(W)
    for k in W:
        if pat(k):
            a = W[k].copy()
            if sl is None: a[...] = 0
            else: a[sl] = 0
            W[k] = a
    return W

d = hp["d_model"]
run(W0, "original")
run(zero(W0, lambda k: k.endswith("in_proj_bias"), slice(2 * d, 3 * d)), "b_v = 0")
run(zero(W0, lambda k: k.endswith("in_proj_bias"), slice(d, 2 * d)), "b_k = 0")
run(zero(W0, lambda k: k.endswith("in_proj_bias")), "b_qkv = 0")
run(zero(W0, lambda k: k.endswith("out_proj.bias")), "b_o = 0")
run(zero(W0, lambda k: k.endswith("linear2.bias")), "b_2 = 0")
run(zero(W0, lambda k: k.endswith("linear1.bias")), "b_1 = 0")
run(zero(W0, lambda k: "norm" in k and k.endswith("bias")), "LN beta = 0")
run(zero(W0, lambda k: k.endswith(("embed.bias",)) or k == "target_embedding"), "embed biases = 0")
run(W0, "zero prompt", pr=0 * g["prompt"][0] + norm["u_mean"])
run(W0, "zero state", x=0 * g["x_err"][0] + norm["x_mean"])
one = zero(W0, lambda k: "layers.1." in k or "layers.2." in k)
run(one, "layers 1,2 zeroed")
