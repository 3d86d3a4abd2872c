"""Randomised comparison of the fused one-launch transformer kernel (bf16 / fp16 operands) with the layer-wise fp32 kernels on the
same random weights: random sequence compositions (state tokens, prompt length, target length, L <= 128), feed-forward widths,
layer counts, state / control dimensions and batch sizes; the gains-mode entry (prediction unpacked into K, k) against the plain one.
usage: fuzz_transformer.py [seconds] [seed] [max_cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else None       # (tests/test_fuzz_gpu.py runs a fixed-size, fixed-seed slice)
rng = np.random.default_rng(seed)
DEV = torch.device("cuda:0")
t_end = time.time() + budget
n_cases, fails, worst = 0, [], 0.0
while time.time() < t_end and (max_cases is None or n_cases < max_cases):
    n, m = [(12, 4), (4, 1), (6, 2), (3, 1), (10, 5)][int(rng.integers(0, 5))]
    c = m * (1 + n)
    if c > 64:
        continue
    L = int(rng.integers(4, 129))
    P = int(rng.integers(1, min(10, L - 2) + 1))
    T = int(rng.integers(1, L - P - 1 + 1))
    NS = L - P - T
    if NS < 1:
        continue
    ff = 64 * int(rng.integers(1, 17))
    layers = int(rng.integers(1, 5))
    B = int(rng.choice([1, 2, 5, 33, 256, 700]))
    prec = "bf16" if rng.random() < 0.7 else "fp16"
    tf = q.TransformerILQR.random_init(n, c, prompt_len=P, target_len=T, d_model=128, nhead=4, num_decoder_layers=layers,
                                       dim_feedforward=ff, max_seq_len=130, seed=int(rng.integers(0, 1 << 30)), device=DEV, precision=prec)
    assert tf.fused_kernel_covers()
    x = torch.as_tensor(rng.standard_normal((B, NS, n)), dtype=torch.float32, device=DEV).contiguous()
    pr = torch.as_tensor(rng.standard_normal((B, P, c)), dtype=torch.float32, device=DEV).contiguous()
    fused = tf.predict_batch(x, pr).double()
    ref = tf._predict_fp32(x, pr).double()
    err = float((fused - ref).norm() / ref.norm())
    worst = max(worst, err)
    bad = []
    if not (err < 3e-2) or not bool(torch.isfinite(fused).all()):
        bad.append(f"fused vs fp32 {err:.2e}")
    # gains mode: the same prediction unpacked by the kernel into K (B, N, m, n), k (B, N, m) for a horizon N = T + P
    N = T + P
    K = torch.full((B, N, m, n), -7.0, device=DEV); k = torch.full((B, N, m), -7.0, device=DEV)
    act = torch.as_tensor((rng.random(B) < 0.7).astype(np.int32), device=DEV)
    tf.predict_gains(x, pr, K, k, act)
    rows = tf.predict_batch(x, pr).view(B, T, m, 1 + n)
    live = act.bool()
    if bool(live.any()) and not (torch.equal(k[live][:, :T], rows[live][..., 0]) and torch.equal(K[live][:, :T], rows[live][..., 1:])):
        bad.append("gains mode differs from the plain prediction")
    if bool((~live).any()) and not bool((K[~live] == -7.0).all()):
        bad.append("gains mode wrote an inactive trajectory")
    if not bool((K[:, T:] == -7.0).all()):
        bad.append("gains mode wrote beyond the predicted rows")
    n_cases += 1
    if bad:
        fails.append((n, m, NS, P, T, ff, layers, B, prec, bad))
        if len(fails) <= 20:
            print("MISMATCH", fails[-1], flush=True)
    if n_cases % 100 == 0:
        print(f"{n_cases} cases, {len(fails)} mismatches, worst fused-vs-fp32 {worst:.2e}", flush=True)
print(f"done: {n_cases} cases, {len(fails)} mismatches, worst fused-vs-fp32 rel-Fro {worst:.2e} (seed {seed})")
sys.exit(1 if fails else 0)
