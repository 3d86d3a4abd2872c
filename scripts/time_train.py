"""Device training step (csrc/tf_train.hip) vs torch autograd (rocBLAS) on the shipped quadrotor predictor shape:
milliseconds per mini-batch (forward + loss + backward + Adam) and the worst relative gradient error, on one GPU."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "quattro-transformer-ilqr_amd"))
from quattro_ilqr_amd import train_hip, training  # noqa: E402

DEV = "cuda:0"
n, c, d, H, layers, ff, NS, P, T = 12, 52, 128, 4, 3, 512, 51, 1, 49
for B in (16, 64, 256, 512):
    params, buffers = training.init_params(n, c, d, H, layers, ff, 110, T, seed=0, device=DEV)
    tr = train_hip.HipTrainer(n, c, d, H, layers, ff, NS, P, T, 0.0, buffers["pos_encoder.pe"].cpu().numpy(), DEV)
    tr.load_state_dict({k: v.detach() for k, v in params.items()})
    g = torch.Generator().manual_seed(1)
    x, u, y = (torch.randn(s, generator=g).to(DEV) for s in ((B, NS, n), (B, P, c), (B, T, c)))
    tr.forward_backward(x, u, y)
    loss = F.mse_loss(training.forward(params, buffers, x, u, H), y)
    loss.backward()
    worst = max(float((tr.view(tr.grads, k).double() - v.grad.double()).norm() / v.grad.double().norm()) for k, v in params.items())
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)

    def hip_step():
        tr.forward_backward(x, u, y)
        tr.adam_step()

    def torch_step():
        opt.zero_grad(set_to_none=True)
        F.mse_loss(training.forward(params, buffers, x, u, H), y).backward()
        opt.step()

    out = {}
    for name, fn in (("hip", hip_step), ("torch", torch_step)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        out[name] = 1e3 * (time.perf_counter() - t0) / 20
    flops = 3 * 2 * B * (NS + P + T) * (layers * (4 * d * d + 2 * d * ff) + n * d) + 3 * 2 * B * T * d * c
    print(f"B={B:4d}  hip {out['hip']:.3f} ms  torch {out['torch']:.3f} ms  worst grad rel err {worst:.2e}  "
          f"(linear-layer work {flops / 1e9:.1f} GFLOP -> {flops / out['hip'] / 1e9:.1f} TFLOP/s in the hip step)")
