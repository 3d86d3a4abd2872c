"""Median launch time of the predictor kernel alone (B = 4096, quadrotor architecture, random-init weights)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import torch
from quattro_ilqr_amd import TransformerILQR
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16"
tf = TransformerILQR.random_init(12, 52, prompt_len=1, target_len=49, device=dev, precision=PREC)
x = torch.randn(B, 51, 12, device=dev); p = torch.randn(B, 1, 52, device=dev)
for _ in range(5):
    tf.predict_batch(x, p)
torch.cuda.synchronize()
ts = []
for _ in range(40):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); tf.predict_batch(x, p); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
# gains mode (what the solver calls): the prediction unpacked straight into K / k
K = torch.zeros(B, 50, 4, 12, device=dev); k = torch.zeros(B, 50, 4, device=dev); act = torch.ones(B, dtype=torch.int32, device=dev)
tg = []
for _ in range(45):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); tf.predict_gains(x, p, K, k, act); b.record(); torch.cuda.synchronize()
    tg.append(a.elapsed_time(b))
tg = np.array(tg[5:])
print(f"   gains mode: median {np.median(tg)*1e3:.1f} us  min {tg.min()*1e3:.1f}")
ts = np.array(ts)
print(f"{os.environ.get('QUATTRO_HIP_LIB', 'default')} {PREC}: B={B} median {np.median(ts)*1e3:.1f} us  min {ts.min()*1e3:.1f}  "
      f"({135.64e6*B/np.median(ts)/1e-3/1e12/2500*100:.1f}% of 2.5 PF)")
