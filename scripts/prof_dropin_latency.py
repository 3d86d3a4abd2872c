"""Where the host time of a single-trajectory iLQR_TF.optimize goes (cProfile over repeated solves)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd")]
import numpy as np, torch
import quattro_ilqr_amd as q

which = sys.argv[1] if len(sys.argv) > 1 else "quadrotor"
log = (sys.argv[2] if len(sys.argv) > 2 else "log") == "log"
if which == "quadrotor":
    mpc = q.QuadrotorMPC(horizon=50, dt=0.01, integration_method="euler")
    x0 = np.zeros(12); x0[2] = 0.5; x0[6] = 0.1
else:
    mpc = q.CartPoleMPC(horizon=30, dt=0.01, integration_method="euler", ilqr_only=True)
    x0 = np.array([0.0, 0.0, 0.1, 0.0])
mpc.ilqr.enable_log = log
cold = [np.zeros_like(np.asarray(u, dtype=np.float64)) for u in mpc.ilqr.u]

def one():
    mpc.ilqr.u = [c.copy() for c in cold]
    mpc.ilqr.logs = []
    mpc.ilqr.x0 = x0
    mpc.ilqr.optimize(mpc.x_ref)

for _ in range(5):
    one()
torch.cuda.synchronize()
R = 300
t = time.perf_counter()
for _ in range(R):
    one()
print(f"{which} log={log}: {1e3 * (time.perf_counter() - t) / R:.4f} ms per solve, iterations {len(mpc.ilqr.logs) if log else len(mpc.ilqr.backward_pass_time) // (R + 5)}")
pr = cProfile.Profile()
pr.enable()
for _ in range(R):
    one()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
