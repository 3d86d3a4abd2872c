"""Container-side calibration of the CPU baseline: one quadrotor N = 50 trajectory, 5 iLQR
iterations, timed (a) with the oracle's restatement (oracle/ilqr.py, what bench.py's cpu_baseline leg runs on the GPU box)
and (b) with the REFERENCE itself imported from /root/reference (build container only) on the same inputs.  Prints both
single-core rates and their ratio; tests/test_oracle_golden.py asserts the ratio stays within +-20 %.

Measured in this container (2026-10, 8 shared cores): oracle 55.4, reference 55.3 steps/s/core (ratio 1.00); on other
occasions 62-78 steps/s for both with ratios 0.97-1.03 (the host's speed drifts, hence the alternating rounds below).  SURVEY §6.2
quotes 94 steps/s/core for the reference when the survey was taken: the absolute figure moves with the host (the GPU
box's cores do 175), the ratio does not."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def inputs(seed=9001, N=50):
    rng = np.random.default_rng(seed)
    x_ref = np.zeros(12); x_ref[2] = 0.5
    x0 = x_ref + rng.uniform(-1.0, 1.0, 12) * np.array([0.5, 0.5, 0.01, 0, 0, 0, 0.2, 0.2, 0.5, 0, 0, 0])
    u = [2.4525 + 0.1 * rng.standard_normal(4) for _ in range(N)]
    return x_ref, x0, u


def oracle_once(iters=5, N=50):
    from oracle import ilqr as o_ilqr, models as o_models
    spec = o_models.quadrotor_spec()
    _, x0, u = inputs()
    t0 = time.perf_counter()
    for _ in range(iters):
        u, _, _ = o_ilqr.optimize(spec.f, spec.L, spec.Lf, x0, u, N, max_iter=1, tol=-1.0, keep_logs=True)
    return iters * N / (time.perf_counter() - t0)


def reference_once(iters=5, N=50):
    sys.dont_write_bytecode = True
    for pth in (REF, os.path.join(REF, "examples/quadrotor")):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    from quadrotor_mpc import QuadrotorMPC
    m = QuadrotorMPC(horizon=N, dt=0.01, integration_method="euler")
    x_ref, x0, u = inputs()
    m.ilqr.x0, m.ilqr.u, m.ilqr.max_iter, m.ilqr.tol, m.ilqr.logs = x0, u, iters, -1.0, []
    t0 = time.perf_counter()
    m.ilqr.optimize(m.x_ref)
    return len(m.ilqr.logs) * N / (time.perf_counter() - t0)


def oracle_rate(reps=3):
    return max(oracle_once() for _ in range(reps))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        print(f"oracle    : {oracle_rate():.1f} steps/s on one core")
        sys.exit(0)
    # the two are timed ALTERNATELY (the shared host's speed drifts by tens of per cent over a minute: two blocks timed one
    # after the other compare the host with itself), best of the rounds each, ratio = median of the per-round ratios
    o_all, r_all = [], []
    for _ in range(4):
        o_all.append(oracle_once())
        r_all.append(reference_once())
    o, r = max(o_all), max(r_all)
    ratio = float(np.median([a / b for a, b in zip(o_all, r_all)]))
    print(f"oracle    : {o:.1f} steps/s on one core")
    print(f"reference : {r:.1f} steps/s on one core   (oracle / reference = {ratio:.3f})")
