"""Compile (hipcc, no GPU needed) every user-model library the GPU tests and the bench ask for, so that the in-tree cache
(quattro_ilqr_amd/_user_models/, git-ignored but part of the gpurun snapshot) travels to the GPU box ready to load."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "quattro-transformer-ilqr_amd"), os.path.join(ROOT, "tests")]
import quattro_ilqr_amd as q
from quattro_ilqr_amd import user_model
import test_user_model_gpu as t

built = [user_model.example_planar_model().lib_path, t.planar_model("rk4").lib_path]
b = q.quadrotor_model()
built.append(q.compile_model("quadrotor_user", 12, 4, rate=t.QUAD_RATE, dt=b.dt, integrator="euler", phys=b.phys, q=b.q, r=b.r, qf=b.qf,
                             x_ref=b.x_ref, barrier_alpha=b.barrier_alpha, barrier_beta=b.barrier_beta).lib_path)
for W in (5, 8):          # tests/test_user_model_gpu.py::test_user_model_at_the_largest_dimensions (the library depends on the rate body and (n, m) only)
    built.append(q.compile_model(f"chain{2 * W}x{W}", 2 * W, W, rate=t.CHAIN_RATE, dt=0.02, integrator="rk4", phys=(2.0, 0.3, 1.5)).lib_path)
print("\n".join(sorted(set(built))))
# entries of earlier source / flag states (the cache key covers both) are dead weight in every snapshot sent to a GPU box
import shutil
keep = {os.path.basename(os.path.dirname(b)) for b in built}
cache = os.path.dirname(os.path.dirname(built[0]))
if os.path.basename(cache) == "_user_models":
    for e in os.listdir(cache):
        if e not in keep and not e.startswith("pendulum_cpu_test_") and os.path.isdir(os.path.join(cache, e)):     # (tests/test_host_cpu.py's own)
            shutil.rmtree(os.path.join(cache, e))
            print("pruned", e)
