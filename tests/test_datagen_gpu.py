"""collect(): the batched GPU solver emits the same per-iteration log entries the reference's optimize() logged for the
same initial states (G10 fixture), and the dataset / normaliser built from them agree with the reference's."""
import sys

import numpy as np
import pytest

from conftest import PKG_DIR, load_golden, rel_fro

pytestmark = pytest.mark.gpu
sys.path.insert(0, PKG_DIR)
DEV = "cuda:0"


@pytest.mark.parametrize("model,N", [("cartpole", 30), ("quadrotor", 50)])
def test_collected_logs_reproduce_the_reference_logs(model, N):
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import datagen
    g = load_golden(f"dataset_{model}.npz")
    md = q.model_by_name(model)
    tol = 1e-1 if model == "cartpole" else 1e-3            # cartpole_mpc.py:215 / quadrotor_mpc.py:59
    solver = q.QuattroILQR(md, N, max_iter=int(g["max_iter"]), tol=tol, device=DEV)
    log = datagen.collect(solver, g["x0"])
    # same entries, in the same order: trajectory by trajectory, iteration by iteration
    ref_it = g["log_iteration"]
    assert len(log) == ref_it.shape[0]
    assert np.array_equal(log.iteration, ref_it)
    ref_traj = np.cumsum(ref_it == 0) - 1
    assert np.array_equal(log.traj, ref_traj)
    assert rel_fro(log.x_seq, g["log_x_seq"]) < 2e-5
    # the reference's K carries its finite-difference noise (1e-5..1e-4 relative): bound, do not match
    assert rel_fro(log.K_seq, g["log_K_seq"]) < 5e-4 and rel_fro(log.k_seq, g["log_k_seq"]) < 5e-4
    # internal consistency of an entry: new_x_seq of iteration i is x_seq of iteration i+1 of the same trajectory
    for e in range(len(log) - 1):
        if log.traj[e] == log.traj[e + 1]:
            assert log.found_update[e]
            assert np.array_equal(log.new_x_seq[e], log.x_seq[e + 1])
            assert log.new_cost[e] == log.current_cost[e + 1] and log.new_cost[e] <= log.current_cost[e]
    P = int(g["prompt_len"])
    x_data, kK_data = datagen.create_dataset(log.x_seq, log.k_seq, log.K_seq, P)
    assert x_data.shape == g["x_data"].shape and kK_data.shape == g["kK_data"].shape
    norm = datagen.fit_normalizer(x_data, kK_data)
    assert np.allclose(norm["x_mean"], g["x_mean"], rtol=1e-4, atol=1e-5) and np.allclose(norm["x_std"], g["x_std"], rtol=1e-4, atol=1e-5)
    assert np.allclose(norm["u_mean"], g["u_mean"], rtol=2e-3, atol=1e-4) and np.allclose(norm["u_std"], g["u_std"], rtol=2e-3, atol=1e-4)


def test_collect_large_batch_counts_and_masks():
    import torch
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import datagen
    md = q.cartpole_model()
    rng = np.random.default_rng(3)
    B, N = 513, 30
    x0 = np.stack([rng.uniform(-0.5, 0.5, B), np.zeros(B), rng.uniform(-0.5, 0.5, B), np.zeros(B)], axis=1)
    solver = q.QuattroILQR(md, N, max_iter=8, tol=1e-1, device=DEV)
    log = datagen.collect(solver, x0)
    ref = q.QuattroILQR(md, N, max_iter=8, tol=1e-1, device=DEV).solve(x0)
    iters = ref["iters"].cpu().numpy()
    assert len(log) == int(iters.sum())
    assert np.array_equal(np.bincount(log.traj, minlength=B), iters)
    assert torch.equal(solver.x, ref["x"]) and torch.equal(solver.u, ref["u"])       # recording does not perturb the solve
    none = ~log.found_update
    assert np.isnan(log.alpha[none]).all() and np.isnan(log.new_cost[none]).all() and np.isnan(log.new_x_seq[none]).all()


def test_collect_in_trajectory_slices_equals_one_ring():
    """A ring budget smaller than the batch's log: the batch is collected in slices of trajectories; same entries, same order."""
    import quattro_ilqr_amd as q
    from quattro_ilqr_amd import datagen
    md = q.cartpole_model()
    N, B = 30, 37
    rng = np.random.default_rng(4)
    x0 = np.zeros((B, 4)); x0[:, 0] = rng.uniform(-0.5, 0.5, B); x0[:, 2] = rng.uniform(-0.5, 0.5, B)
    sv = q.QuattroILQR(md, N, max_iter=5, tol=1e-1, device="cuda:0")
    one = datagen.collect(sv, x0)
    probe = q.ops.SolveLog(md, N, 1, 1, "cuda:0")
    sliced = datagen.collect(sv, x0, ring_bytes=5 * probe.rec_bytes * 8)            # eight trajectories per slice
    assert len(one) == len(sliced) > B
    for name in ("traj", "iteration", "x_seq", "u_seq", "K_seq", "k_seq", "current_cost", "found_update"):
        assert np.array_equal(getattr(one, name), getattr(sliced, name)), name
    assert np.array_equal(np.isnan(one.new_cost), np.isnan(sliced.new_cost))
    assert np.all(np.diff(one.traj) >= 0)                                            # trajectory-major, iterations ascending inside
