"""Training-set format: host-side restatement of logs -> dataset -> normaliser -> fit() slices,
pinned to G10 fixtures produced by the reference's own _create_dataset / DataNormalizer (tests/golden/make_golden.py)."""
import os
import pickle
import sys

import numpy as np
import pytest

from conftest import PKG_DIR, load_golden

sys.path.insert(0, PKG_DIR)

REFERENCE_LOG_KEYS = ["iteration", "x_seq", "u_seq", "current_cost", "k_seq", "K_seq", "alpha", "new_x_seq", "new_u_seq",
                      "new_cost", "found_update"]           # quattro_ilqr_tf.py:453-466, in order


@pytest.mark.parametrize("model", ["cartpole", "quadrotor"])
def test_dataset_normaliser_and_slices_match_the_reference(model):
    from quattro_ilqr_amd import datagen
    g = load_golden(f"dataset_{model}.npz")
    P = int(g["prompt_len"])
    x_data, kK_data = datagen.create_dataset(g["log_x_seq"], g["log_k_seq"], g["log_K_seq"], P)
    assert x_data.dtype == np.float32 and kK_data.dtype == np.float32
    assert np.array_equal(x_data, g["x_data"]) and np.array_equal(kK_data, g["kK_data"])
    # the (m, 1+n) flattening: column 0 of each control's block is k, the rest its row of K  (F7)
    m, n = g["log_K_seq"].shape[2:]
    blk = kK_data.reshape(kK_data.shape[0], kK_data.shape[1], m, 1 + n)
    assert np.array_equal(blk[..., 0], g["log_k_seq"].astype(np.float32))
    assert np.array_equal(blk[..., 1:], g["log_K_seq"].astype(np.float32))
    norm = datagen.fit_normalizer(x_data, kK_data)
    for key in ("x_mean", "x_std", "u_mean", "u_std"):
        assert np.array_equal(norm[key], g[key]), key
    x_norm, u_prompt, u_target = datagen.training_slices(x_data, kK_data, norm, P)
    assert np.array_equal(x_norm, g["x_norm"], equal_nan=True)
    assert np.array_equal(u_prompt, g["u_prompt"]) and np.array_equal(u_target, g["u_target"])
    assert u_target.shape[1] == int(g["target_len"])
    # too-short sequences are dropped (transformer_ilqr.py:87-88)
    xs, ks = datagen.create_dataset(g["log_x_seq"][:, :P + 1], g["log_k_seq"][:, :P], g["log_K_seq"][:, :P], P)
    assert xs.shape[0] == 0 and ks.shape[0] == 0


def _synthetic_log(E=5, N=6, n=4, m=1, seed=0):
    from quattro_ilqr_amd.datagen import IterationLog
    r = np.random.default_rng(seed)
    found = np.array([True, True, False, True, False])[:E]
    new_x = r.standard_normal((E, N + 1, n)).astype(np.float32); new_x[~found] = np.nan
    new_u = r.standard_normal((E, N, m)).astype(np.float32); new_u[~found] = np.nan
    return IterationLog(traj=np.array([0, 0, 0, 1, 1], dtype=np.int32)[:E], iteration=np.array([0, 1, 2, 0, 1], dtype=np.int32)[:E],
                        x_seq=r.standard_normal((E, N + 1, n)).astype(np.float32),
                        u_seq=r.standard_normal((E, N, m)).astype(np.float32), current_cost=r.uniform(1, 2, E),
                        k_seq=r.standard_normal((E, N, m)).astype(np.float32),
                        K_seq=r.standard_normal((E, N, m, n)).astype(np.float32),
                        alpha=np.where(found, 0.5, np.nan), new_x_seq=new_x, new_u_seq=new_u,
                        new_cost=np.where(found, 0.9, np.nan), found_update=found)


def test_pickle_stream_has_the_reference_entry_format(tmp_path):
    from quattro_ilqr_amd import datagen
    log = _synthetic_log()
    path = os.path.join(tmp_path, "mpc_logs_combined.pkl")
    assert datagen.write_pickle_stream(path, log.select(slice(0, 3)), append=False) == 3
    assert datagen.write_pickle_stream(path, log.select(slice(3, 5))) == 2          # appended, like the flushes at :205
    entries = []
    with open(path, "rb") as fh:            # the reader loop of transformer_training.py:17-23 (our own file)
        while True:
            try:
                entries.append(pickle.load(fh))
            except EOFError:
                break
    assert len(entries) == 5
    for e, entry in enumerate(entries):
        assert list(entry.keys()) == REFERENCE_LOG_KEYS
        assert entry["x_seq"].shape == (7, 4) and entry["x_seq"].dtype == np.float64
        assert isinstance(entry["k_seq"], list) and len(entry["k_seq"]) == 6 and entry["k_seq"][0].shape == (1,)
        assert isinstance(entry["K_seq"], list) and entry["K_seq"][0].shape == (1, 4)
        assert isinstance(entry["u_seq"], list) and entry["u_seq"][0].shape == (1,)
        assert entry["iteration"] == int(log.iteration[e]) and entry["found_update"] == bool(log.found_update[e])
        if entry["found_update"]:
            assert entry["alpha"] == 0.5 and entry["new_x_seq"].shape == (7, 4) and len(entry["new_u_seq"]) == 6
        else:
            assert entry["alpha"] is None and entry["new_x_seq"] is None and entry["new_u_seq"] is None and entry["new_cost"] is None
    # a DataFrame of these entries is what _create_dataset consumes: same arrays through our restatement
    x_data, kK_data = datagen.create_dataset(np.array([e["x_seq"] for e in entries]), np.array([np.array(e["k_seq"]) for e in entries]),
                                             np.array([np.array(e["K_seq"]) for e in entries]), prompt_len=2)
    x2, kK2 = datagen.create_dataset(log.x_seq, log.k_seq, log.K_seq, prompt_len=2)
    assert np.array_equal(x_data, x2) and np.array_equal(kK_data, kK2)


def test_npz_round_trip(tmp_path):
    from quattro_ilqr_amd import datagen
    log = _synthetic_log()
    path = os.path.join(tmp_path, "logs.npz")
    datagen.write_npz(path, log)
    back = datagen.read_npz(path)
    for f in ("traj", "iteration", "x_seq", "u_seq", "current_cost", "k_seq", "K_seq", "alpha", "new_x_seq", "new_u_seq",
              "new_cost", "found_update"):
        assert np.array_equal(getattr(back, f), getattr(log, f), equal_nan=True), f
