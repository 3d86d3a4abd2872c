"""Training-set writer: the batched GPU solver as the data generator.

The reference produces its transformer training data with ten CPU processes, each running MPC episodes and appending
`ilqr.logs` — one dict per iLQR iteration, quattro_ilqr_tf.py:453-466 — to a pickle stream
(examples/quadrotor/training/training_data_collection.py:196-214, combined at :265-290); training then reads the
stream back (transformer_training.py:9-42), stacks `x_seq` and `[k | K]` per entry
(TransformerILQR._create_dataset, transformer_ilqr.py:70-92) and fits the normaliser (transformer_model.py:27-31).

Here one `collect()` call runs B independent solves on the device and returns the same per-iteration entries as
arrays (`IterationLog`); `write_pickle_stream` emits the reference's on-disk format (a sequence of pickled dicts, one
per iteration), `write_npz` a plain-array equivalent that loads without unpickling, and `create_dataset` /
`fit_normalizer` / `training_slices` restate the three host-side steps that turn logs into training tensors.

Layout quirk kept on purpose: the DATASET flattens each step as (m, 1+n) — [k_i, K_i,:] per control —
while the PROMPT the solver feeds the predictor is [k | K.flat].  Both are what the reference does.
"""
import pickle
from dataclasses import dataclass, fields

import numpy as np
import torch

from . import ops

ALPHA_NONE = float("nan")


@dataclass
class IterationLog:
    """E log entries (one per trajectory per iLQR iteration, ordered by trajectory then iteration).  Field names follow
    the reference's log dict; `traj` is the index of the trajectory in the collected batch."""
    traj: np.ndarray            # (E,)  int32
    iteration: np.ndarray       # (E,)  int32
    x_seq: np.ndarray           # (E, N+1, n) nominal of the iteration
    u_seq: np.ndarray           # (E, N, m)   controls AFTER the accept (the reference logs the updated u_seq, :448-457)
    current_cost: np.ndarray    # (E,)  fp64
    k_seq: np.ndarray           # (E, N, m)
    K_seq: np.ndarray           # (E, N, m, n)
    alpha: np.ndarray           # (E,)  accepted step, NaN if none (reference: None)
    new_x_seq: np.ndarray       # (E, N+1, n) accepted candidate, NaN if none
    new_u_seq: np.ndarray       # (E, N, m)
    new_cost: np.ndarray        # (E,)  fp64, NaN if none
    found_update: np.ndarray    # (E,)  bool

    def __len__(self):
        return int(self.traj.shape[0])

    def select(self, idx):
        return IterationLog(**{f.name: getattr(self, f.name)[idx] for f in fields(self)})


def collect(solver, x0, u_init=None, max_iter=None, ring_bytes=1 << 30):
    """Run solver (a pure-mode QuattroILQR) on the batch x0 (B, n) and record every iteration of every trajectory that
    was still iterating, exactly the entries iLQR_TF.optimize would append to `logs` for each of them.

    Round 4: the entries are written by the DEVICE — the solve runs as usual (one persistent launch where the model has such a
    kernel) with a per-iteration log ring attached (ops.SolveLog, csrc/solve_log.h), and the host reads the ring once at the end:
    no synchronisation and no copy per iteration (rounds 2-3 copied every iteration's K out between two launches).  Batches
    whose ring would exceed `ring_bytes` are collected in slices of trajectories."""
    if solver.tf is not None:
        raise ValueError("training data comes from the pure iLQR solver (tf=None), as in the reference's collection runs")
    md, N, dev = solver.model, solver.horizon, solver.device
    n, m = md.n, md.m
    x0_t = torch.as_tensor(np.asarray(x0) if not isinstance(x0, torch.Tensor) else x0, dtype=torch.float32,
                           device=dev).reshape(-1, n).contiguous()
    B = x0_t.shape[0]
    u_t = None if u_init is None else torch.as_tensor(u_init, dtype=torch.float32, device=dev).reshape(B, N, m).contiguous()
    max_iter = solver.max_iter if max_iter is None else int(max_iter)
    cap = max(1, max_iter)
    probe = ops.SolveLog(md, N, 1, 1, dev)
    per_traj = cap * probe.rec_bytes
    slice_B = max(1, min(B, int(ring_bytes) // per_traj))
    alphas = np.asarray(solver.alphas, dtype=np.float64)
    chunks = {f.name: [] for f in fields(IterationLog)}
    ring = None
    for b0 in range(0, B, slice_B):
        Bs = min(slice_B, B - b0)
        if ring is None or ring.B != Bs:
            ring = ops.SolveLog(md, N, Bs, cap, dev)
        out = solver.solve(x0_t[b0:b0 + Bs], None if u_t is None else u_t[b0:b0 + Bs], max_iter=max_iter, log=ring, want_alpha=False)
        iters = out["iters"].cpu().numpy().astype(np.int64)
        x_fin, u_fin = out["x"].cpu().numpy(), out["u"].cpu().numpy()
        rec = ring.decode(ring.buf.cpu().numpy(), Bs * cap)
        shp = lambda a: a.reshape((Bs, cap) + a.shape[1:])
        rx, ru, rK, rk = shp(rec["x"]), shp(rec["u"]), shp(rec["K"]), shp(rec["k"])
        rcost, raidx = shp(rec["cost"]), shp(rec["alpha_idx"])
        bi, ii = np.nonzero(np.arange(cap)[None, :] < iters[:, None])           # entries (trajectory, iteration), trajectory-major
        found = raidx[bi, ii] >= 0
        last = ii + 1 >= iters[bi]                                               # the accepted candidate of a trajectory's last
        nxt = np.minimum(ii + 1, cap - 1)                                        # iteration is its result, else the next nominal
        new_x = np.where(last[:, None, None], x_fin[bi], rx[bi, nxt])
        new_u = np.where(last[:, None, None], u_fin[bi], ru[bi, nxt])
        u_after = np.where(found[:, None, None], new_u, ru[bi, ii])             # the reference logs the UPDATED u_seq (:448-457)
        new_x[~found] = np.nan
        new_u[~found] = np.nan
        chunks["traj"].append((b0 + bi).astype(np.int32))
        chunks["iteration"].append(ii.astype(np.int32))
        chunks["x_seq"].append(rx[bi, ii])
        chunks["u_seq"].append(u_after)
        chunks["current_cost"].append(rcost[bi, ii, 0])
        chunks["k_seq"].append(rk[bi, ii])
        chunks["K_seq"].append(rK[bi, ii])
        chunks["alpha"].append(np.where(found, alphas[np.clip(raidx[bi, ii], 0, None)], ALPHA_NONE))
        chunks["new_x_seq"].append(new_x)
        chunks["new_u_seq"].append(new_u)
        chunks["new_cost"].append(np.where(found, rcost[bi, ii, 1], np.nan))
        chunks["found_update"].append(found)
    return IterationLog(**{k: np.concatenate(v, axis=0) for k, v in chunks.items()})


# ------------------------------------------------------------------------------------------------ on-disk formats
def to_entries(log):
    """IterationLog -> list of dicts shaped like the reference's log entries (fp64 arrays; u_seq/k_seq/K_seq as lists of
    per-step arrays, alpha/new_* = None where no step was accepted)."""
    out = []
    for e in range(len(log)):
        found = bool(log.found_update[e])
        out.append({
            "iteration": int(log.iteration[e]),
            "x_seq": log.x_seq[e].astype(np.float64),
            "u_seq": [r for r in log.u_seq[e].astype(np.float64)],
            "current_cost": float(log.current_cost[e]),
            "k_seq": [r for r in log.k_seq[e].astype(np.float64)],
            "K_seq": [r for r in log.K_seq[e].astype(np.float64)],
            "alpha": float(log.alpha[e]) if found else None,
            "new_x_seq": log.new_x_seq[e].astype(np.float64) if found else None,
            "new_u_seq": [r for r in log.new_u_seq[e].astype(np.float64)] if found else None,
            "new_cost": float(log.new_cost[e]) if found else None,
            "found_update": found,
        })
    return out


def write_pickle_stream(path, log, append=True):
    """The reference's combined-log format: consecutive `pickle.dump(entry)` records, one dict per iteration
    (training_data_collection.py:265-290; read back by transformer_training.py:9-29)."""
    with open(path, "ab" if append else "wb") as fh:
        for entry in to_entries(log):
            pickle.dump(entry, fh)
    return len(log)


def write_npz(path, log):
    """Plain-array equivalent of the pickle stream (loads with numpy.load, nothing is unpickled)."""
    np.savez_compressed(path, **{f.name: getattr(log, f.name) for f in fields(IterationLog)})


def read_npz(path):
    with np.load(path) as z:
        return IterationLog(**{f.name: z[f.name] for f in fields(IterationLog)})


# ------------------------------------------------------------------------------------------------ logs -> training tensors
def create_dataset(x_seq, k_seq, K_seq, prompt_len):
    """TransformerILQR._create_dataset (transformer_ilqr.py:70-92): per entry kK[t] = concat(k[t][:, None], K[t], -1)
    flattened to m*(1+n); entries with no more than prompt_len steps are dropped.  float32 like the reference."""
    x = np.asarray(x_seq, dtype=np.float32)
    k = np.asarray(k_seq, dtype=np.float32)
    K = np.asarray(K_seq, dtype=np.float32)
    kK = np.concatenate([k[..., None], K], axis=-1).reshape(k.shape[0], k.shape[1], -1)
    if kK.shape[1] <= prompt_len:
        return x[:0], kK[:0]
    return x, kK


def fit_normalizer(x_data, kK_data, eps=1e-6):
    """DataNormalizer.fit (transformer_model.py:27-31): mean / (std + eps) over entries and time."""
    return dict(x_mean=x_data.mean(axis=(0, 1)), x_std=x_data.std(axis=(0, 1)) + eps,
                u_mean=kK_data.mean(axis=(0, 1)), u_std=kK_data.std(axis=(0, 1)) + eps)


def training_slices(x_data, kK_data, norm, prompt_len):
    """The tensors TransformerILQR.fit trains on (transformer_ilqr.py:108-114): normalised states, the LAST prompt_len
    rows of the normalised gains as prompt, the first T - prompt_len rows as target (T = x_data.shape[1])."""
    x_norm = (x_data - norm["x_mean"]) / norm["x_std"]
    kK_norm = (kK_data - norm["u_mean"]) / norm["u_std"]
    T = x_data.shape[1]
    return x_norm, kK_norm[:, -prompt_len:, :], kK_norm[:, :T - prompt_len, :]
