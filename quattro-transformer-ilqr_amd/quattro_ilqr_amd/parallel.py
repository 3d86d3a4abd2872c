"""Multi-GPU: independent trajectories shard across ranks, one process per GPU.

The reference has no distributed code at all (one trajectory per iLQR_TF instance); trajectories are
independent optimisation problems, so the data path needs NO collective.  The only exchange is the one the north star
names: an all-gather of the resulting gain stacks (K, k) so every rank holds the gains of the whole batch
(RCCL over xGMI when the tensors are on GPUs; the same code runs over gloo on CPU tensors in the tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous shard [lo, hi) of `total` trajectories owned by `rank`; sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank outside [0, world)")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_gains(K, k):
    """(B,N,m,n), (B,N,m) -> one contiguous (B,N,m,1+n) buffer [k | K] per control row: one collective instead of two."""
    return torch.cat([k.unsqueeze(-1), K], dim=-1).contiguous()


def unpack_gains(buf):
    return buf[..., 1:].contiguous(), buf[..., 0].contiguous()


def all_gather_gains(K, k, group=None, equal_shards=False, out=None):
    """Every rank contributes its shard's gains; returns (K_all, k_all) ordered by rank (= batch order of shard_bounds).
    Shards may differ in size by one trajectory: shorter shards are padded to the longest for the collective.
    equal_shards=True (every rank holds the same number of trajectories, e.g. 4096 per GPU) skips the size exchange and
    its host synchronisations: the whole gather is then ONE collective on one packed buffer.  `out` may hold a
    (world*B, N, m, 1+n) receive buffer to reuse; with equal shards the results are returned as views into it (no
    unpacking copy of the gathered 340 MB)."""
    world = dist.get_world_size(group)
    if world == 1:
        return K, k
    mine = pack_gains(K, k)
    if equal_shards:
        shape = (world * mine.shape[0],) + tuple(mine.shape[1:])
        if out is None or tuple(out.shape) != shape or out.dtype != mine.dtype or out.device != mine.device:
            out = torch.empty(shape, dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(out, mine, group=group)
        return out[..., 1:], out[..., 0]
    sizes = [torch.zeros(1, dtype=torch.int64, device=mine.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([mine.shape[0]], dtype=torch.int64, device=mine.device), group=group)
    sizes = [int(s.item()) for s in sizes]
    bmax = max(sizes)
    if mine.shape[0] < bmax:
        pad = torch.zeros((bmax - mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        mine = torch.cat([mine, pad], dim=0)
    out = torch.empty((world * bmax,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    parts = [out[r * bmax: r * bmax + sizes[r]] for r in range(world)]
    return unpack_gains(torch.cat(parts, dim=0))


class GainGather:
    """The benchmark's / a sharded solve's exchange with equal shards, without any repacking copy: every rank's gains
    live in ONE flat buffer `[K (B*N*m*n) | k (B*N*m)]` (QuattroILQR allocates K and k as views of such a buffer:
    `solver.gains_flat`), so the whole gather is one `all_gather_into_tensor` of that buffer, straight out of the memory
    the sweep wrote, into a preallocated `(world, flat)` receive buffer.  The results are views of the receive buffer:
    K_all (world, B, N, m, n), k_all (world, B, N, m) — rank r's shard at index r (= batch order of shard_bounds).

    Replaces nothing in the reference (which has no distributed code); it is the north star's "RCCL gather
    over xGMI of the resulting K/k gain stacks"."""

    def __init__(self, B, N, m, n, dtype, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.shape = (B, N, m, n)
        self.nK, self.nk = B * N * m * n, B * N * m
        self.recv = torch.empty((self.world, self.nK + self.nk), dtype=dtype, device=device)

    @property
    def bytes_received_per_rank(self):
        """Bytes this rank receives from its peers in one gather."""
        return (self.world - 1) * (self.nK + self.nk) * self.recv.element_size()

    def views(self):
        B, N, m, n = self.shape
        K_all = self.recv[:, :self.nK].view(self.world, B, N, m, n)
        k_all = self.recv[:, self.nK:].view(self.world, B, N, m)
        return K_all, k_all

    def __call__(self, gains_flat, async_op=False):
        """gains_flat: this rank's contiguous `[K | k]` buffer (nK + nk elements).  -> (K_all, k_all[, work])"""
        if gains_flat.numel() != self.nK + self.nk or not gains_flat.is_contiguous():
            raise ValueError(f"gains_flat must be a contiguous buffer of {self.nK + self.nk} elements")
        work = None
        if self.world == 1:
            self.recv[0].copy_(gains_flat.reshape(-1))
        else:
            work = dist.all_gather_into_tensor(self.recv.view(-1), gains_flat.reshape(-1), group=self.group,
                                               async_op=async_op)
        K_all, k_all = self.views()
        return (K_all, k_all, work) if async_op else (K_all, k_all)


class ShardedILQR:
    """Runs a QuattroILQR on this rank's contiguous shard of a global batch and gathers the gains."""

    def __init__(self, solver, group=None):
        self.solver, self.group = solver, group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def solve(self, x0_global, u_init_global=None, gather=True, **kw):
        lo, hi = shard_bounds(x0_global.shape[0], self.rank, self.world)
        out = self.solver.solve(x0_global[lo:hi], None if u_init_global is None else u_init_global[lo:hi], **kw)
        out["shard"] = (lo, hi)
        if gather and self.world > 1:
            out["K_all"], out["k_all"] = all_gather_gains(out["K"], out["k"], self.group)
        return out
