"""world_size-2 checks of the multi-GPU path on CPU (gloo): shard bounds and the all-gather of the gain stacks.
The collective code is device-agnostic (RCCL when the tensors live on GPUs)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import PKG_DIR, ROOT

torch = pytest.importorskip("torch")


def test_shard_bounds_cover_the_batch_exactly_once():
    from quattro_ilqr_amd.parallel import shard_bounds
    for total in (0, 1, 7, 8, 4096, 32768, 32771):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(32768, 3, 8) == (12288, 16384)          # BASELINE configs[3]: 4096 per GPU
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_pack_unpack_gains_roundtrip():
    from quattro_ilqr_amd.parallel import pack_gains, unpack_gains
    K = torch.randn(5, 7, 4, 12)
    k = torch.randn(5, 7, 4)
    buf = pack_gains(K, k)
    assert buf.shape == (5, 7, 4, 13) and buf.is_contiguous()
    K2, k2 = unpack_gains(buf)
    assert torch.equal(K, K2) and torch.equal(k, k2)


WORKER = textwrap.dedent("""
    import os, sys
    sys.path[:0] = [{root!r}, {pkg!r}]
    import torch, torch.distributed as dist
    from quattro_ilqr_amd.parallel import all_gather_gains, shard_bounds
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total, N, m, n = 9, 5, 4, 12                      # 9 trajectories over 2 ranks: shards of 5 and 4 (uneven)
    g = torch.Generator().manual_seed(0)
    K_all = torch.randn(total, N, m, n, generator=g)
    k_all = torch.randn(total, N, m, generator=g)
    lo, hi = shard_bounds(total, rank, world)
    K, k = all_gather_gains(K_all[lo:hi].contiguous(), k_all[lo:hi].contiguous())
    assert torch.equal(K, K_all) and torch.equal(k, k_all), "gathered gains differ from the global stack"
    # equal shards (the benchmark's case: same batch on every rank): one collective, no size exchange, views into a
    # reusable receive buffer
    Ke, ke = K_all[:8], k_all[:8]
    buf = None
    for _ in range(2):
        K2, k2 = all_gather_gains(Ke[4 * rank: 4 * rank + 4].contiguous(), ke[4 * rank: 4 * rank + 4].contiguous(),
                                  equal_shards=True, out=buf)
        assert torch.equal(K2, Ke) and torch.equal(k2, ke), "equal-shard gather differs"
        assert buf is None or K2._base is buf
        buf = K2._base
    # the benchmark's exchange: K and k are views of ONE flat buffer per rank, ONE collective, results are views of the
    # preallocated receive buffer (no packing / unpacking copy anywhere)
    from quattro_ilqr_amd.parallel import GainGather
    B = 4
    flat = torch.empty(B * N * m * (n + 1))
    Kv, kv = flat[:B * N * m * n].view(B, N, m, n), flat[B * N * m * n:].view(B, N, m)
    Kv.copy_(Ke[B * rank: B * rank + B]); kv.copy_(ke[B * rank: B * rank + B])
    gg = GainGather(B, N, m, n, torch.float32, "cpu")
    assert gg.bytes_received_per_rank == (world - 1) * flat.numel() * 4
    for _ in range(2):
        Kg, kg = gg(flat)
        assert Kg.shape == (world, B, N, m, n) and kg.shape == (world, B, N, m)
        assert Kg.untyped_storage().data_ptr() == gg.recv.untyped_storage().data_ptr()
        assert torch.equal(Kg.reshape(world * B, N, m, n), Ke) and torch.equal(kg.reshape(world * B, N, m), ke)
    Kg, kg, work = gg(flat, async_op=True)
    work.wait()
    assert torch.equal(Kg.reshape(world * B, N, m, n), Ke)
    try:
        gg(flat[:-1])
        raise SystemExit("a short buffer must be refused")
    except ValueError:
        pass
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_all_gather_gains_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, pkg=PKG_DIR))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {rank} ok" in out


def _parse_one_json_line(stdout):
    import json
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_self_launches_its_ranks_dry_rehearsal():
    """`python bench.py --gpus 2` with no WORLD_SIZE must start its own ranks.  QT_BENCH_REHEARSAL=dry runs
    the whole N > 1 path except the kernels (there is no GPU here): launcher, rendezvous, the [K | k] gather through
    parallel.GainGather, max-over-ranks timing and exactly ONE JSON line from rank 0 on stdout."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(QT_BENCH_REHEARSAL="dry", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batch", "64"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = _parse_one_json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 3 and out["gather_ok"] is True
    assert out["launched_by"] == "bench.py" and out["config"]["global_batch"] == 128
    assert out["gather_ms"] > 0 and out["scaling"] == "weak"


def test_bench_world_size_8_dry_rehearsal_is_run_ready():
    """BASELINE configs[3] (8 x 4096 trajectories) has never met an 8-GPU node: the whole non-kernel path at WORLD SIZE 8 —
    launcher, rendezvous, barriers, the one all-gather of the [K | k] buffers at the configured shape, max-over-ranks timing —
    must produce exactly one JSON line that interprets itself (per-rank arrays of length 8, the expected gather band)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(QT_BENCH_REHEARSAL="dry", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = _parse_one_json_line(r.stdout)
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["gather_ok"] is True
    assert out["config"]["global_batch"] == 32768 and out["config"]["batch_per_gpu"] == 4096
    assert len(out["ms_per_step_per_rank"]) == 8 and len(out["comm"]["gather_ms_per_rank"]) == 8
    assert out["comm"]["gather_bytes_received_per_rank"] == 7 * 4096 * 50 * 4 * 13 * 4          # 7 peers x 42.6 MB
    exp = out["comm"]["expected_ms"]
    assert 0.2 < exp["low"] < 0.35 and 0.9 < exp["high"] < 1.1 and exp["bytes_received_per_rank"] == 7 * 4096 * 50 * 4 * 13 * 4
    assert out["scaling"] == "weak" and out["metric"].startswith("iLQR iterations/sec")


def test_bench_under_torch_distributed_run_dry_rehearsal():
    """The driver's documented N > 1 command line (python -m torch.distributed.run ... bench.py --gpus N) takes the same
    path without the self-launch."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(QT_BENCH_REHEARSAL="dry", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--steps", "2", "--warmup", "1", "--batch", "32"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = _parse_one_json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["launched_by"] == "external launcher" and out["gather_ok"] is True


def test_bench_launcher_reports_a_failing_rank():
    """A rank that dies must make the launcher exit non-zero (and not hang on the survivors)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(QT_BENCH_REHEARSAL="dry", OMP_NUM_THREADS="1", QT_BENCH_TEST_FAIL_RANK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--batch", "8"],
                       env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "0"], env=env,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0
