import os, sys, torch, torch.distributed as dist
sys.path[:0] = ["/root/repo", "/root/repo/quattro-transformer-ilqr_amd"]
from quattro_ilqr_amd import parallel
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"]); lr = int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(lr); dev = torch.device("cuda", lr)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
K = torch.randn(5, 7, 4, 12, device=dev); k = torch.randn(5, 7, 4, device=dev)
buf = parallel.pack_gains(K, k); out = torch.empty_like(buf)
dist.all_gather_into_tensor(out, buf); dist.barrier()
t = torch.tensor([1.0], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert torch.equal(out, buf); print("rccl world", world, "ok")
dist.destroy_process_group()
