"""Can two ranks share ONE GPU under RCCL (only to exercise the nccl code path on a 1-GPU box)?"""
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
r, w = dist.get_rank(), dist.get_world_size()
a = torch.full((w * 4,), float(r), device="cuda")
b = torch.empty_like(a)
dist.all_to_all_single(b, a)
t = torch.tensor([1.0], device="cuda", dtype=torch.float64)
dist.all_reduce(t)
print("rank", r, "alltoall", b.tolist(), "allreduce", t.item(), flush=True)
dist.destroy_process_group()
