"""The torch.distributed calls SlabForce / SlabDecompose / bench.py make, issued on backend "nccl" (RCCL)
with world_size 1 -- all a 1-GPU box allows (RCCL refuses two ranks on one device, see
nccl_same_gpu_probe.py).  It checks the API shapes (float64 slices, async all-to-all + wait on the
current stream, coalesced isend/irecv, uneven all-to-all-v with zero-length splits, barrier, destroy),
not the multi-GPU data path, which the gloo world_size-2 tests and the virtual-rank GPU tests cover.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 tools/nccl_api_probe.py
"""
import os
import torch
import torch.distributed as dist

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
g = dist.group.WORLD
P, r = dist.get_world_size(), dist.get_rank()

s = torch.full((1,), 3.0, dtype=torch.float64, device=dev)
dist.all_reduce(s, op=dist.ReduceOp.SUM, group=g)
assert s.item() == 3.0 * P

n = 1 << 24
a = torch.arange(n + 5, dtype=torch.float64, device=dev)
b = torch.zeros(n + 5, dtype=torch.float64, device=dev)
dist.all_to_all_single(b[:n], a[:n], group=g)
assert torch.equal(b[:n], a[:n]) and b[n:].sum().item() == 0
c = torch.zeros_like(b)
w = dist.all_to_all_single(c[:n], (a * 2)[:n], group=g, async_op=True)
w.wait()
assert torch.equal(c[:n], 2 * a[:n])

send = [torch.full((1000,), float(i + 1), dtype=torch.float64, device=dev) for i in range(3)]
recv = [torch.zeros(1000, dtype=torch.float64, device=dev) for i in range(3)]
ops = []
for x, y in zip(send, recv):
    ops.append(dist.P2POp(dist.isend, x, (r + 1) % P, group=g))
    ops.append(dist.P2POp(dist.irecv, y, (r - 1) % P, group=g))
for w in dist.batch_isend_irecv(ops):
    w.wait()
assert all(torch.equal(x, y) for x, y in zip(send, recv))

# the per-range exchange of the pipelined transposes: slices of big buffers, asynchronous, several batches queued
big_s = torch.arange(1 << 20, dtype=torch.float64, device=dev)
big_r = torch.zeros(1 << 20, dtype=torch.float64, device=dev)
pending = []
for c in range(4):
    ops = []
    for peer in range(P):
        s_, r_ = big_s[c * 1000 + peer * 5000: c * 1000 + peer * 5000 + 1000], big_r[c * 1000 + peer * 5000: c * 1000 + peer * 5000 + 1000]
        if peer == r:
            r_.copy_(s_)
        else:
            ops += [dist.P2POp(dist.isend, s_, peer, group=g), dist.P2POp(dist.irecv, r_, peer, group=g)]
    pending.append(dist.batch_isend_irecv(ops) if ops else [])
for ws in pending:
    for w in ws:
        w.wait()
assert torch.equal(big_r[:4000], big_s[:4000])

cnt = torch.tensor([7], dtype=torch.int64, device=dev)
got = torch.zeros_like(cnt)
dist.all_to_all_single(got, cnt, group=g)
rows = torch.rand(7, 3, dtype=torch.float64, device=dev)
out = torch.zeros(7, 3, dtype=torch.float64, device=dev)
dist.all_to_all_single(out, rows, output_split_sizes=[7], input_split_sizes=[7], group=g)
assert torch.equal(out, rows)
e = torch.zeros(0, 3, dtype=torch.float32, device=dev)
dist.all_to_all_single(torch.zeros(0, 3, dtype=torch.float32, device=dev), e, output_split_sizes=[0], input_split_sizes=[0], group=g)

t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("nccl api probe ok")
