#!/usr/bin/env python3
"""One-rank RCCL probe of the collective calls bench.py makes for N > 1 (shapes, async handles, stream semantics):
    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 scripts/rccl_probe.py"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
world, rank = dist.get_world_size(), dist.get_rank()
res = [torch.randn((5, 3100, 1), dtype=torch.float64, device="cuda") for _ in range(2)]
gathered = [torch.empty((world, 5, 3100, 1), dtype=torch.float64, device="cuda") for _ in range(2)]
pending = [None, None]
for i in range(6):
    b = i % 2
    if pending[b] is not None:
        pending[b].wait()
    res[b].add_(1.0)
    pending[b] = dist.all_gather_into_tensor(gathered[b], res[b], async_op=True)
for w in pending:
    w.wait()
torch.cuda.synchronize()
dist.barrier()
assert torch.equal(gathered[0][rank], res[0]) and torch.equal(gathered[1][rank], res[1])
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
lst = [torch.empty_like(res[0]) for _ in range(world)]
dist.all_gather(lst, res[0])
assert torch.equal(lst[rank], res[0])
print("rccl probe ok: backend", dist.get_backend(), "world", world)
dist.destroy_process_group()
