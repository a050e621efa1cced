"""On the GPU box: the collective calls of bench.py's multi-rank path over RCCL itself (backend "nccl"), as far as one GPU
allows -- a ONE-rank group (RCCL refuses two ranks on one device): init with device_id, the int64 count all-reduce, the
float64 MAX all-reduce, barrier, destroy.  usage: python tools/rccl_probe.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
dist.init_process_group(backend="nccl", device_id=dev)
c = torch.arange(11, dtype=torch.int64, device=dev)
dist.all_reduce(c)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, dist.ReduceOp.MAX)
torch.cuda.synchronize()
dist.barrier()
torch.cuda.synchronize()
assert c.cpu().tolist() == list(range(11)) and float(t.item()) == 1.25
dist.barrier()
dist.destroy_process_group()
print("rccl one-rank group ok: init + all_reduce(int64 SUM) + all_reduce(float64 MAX) + barrier in %.2f s" % (time.perf_counter() - t0))
