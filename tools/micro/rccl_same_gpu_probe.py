#!/usr/bin/env python3
"""Can two ranks share ONE GPU under RCCL on this box?  (Would let the partitioned solver's RCCL transport run for real, with N = 2, on the
one-GPU development boxes.)  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/micro/rccl_same_gpu_probe.py"""
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.full((4,), float(rank + 1), device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
print(f"rank {rank}: all_reduce on a shared GPU -> {t.tolist()}", flush=True)
dist.barrier()
dist.destroy_process_group()
