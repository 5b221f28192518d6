"""What a one-GPU box can check of the N > 1 path of bench.py before a multi-GPU node sees it: torch.distributed with
backend "nccl" (= RCCL) initialises in this image, all_reduce / barrier / batch_isend_irecv to self run on the device,
and dist.ShardedOperator's TorchComm + the sd_*_sharded entry points accept it (world_size 1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import datetime

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
t = torch.arange(8, dtype=torch.float64, device=dev)
dist.all_reduce(t)
dist.barrier()
a = torch.arange(1024, dtype=torch.float64, device=dev)
b = torch.zeros_like(a)
reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, b, 0), dist.P2POp(dist.isend, a, 0)])
for r in reqs:
    r.wait()
torch.cuda.synchronize()
ok = bool(torch.equal(a, b)) and bool(torch.equal(t, torch.arange(8, dtype=torch.float64, device=dev)))
print("nccl one-rank smoke:", "ok" if ok else "FAILED", "| backend", dist.get_backend(), flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
