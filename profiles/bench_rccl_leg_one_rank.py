"""bench.py's leg through the library's own RCCL communicator (c_rccl_path), exercised with ONE rank on this GPU: a 1-rank NCCL
process group, RcclComm, ring self-test, sd_apply_sharded against the torch path, timing.  What needs a peer cannot run here;
every line of the leg's Python does.  usage: python profiles/bench_rccl_leg_one_rank.py [L=20]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import __graft_entry__ as g
import bench

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29671")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pkg = g.load_package()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model = pkg.XXZChain(L, nup=L // 2)
op = pkg.ShardedOperator(model, 0, 1)
a = op.fill_randn(op.empty(torch.complex128, dev), 1)
b = torch.empty_like(a)


def all_ok(flag):
    t = torch.tensor([1.0 if flag else 0.0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() == 1.0)


res = bench.c_rccl_path(pkg, op, a, b, 5, dist, "nccl", dev, 0, all_ok, lambda: None)
print("c_rccl_path_ms:", res, flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if isinstance(res, float) and res > 0 else 1)
