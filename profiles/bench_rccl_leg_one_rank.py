"""The library's own RCCL communicator (what `bench.py --gpus N` times as its headline through ShardedOperator.comm() / apply_lib),
exercised with ONE rank on this GPU: a 1-rank NCCL process group, RcclComm from a broadcast id, ring self-test, sd_apply_sharded on
that communicator against the Python-issued step, a timed loop.  What needs a peer cannot run here; every call of the path does.
usage: python profiles/bench_rccl_leg_one_rank.py [L=20]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import __graft_entry__ as g

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29671")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pkg = g.load_package()
from spindynamics_jl_amd.dist import RcclComm
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model = pkg.XXZChain(L, nup=L // 2)
op = pkg.ShardedOperator(model, 0, 1)
a = op.fill_randn(op.empty(torch.complex128, dev), 1)
b = torch.empty_like(a)
ref = torch.empty_like(a)
comm = RcclComm(op, dev)
lib, m = pkg.lib(), op.model
m.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
pkg.check(lib.sd_comm_selftest(m.ctx.h, comm.h), m.ctx.h)
op.apply(ref, a)
pkg.check(lib.sd_apply_sharded(m.ctx.h, m.h, comm.h, 2, b.data_ptr(), a.data_ptr(), op.n_local, 1), m.ctx.h)
torch.cuda.synchronize()
same = bool(torch.equal(ref, b))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
x, y = a, b
for _ in range(5):
    pkg.check(lib.sd_apply_sharded(m.ctx.h, m.h, comm.h, 2, y.data_ptr(), x.data_ptr(), op.n_local, 1), m.ctx.h)
    x, y = y, x
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("rccl_one_rank_ms:", ms, "bit-identical:", same, flush=True)
comm.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if same and ms > 0 else 1)
