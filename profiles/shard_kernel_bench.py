"""Per-rank kernel time of the sharded apply at sizes one GPU cannot hold whole (no peers needed: the halo buffer is filled
locally; the exchange itself cannot be measured on a 1-GPU box).  Usage: python profiles/shard_kernel_bench.py L world rank..."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package()
L, world = int(sys.argv[1]), int(sys.argv[2])
for rank in [int(r) for r in sys.argv[3:]]:
    model = pkg.XXZChain(L, nup=L // 2)
    op = pkg.ShardedOperator(model, rank, world, exchange_fn=lambda o, p, h: None)
    a = op.empty(torch.complex128, "cuda")
    b = op.empty(torch.complex128, "cuda")
    op.fill_randn(a, 1)
    halo = op.halo(a)
    halo.fill_(0.5)
    info = model.shard_info()
    res = {"L": L, "world": world, "rank": rank, "n_local": op.n_local, "n_halo": op.n_halo, "N": model.N,
           "n_interior_rows": int(info.n_interior_rows), "n_boundary_rows": op.n_local - int(info.n_interior_rows),
           "n_interior_tiles": int(info.n_interior_tiles), "mode": op.mode}
    for name, part in (("all", 0), ("interior", 1), ("boundary", 2)):
        for _ in range(2):
            op._launch(b, a, halo, 0, part=part)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            op._launch(b, a, halo, 0, part=part)
        e1.record()
        torch.cuda.synchronize()
        res[name + "_ms"] = e0.elapsed_time(e1) / 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if op.packed:
        op.pack(a)
        e0.record()
        for _ in range(5):
            op.pack(a)
        e1.record()
        torch.cuda.synchronize()
        res["pack_ms"] = e0.elapsed_time(e1) / 5
        res["n_send"] = op.n_send
        res["packed"] = int(info.packed)
    res["local_total_ms"] = res["interior_ms"] + res["boundary_ms"] + res.get("pack_ms", 0.0)
    res["ideal_ms_at_single_gpu_rate"] = None
    print(json.dumps(res), flush=True)
    del a, b, halo, op, model
    torch.cuda.empty_cache()
