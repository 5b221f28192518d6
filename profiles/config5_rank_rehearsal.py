"""BASELINE config 5 (XXZChain L=36 nup=18, KPM S(q,w) on 8 GPUs) on ONE GPU: one rank's share of the sharded moment recursion at
full size.  The rank's model, vectors, halo and send buffers are the real ones (sd_kpm_moments_sharded behind a callback
communicator); the communicator moves nothing (the halo keeps what is in it, all-reduce is the identity), so the moments are
meaningless -- what is measured is that the recursion code path runs at this size on a rank, its time per step and its device memory.
usage: python profiles/config5_rank_rehearsal.py [rank=3] [M=64] [L=36] [world=8]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
from spindynamics_jl_amd import _lib

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L = int(sys.argv[3]) if len(sys.argv) > 3 else 36
world = int(sys.argv[4]) if len(sys.argv) > 4 else 8

t0 = time.time()
model = pkg.XXZChain(L, nup=L // 2)
op = pkg.ShardedOperator(model, rank, world, exchange_fn=lambda o, p, h: None)
plan_s = time.time() - t0
dev = torch.device("cuda")
free0, total = torch.cuda.mem_get_info()
phi = op.empty(torch.complex128, dev)
op.fill_randn(phi, 7)
nrm = float(torch.linalg.vector_norm(phi))
phi /= nrm

calls = {"start": 0, "wait": 0, "reduce": 0}


def ex_start(_u, _dtype, _src, _halo):
    calls["start"] += 1
    return 0


def ex_wait(_u):
    calls["wait"] += 1
    return 0


def allreduce(_u, _vals, _count):
    calls["reduce"] += 1
    return 0


cbs = _lib.sd_comm_callbacks(None, _lib.EXCHANGE_START_FN(ex_start), _lib.EXCHANGE_WAIT_FN(ex_wait), _lib.ALLREDUCE_FN(allreduce))
h = C.c_void_p()
pkg.check(pkg.lib().sd_comm_from_callbacks(C.byref(cbs), rank, world, C.byref(h)))
mu = np.zeros(M)
model.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
a, b = L / 2 + 1.0, 0.0
res = {"config": 5, "what": "one rank's share of sd_kpm_moments_sharded at full size, communicator moves nothing", "L": L, "world": world,
       "rank": rank, "rows_owned": op.n_local, "rows_imported": op.n_halo, "rows_packed": op.n_send, "plan_seconds": plan_s}
for label, m_ in (("warm", 8), ("timed", M)):
    torch.cuda.synchronize()
    t0 = time.time()
    pkg.check(pkg.lib().sd_kpm_moments_sharded(model.ctx.h, model.h, h, phi.data_ptr(), op.n_local, m_, a, b,
                                               mu.ctypes.data_as(C.POINTER(C.c_double))), model.ctx.h)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if label == "timed":
        free1, _ = torch.cuda.mem_get_info()
        res.update({"moments": m_, "applies": m_ // 2, "seconds": dt, "ms_per_apply_step": dt / (m_ // 2) * 1e3,
                    "device_GB_in_use_at_end": (total - free1) / 1e9, "device_GB_total": total / 1e9,
                    "callback_calls": dict(calls), "mu_finite": bool(np.isfinite(mu).all())})
pkg.lib().sd_comm_destroy(h)
print(json.dumps(res), flush=True)
