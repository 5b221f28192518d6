"""Streaming kernels next to the apply (Sz_q, observables, pack, fill, BLAS-1) at L=32: ms and GB/s of algorithmic bytes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package()
L = int(os.environ.get("SD_AUX_L", "32"))
FULL = os.environ.get("SD_AUX_FULL") == "1"          # full 2^L basis (nup = nothing): state = row, one row per thread
m = pkg.XXZChain(L) if FULL else pkg.XXZChain(L, nup=L // 2)
op = pkg.ShardedOperator(m, 0, 1)
psi = op.empty(torch.complex128, "cuda")
op.fill_randn(psi, 3)
psi /= op.norm(psi)


def ev(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def line(what, ms, bpr):
    print(json.dumps({"what": what, "L": L, "basis": "full" if FULL else "sector", "N": m.N, "ms": ms, "alg_B_per_row": bpr, "GBs": bpr * m.N / ms / 1e6}), flush=True)


line("Sz_q_vector (c128 in, c128 out)", ev(lambda: pkg.Sz_q_vector(m, psi, 0.7)), 32)
line("magnetization_per_site", ev(lambda: pkg.magnetization_per_site(psi, m)), 16)
line("connected_correlations", ev(lambda: pkg.connected_correlations(psi, m)), 16)
line("fill_randn (tiles)", ev(lambda: op.fill_randn(psi, 5)), 16)
line("norm", ev(lambda: op.norm(psi)), 16)
line("dot", ev(lambda: op.dot(psi, psi)), 32)
