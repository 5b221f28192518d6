"""Apply time of the variants in DESIGN.md §5's table (periodic chain, full 2^L basis, f64): python profiles/variants_bench.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package()


def timed(model, dtype, steps=20):
    a = torch.ones(model.N, dtype=dtype, device="cuda")
    b = torch.empty_like(a)
    for _ in range(3):
        pkg.apply_H(b, a, model)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        pkg.apply_H(b, a, model)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for name, model, dt in (
        ("open L=28 c128", pkg.XXZChain(28, nup=14), torch.complex128),
        ("periodic L=28 c128", pkg.XXZChain(28, nup=14, boundary="periodic"), torch.complex128),
        ("full basis L=24 c128", pkg.XXZChain(24), torch.complex128),
        ("open L=30 f64", pkg.XXZChain(30, nup=15), torch.float64),
        ("open L=30 c128", pkg.XXZChain(30, nup=15), torch.complex128)):
    ms = timed(model, dt)
    print(json.dumps({"case": name, "N": model.N, "ms": ms, "Grows_per_s": model.N / ms / 1e6, "path": model.device_path}), flush=True)
