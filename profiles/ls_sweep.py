"""Apply time vs suffix length (SD_SUFFIX_BITS) per dtype: python profiles/ls_sweep.py [L]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def timed(model, dtype, steps=30):
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


for ls in ("12", "13", "14", "12", "13"):
    os.environ["SD_SUFFIX_BITS"] = ls          # read when the model's plan is built
    m = pkg.XXZChain(L, nup=L // 2)
    print(json.dumps({"L": L, "LS": int(ls), "f64_ms": timed(m, torch.float64), "c128_ms": timed(m, torch.complex128)}), flush=True)
    del m
