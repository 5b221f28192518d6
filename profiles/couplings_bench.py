"""Apply time for couplings that defeat the exact shortcuts (closed-form diagonal, FMA): python profiles/couplings_bench.py [L]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for name, kw in (("Jxy=1 Jz=1 hz=0 (closed-form diagonal, fma)", {}),
                 ("Jz=0.7 (list-order diagonal)", {"Jz": 0.7}),
                 ("Jz=0.7 hz=0.3 (list-order diagonal with fields)", {"Jz": 0.7, "hz": 0.3}),
                 ("Jxy=0.9 Jz=0.7 hz=0.3 (no fma either)", {"Jxy": 0.9, "Jz": 0.7, "hz": 0.3})):
    m = pkg.XXZChain(L, nup=L // 2, **kw)
    a = torch.ones(m.N, dtype=torch.complex128, device="cuda")
    b = torch.empty_like(a)
    for _ in range(3):
        pkg.apply_H(b, a, m)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        pkg.apply_H(b, a, m)
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"case": name, "L": L, "ms": e0.elapsed_time(e1) / 20}), flush=True)
    del a, b, m
