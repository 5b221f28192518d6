"""A/B helper: apply time of open vs periodic chains (python profiles/periodic_ab.py [L]); SD_LIB_PATH selects the build."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 28
for bc in ("open", "periodic"):
    m = pkg.XXZChain(L, nup=L // 2, boundary=bc)
    a = torch.ones(m.N, dtype=torch.complex128, device="cuda")
    b = torch.empty_like(a)
    for _ in range(3):
        pkg.apply_H(b, a, m)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        pkg.apply_H(b, a, m)
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"lib": os.path.basename(os.environ.get("SD_LIB_PATH", "new")), "bc": bc, "L": L,
                      "ms": e0.elapsed_time(e1) / 30}), flush=True)
