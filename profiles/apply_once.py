"""A few applies of one configuration, for counter passes: python profiles/apply_once.py L dtype(f64|c128) [steps] [open|periodic]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package()
L = int(sys.argv[1])
dt = torch.float64 if sys.argv[2] == "f64" else torch.complex128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
bc = sys.argv[4] if len(sys.argv) > 4 else "open"
m = pkg.XXZChain(L, nup=L // 2, boundary=bc)
a = torch.ones(m.N, dtype=dt, device="cuda")
b = torch.empty_like(a)
for _ in range(steps):
    pkg.apply_H(b, a, m)
torch.cuda.synchronize()
print("N", m.N)
