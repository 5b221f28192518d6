"""A few applies of one configuration, for counter passes: python profiles/apply_once.py L dtype(f64|c128) [steps] [open|periodic|j1j2]
(j1j2: build_model with second-neighbour bonds of half the strength behind the chain bonds: the general-bond plan)"""
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
if bc == "j1j2":
    hop = [(i, i + d, 0.5 * J) for d, J in ((1, 1.0), (2, 0.5)) for i in range(1, L - d + 1)]
    zz = [(i, i + d, J) for d, J in ((1, 1.0), (2, 0.5)) for i in range(1, L - d + 1)]
    m = pkg.build_model(L, nup=L // 2, hopping=hop, zz=zz)
else:
    m = pkg.XXZChain(L, nup=L // 2, boundary=bc)
a = torch.ones(m.N, dtype=dt, device="cuda")
b = torch.empty_like(a)
for _ in range(steps):
    pkg.apply_H(b, a, m)
torch.cuda.synchronize()
print("N", m.N)
