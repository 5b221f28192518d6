"""H|psi> for models beyond the nearest-neighbour chain (build_model with longer bonds, src/SpinModel.jl:23-46): J1-J2 chain, third neighbours,
all pairs (long_range_hopping).  ms per apply and rows/s, ComplexF64:  python profiles/general_bonds_bench.py [L=28]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g

pkg = g.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 28
nup = L // 2


def chain(ranges, periodic=False):
    hop, zz = [], []
    for d, J in ranges:
        for i in range(1, L + 1):
            j = i + d
            if j > L:
                if not periodic:
                    continue
                j -= L
            hop.append((i, j, 0.5 * J)); zz.append((i, j, J))
    return hop, zz


cases = [("nearest neighbours (XXZChain)", chain([(1, 1.0)])),
         ("J1-J2 (J2 = 0.5)", chain([(1, 1.0), (2, 0.5)])),
         ("J1-J2-J3", chain([(1, 1.0), (2, 0.5), (3, 0.25)])),
         ("J1-J2 periodic", chain([(1, 1.0), (2, 0.5)], True)),
         ("all pairs, J = 1/r^2", ([(i, j, 0.5 / (j - i) ** 2) for i in range(1, L + 1) for j in range(i + 1, L + 1)],
                                   [(i, j, 1.0 / (j - i) ** 2) for i in range(1, L + 1) for j in range(i + 1, L + 1)]))]
for name, (hop, zz) in cases:
    if "all pairs" in name and L > 26:
        continue
    m = pkg.build_model(L, nup=nup, hopping=hop, zz=zz)
    a = torch.randn(m.N, dtype=torch.complex128, device="cuda")
    b = torch.empty_like(a)
    for _ in range(2):
        pkg.apply_H(b, a, m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        pkg.apply_H(b, a, m)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"case": name, "L": L, "N": m.N, "bonds": len(hop), "device_path": m.device_path, "ms": round(ms, 4),
                      "Grows_per_s": round(m.N / ms / 1e6, 2), "ms_per_bond_per_Grow": round(ms / len(hop) / (m.N / 1e9), 3)}), flush=True)
    del a, b, m
