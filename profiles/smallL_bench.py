"""Per-step wall time of the un-reorthogonalised Lanczos recursion (lanczos_tridiag, src/Lanczos.jl:196-246) on small and
medium systems, where launches and host round trips, not bandwidth, set the pace: python profiles/smallL_bench.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
for L in (12, 16, 20, 24, 26):
    m = pkg.XXZChain(L, nup=L // 2)
    v = np.random.default_rng(L).standard_normal(m.N) + 0j
    steps = min(200, m.N - 1)
    pkg.lanczos_tridiag(pkg.apply_H, m, v, lanc_m=5)
    t0 = time.time()
    out = pkg.lanczos_tridiag(pkg.apply_H, m, v, lanc_m=steps)
    dt = time.time() - t0
    print(json.dumps({"what": "lanczos_tridiag", "L": L, "N": m.N, "steps": steps, "us_per_step": dt / steps * 1e6,
                      "alpha0": float(out[0][0])}), flush=True)
