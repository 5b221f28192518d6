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

for L in (16, 20, 24):
    m = pkg.XXZChain(L, nup=L // 2)
    v = np.random.default_rng(L).standard_normal(m.N) + 0j
    v /= np.linalg.norm(v)
    def median_ms(fn, reps=31):          # medians: single calls show 50-80 ms outliers (host jitter) in every build
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.time()
            r = fn()
            ts.append(time.time() - t0)
        return sorted(ts)[reps // 2], r
    dt, out = median_ms(lambda: pkg.krylov_time_evolve(v, 0.1, pkg.apply_H, m, kry_m=30))
    dt2, (lo, hi) = median_ms(lambda: pkg.lanczos_extremal(pkg.apply_H, m, lanc_m=80, seed=1))
    print(json.dumps({"L": L, "N": m.N, "krylov_evolve_kry_m30_ms": dt * 1e3, "lanczos_extremal_80_ms": dt2 * 1e3,
                      "Emin": lo, "Emax": hi, "norm": float(np.linalg.norm(out))}), flush=True)
