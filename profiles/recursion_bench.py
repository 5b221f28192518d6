#!/usr/bin/env python3
"""Timings of the fused recursion steps on one MI355X (BASELINE.json configs 2-4), printed as JSON lines.
  config 4: XXZChain L=32 nup=16, Chebyshev term  (fused apply + rescale + recurrence + accumulate, 80 B/row algorithmic)
  config 3: XXZChain L=30 nup=15, KPM moment step (fused apply + rescale + recurrence + 2 reductions, 64 B/row)
  config 2: XXZChain L=28 nup=14, time_evolve(method=krylov, kry_m=30) end to end incl. PCIe of psi0/psi_t
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as g

pkg = g.load_package()
which = sys.argv[1:] or ["cheb", "kpm", "krylov"]


def ev_time(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if "cheb" in which:
    L = int(os.environ.get("SD_CHEB_L", "32"))
    m = pkg.XXZChain(L, nup=L // 2)
    bufs = [torch.randn(m.N, dtype=torch.complex128, device="cuda") for _ in range(4)]
    state = {"i": 0}

    def step():
        i = state["i"]
        pkg.cheb_step(bufs[(i + 2) % 3], bufs[(i + 1) % 3], bufs[i % 3], bufs[3], m, 9.3, -0.4, 0.01 - 0.02j)
        state["i"] = i + 1
    ms = ev_time(step, 8)
    print(json.dumps({"what": "chebyshev term (fused)", "L": L, "N": m.N, "ms": ms, "alg_B_per_row": 80,
                      "achieved_GBs": 80 * m.N / ms / 1e6, "frac_of_8TBs": 80 * m.N / ms / 1e6 / 8000}), flush=True)
    del bufs
    torch.cuda.empty_cache()

if "kpm" in which:
    L = int(os.environ.get("SD_KPM_L", "30"))
    m = pkg.XXZChain(L, nup=L // 2)
    phi = np.random.default_rng(0).standard_normal(m.N) + 0j
    phi /= np.linalg.norm(phi)
    M = 65
    t0 = time.time()
    mu = pkg.compute_chebyshev_moments(pkg.apply_H, phi, M, 9.0, 0.0, m)
    dt = time.time() - t0
    M2 = 17
    t0 = time.time()
    pkg.compute_chebyshev_moments(pkg.apply_H, phi, M2, 9.0, 0.0, m)
    dt2 = time.time() - t0
    per = (dt - dt2) / (M - M2) * 1e3
    print(json.dumps({"what": "KPM moment step (fused, incl. per-step scalar read-back)", "L": L, "N": m.N, "ms": per,
                      "alg_B_per_row": 64, "achieved_GBs": 64 * m.N / per / 1e6, "frac_of_8TBs": 64 * m.N / per / 1e6 / 8000,
                      "mu0": mu[0], "mu1": mu[1]}), flush=True)

if "krylov" in which:
    L = int(os.environ.get("SD_KRY_L", "28"))
    m = pkg.XXZChain(L, nup=L // 2)
    psi0 = np.random.default_rng(1).standard_normal(m.N) + 1j * np.random.default_rng(2).standard_normal(m.N)
    psi0 /= np.linalg.norm(psi0)
    t0 = time.time()
    out = pkg.time_evolve(m, psi0, 0.5, method="krylov", kry_m=30)
    dt = time.time() - t0
    print(json.dumps({"what": "time_evolve krylov kry_m=30 end-to-end (host in/out)", "L": L, "N": m.N, "s": dt,
                      "norm": float(np.linalg.norm(out))}), flush=True)
