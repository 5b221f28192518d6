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
which = sys.argv[1:] or ["cheb", "kpm", "krylov", "lanczos"]


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
    # the pair form used by sd_chebyshev_evolve / the sharded driver: recurrence-only term + two-term accumulation
    op = pkg.ShardedOperator(m, 0, 1)
    halo = op.halo(bufs[0])
    state["i"] = 0

    def pair():
        i = state["i"]
        op._launch(bufs[(i + 2) % 3], bufs[(i + 1) % 3], halo, 3, a=9.3, b=-0.4, prev=bufs[i % 3])
        op._launch(bufs[i % 3], bufs[(i + 2) % 3], halo, 4, a=9.3, b=-0.4, c=0.01 - 0.02j, c0=0.02 + 0.01j,
                   prev=bufs[(i + 1) % 3], acc=bufs[3])
        state["i"] = i + 2
    ms2 = ev_time(pair, 4) / 2
    print(json.dumps({"what": "chebyshev term, pair form (72 B/row per term)", "L": L, "N": m.N, "ms": ms2, "alg_B_per_row": 72,
                      "achieved_GBs": 72 * m.N / ms2 / 1e6, "frac_of_8TBs": 72 * m.N / ms2 / 1e6 / 8000}), flush=True)
    del bufs, halo
    torch.cuda.empty_cache()

if "kpm" in which:
    L = int(os.environ.get("SD_KPM_L", "30"))
    m = pkg.XXZChain(L, nup=L // 2)
    phi = np.random.default_rng(0).standard_normal(m.N) + 0j
    phi /= np.linalg.norm(phi)
    a_kpm = L / 2 + 1.0                       # |E| <= L/2 for the Heisenberg chain: spectrum inside (-1, 1)
    pkg.compute_chebyshev_moments(pkg.apply_H, phi, 5, a_kpm, 0.0, m)        # warm-up (allocations, first touch)
    for doubling, bpr in ((True, 48), (False, 64)):
        m.ctx.set_kpm_doubling(doubling)
        M, M2 = 129, 33
        t0 = time.time()
        mu = pkg.compute_chebyshev_moments(pkg.apply_H, phi, M, a_kpm, 0.0, m)
        dt = time.time() - t0
        t0 = time.time()
        pkg.compute_chebyshev_moments(pkg.apply_H, phi, M2, a_kpm, 0.0, m)
        dt2 = time.time() - t0
        per_moment = (dt - dt2) / (M - M2) * 1e3
        per_apply = per_moment * (2 if doubling else 1)
        print(json.dumps({"what": "KPM moments, %s (fused step)"
                                  % ("2 per apply (default)" if doubling else "reference loop, 1 per apply"),
                          "L": L, "N": m.N, "ms_per_moment": per_moment, "ms_per_apply_step": per_apply,
                          "alg_B_per_row_step": bpr, "achieved_GBs": bpr * m.N / per_apply / 1e6,
                          "frac_of_8TBs": bpr * m.N / per_apply / 1e6 / 8000, "mu0": mu[0], "mu1": mu[1],
                          "mu_max": float(np.abs(mu).max())}), flush=True)
    m.ctx.set_kpm_doubling(True)

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

if "lanczos" in which:
    # Lanczos step on the device (apply with fused <u|Hu>; one pass w = Hv - a v - b v_prev on un-normalised vectors with fused norm), L=32
    L = int(os.environ.get("SD_LAN_L", "32"))
    m = pkg.XXZChain(L, nup=L // 2)
    pkg.lanczos_extremal(pkg.apply_H, m, lanc_m=3, seed=1)
    ts = {}
    for lm in (6, 22):
        t0 = time.time()
        lo, hi = pkg.lanczos_extremal(pkg.apply_H, m, lanc_m=lm, seed=1)
        ts[lm] = time.time() - t0
    per = (ts[22] - ts[6]) / 16 * 1e3
    print(json.dumps({"what": "Lanczos step (lanczos_extremal, generated start vector)", "L": L, "N": m.N, "ms_per_step": per,
                      "alg_B_per_row": 96, "achieved_GBs": 96 * m.N / per / 1e6, "frac_of_8TBs": 96 * m.N / per / 1e6 / 8000,
                      "Emin_22": lo, "Emax_22": hi}), flush=True)
