"""BASELINE.json configs 2-4 end to end on one MI355X through the public API (host vectors in, host results out).
  config 2: XXZChain L=28 nup=14, time_evolve(method="krylov", kry_m=30)
  config 3: XXZChain L=30 nup=15, dynamical_structure_factor(method="kpm", kpm_m=1024), all L momenta, omega = 0:0.05:5
  config 4: XXZChain L=32 nup=16, time_evolve(method="chebyshev", cheb_n=100), explicit Ebounds
usage: python profiles/config_bench.py [2] [3] [4]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
which = [int(a) for a in sys.argv[1:]] or [2, 3, 4]


def state(n, seed):
    z = np.random.default_rng(seed).standard_normal(2 * n).view(np.complex128)
    z /= np.linalg.norm(z)
    return z


if 2 in which:
    L = 28
    m = pkg.XXZChain(L, nup=L // 2)
    psi0 = state(m.N, 2)
    t0 = time.time()
    out = pkg.time_evolve(m, psi0, 0.5, method="krylov", kry_m=30)
    dt = time.time() - t0
    print(json.dumps({"config": 2, "what": "time_evolve(:krylov, kry_m=30), L=28", "N": m.N, "seconds": dt,
                      "norm": float(np.linalg.norm(out))}), flush=True)
    del psi0, out, m

if 3 in which:
    L = 30
    m = pkg.XXZChain(L, nup=L // 2)
    q = pkg.momenta(m)
    omega = np.arange(0.0, 5.0 + 1e-9, 0.05)
    a, b = L / 2 + 1.0, 0.0            # explicit rescaling: |E| <= L/2 for the Heisenberg chain
    # (i) a REAL psi0 (what groundstate() returns, the reference's own use, examples/example_kpmSqw.jl): every pair
    #     (q, 2pi - q) of momenta(model) is computed once; (ii) a complex psi0: nothing to pair, a few momenta
    for kind in ("real", "complex"):
        if kind == "real":
            psi0 = np.random.default_rng(3).standard_normal(m.N)
            psi0 /= np.linalg.norm(psi0)
            nq = int(os.environ.get("SD_CFG3_NQ", str(len(q))))
        else:
            psi0 = state(m.N, 3)
            nq = int(os.environ.get("SD_CFG3_NQ_COMPLEX", "3"))
        t0 = time.time()
        S = pkg.dynamical_structure_factor(m, psi0, q[:nq], omega, method="kpm", kpm_m=1024, a=a, b=b)
        dt = time.time() - t0
        print(json.dumps({"config": 3, "what": "dynamical_structure_factor(:kpm, kpm_m=1024), L=30, %s psi0, %d momenta, %d omegas"
                                               % (kind, nq, len(omega)),
                          "N": m.N, "seconds": dt, "seconds_per_q": dt / nq, "applies_per_computed_q": 512,
                          "S_finite": bool(np.isfinite(S).all()), "S_min": float(S.min()), "S_max": float(S.max())}), flush=True)
        del psi0, S
    del m

if 4 in which:
    L = 32
    m = pkg.XXZChain(L, nup=L // 2)
    psi0 = state(m.N, 4)
    t0 = time.time()
    out = pkg.time_evolve(m, psi0, 0.5, method="chebyshev", cheb_n=100, Ebounds=(-14.5, 8.5))
    dt = time.time() - t0
    print(json.dumps({"config": 4, "what": "time_evolve(:chebyshev, cheb_n=100), L=32, host psi0 in / psi_t out (2 x 9.6 GB over PCIe)",
                      "N": m.N, "seconds": dt, "norm": float(np.linalg.norm(out))}), flush=True)
    # time stepping: further calls reuse the context's work vectors.  The previous state is kept alive across the timed call
    # (freeing a 9.6 GB numpy array -- munmap -- costs ~0.3-0.4 s of host time that is Python's, not the library's; it is
    # reported separately)
    more, frees = [], []
    for _ in range(2):
        prev = out
        t0 = time.time()
        out = pkg.time_evolve(m, prev, 0.5, method="chebyshev", cheb_n=100, Ebounds=(-14.5, 8.5))
        more.append(time.time() - t0)
        t0 = time.time()
        del prev
        frees.append(time.time() - t0)
    print(json.dumps({"config": 4, "what": "... two further steps of the same evolution (work vectors reused; fresh result array each)",
                      "seconds": more, "freeing_the_previous_state_seconds": frees}), flush=True)

if 5 in which:      # not a BASELINE config: the PublicAPI defaults at scale (Lanczos ground state, Lanczos S(q,w))
    L = 28
    m = pkg.XXZChain(L, nup=L // 2)
    t0 = time.time()
    E0, gs = pkg.groundstate(m, lanc_m=100)
    dt = time.time() - t0
    print(json.dumps({"config": "5a", "what": "groundstate(:lanczos, lanc_m=100), L=28 (full re-orthogonalisation)", "N": m.N,
                      "seconds": dt, "E0_per_site": E0 / L}), flush=True)
    q = pkg.momenta(m)[: L // 2 + 1]
    omega = np.arange(0.0, 5.0 + 1e-9, 0.05)
    t0 = time.time()
    S = pkg.dynamical_structure_factor(m, gs, q, omega, method="lanczos", lanc_m=200, eta=0.05)
    dt = time.time() - t0
    print(json.dumps({"config": "5b", "what": "dynamical_structure_factor(:lanczos, lanc_m=200), L=28 ground state, %d momenta" % len(q),
                      "N": m.N, "seconds": dt, "seconds_per_q": dt / len(q), "S_finite": bool(np.isfinite(S).all()),
                      "S_max": float(S.max())}), flush=True)
