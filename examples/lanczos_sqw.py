#!/usr/bin/env python3
"""The reference's own Lanczos example (examples/example_lanczosSqw.jl: XXZChain(L=16, nup=8), groundstate(lanc_m=100),
dynamical_structure_factor(method=:lanczos, lanc_m=100, eta=0.05) over all momenta and 100 frequencies) through the Python mirror of the
SpinDynamics.jl interface.  Same calls, same keywords; `L` may be raised (python examples/lanczos_sqw.py 24).  No plotting."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

sd = g.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = sd.XXZChain(L, Jxy=1.0, Jz=1.0, nup=L // 2)
print("Hilbert-space dimension:", len(model))

t0 = time.time()
E0, psi0 = sd.groundstate(model, lanc_m=100)
print("groundstate: %.3f s   Ground-state energy: %.12f" % (time.time() - t0, E0))

q = sd.momenta(model)
omega = np.linspace(0.0, 5.0, 100)
t0 = time.time()
S = sd.dynamical_structure_factor(model, psi0, q, omega, method="lanczos", lanc_m=100, eta=0.05)
dt = time.time() - t0
print("dynamical_structure_factor(:lanczos, lanc_m=100, eta=0.05): %.3f s for %d momenta x %d frequencies" % (dt, len(q), len(omega)))
iq = L // 2
print("S(pi, w) peaks at w = %.3f with %.5f;  S(q=0, w) max = %.2e (total S^z is conserved)" % (
    omega[np.argmax(S[iq])], S[iq].max(), S[0].max()))

assert S.shape == (len(q), len(omega)) and np.isfinite(S).all() and S.min() >= 0.0

# the other two calls of the reference's README quick start: real-time evolution and the static structure factor
t0 = time.time()
psi_t = sd.time_evolve(model, psi0, 0.5, method="krylov")
Sq = sd.structure_factor(model, psi0)
print("time_evolve(:krylov, t = 0.5) + structure_factor: %.3f s   |<psi0|psi_t>| = %.12f   S(q = pi) = %.6f" % (
    time.time() - t0, abs(np.vdot(psi0, psi_t)), Sq[max(Sq, key=lambda k: abs(k - np.pi) < 1e-12)]))
assert abs(abs(np.vdot(psi0, psi_t)) - 1.0) < 1e-8          # an eigenstate only picks up a phase
