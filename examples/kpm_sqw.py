#!/usr/bin/env python3
"""The reference's own KPM example (examples/example_kpmSqw.jl: XXZChain(L=20, nup=10), groundstate(lanc_m=100),
dynamical_structure_factor(method=:kpm, kpm_m=80, kernel=:jackson) over all momenta and 100 frequencies) through the Python mirror of
the SpinDynamics.jl interface, i.e. through the C ABI and the HIP kernels.  The same three calls with the reference's names and
keywords; `L` may be raised (python examples/kpm_sqw.py 28).  Prints timings and a few values; no plotting."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

sd = g.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20

model = sd.XXZChain(L, Jxy=1.0, Jz=1.0, nup=L // 2)
print("Hilbert-space dimension:", len(model))

t0 = time.time()
E0, psi0 = sd.groundstate(model, lanc_m=100)
print("groundstate: %.3f s   E0 = %.12f   E0/L = %.6f" % (time.time() - t0, E0, E0 / L))

q = sd.momenta(model)
omega = np.linspace(0.0, 5.0, 100)
t0 = time.time()
S = sd.dynamical_structure_factor(model, psi0, q, omega, method="kpm", kpm_m=80, kernel="jackson")
dt = time.time() - t0
print("dynamical_structure_factor(:kpm, kpm_m=80): %.3f s for %d momenta x %d frequencies" % (dt, len(q), len(omega)))
# the same call with one momentum at a time (what the recursion looked like before the momenta shared their launches)
model.ctx.set_q_batch(False)
t0 = time.time()
S1 = sd.dynamical_structure_factor(model, psi0, q, omega, method="kpm", kpm_m=80, kernel="jackson")
dt1 = time.time() - t0
model.ctx.set_q_batch(True)
print("  one momentum at a time: %.3f s (x%.1f); max |S - S_one_at_a_time| = %.1e" % (dt1, dt1 / dt, np.abs(S - S1).max()))
iq = L // 2
print("S(pi, w) peaks at w = %.3f with %.5f;  sum over w of S(pi, w) dw = %.5f" % (
    omega[np.argmax(S[iq])], S[iq].max(), S[iq].sum() * (omega[1] - omega[0])))
assert np.isfinite(S).all() and S.shape == (len(q), len(omega))
