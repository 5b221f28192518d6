#!/usr/bin/env python3
"""The reference's own time-evolution example (examples/example_time_evolution.jl: XXZChain(L=15, Jz=0.5, nup=14), one flipped spin in
the middle, 150 times in [0, 5]; exact propagator from the columns of apply_H!, time_evolve(:chebyshev, cheb_n=20) and
time_evolve(:krylov, kry_m=15) step by step, magnetization_per_site and the fidelities against the exact state) through the Python
mirror of the SpinDynamics.jl interface.  Same calls, same keywords; the states stay on the device between the steps when `device` is
given (python examples/time_evolution.py [L] [device]).  Prints timings, the worst fidelity and the spin front; no plotting."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg
import __graft_entry__ as g

sd = g.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 15
on_device = len(sys.argv) > 2 and sys.argv[2] == "device"
nup = L - 1
model = sd.XXZChain(L, Jxy=1.0, Jz=0.5, nup=nup)
middle_site = (L + 1) // 2
psi0 = sd.polarized_state_with_flips(model, [middle_site]).astype(np.complex128)
N = len(model)
print("Hilbert-space dimension:", N)
print("Flipped site:", middle_site)

# exact Hamiltonian from the columns of apply_H!, as the reference's script builds it
H = np.zeros((N, N), dtype=np.complex128)
e, col = np.zeros(N, dtype=np.complex128), np.zeros(N, dtype=np.complex128)
for j in range(N):
    e[:] = 0
    e[j] = 1
    sd.apply_H(col, e, model)
    H[:, j] = col
assert np.abs(H - H.conj().T).max() == 0.0

times = np.linspace(0.0, 5.0, 150)
dt = times[1] - times[0]
U = scipy.linalg.expm(-1j * dt * H)
mags = {k: np.empty((L, len(times))) for k in ("exact", "cheb", "krylov")}
fid = {k: np.ones(len(times)) for k in ("cheb", "krylov")}
psi = {"exact": psi0.copy(), "cheb": psi0.copy(), "krylov": psi0.copy()}
if on_device:
    import torch
    psi["cheb"], psi["krylov"] = torch.from_numpy(psi0).cuda(), torch.from_numpy(psi0).cuda()
host = (lambda x: x.cpu().numpy()) if on_device else (lambda x: x)
for k in mags:
    mags[k][:, 0] = sd.magnetization_per_site(psi0, model)

t0 = time.time()
for n in range(len(times) - 1):
    psi["exact"] = U @ psi["exact"]
    psi["exact"] /= np.linalg.norm(psi["exact"])
    psi["cheb"] = sd.time_evolve(model, psi["cheb"], dt, method="chebyshev", cheb_n=20)
    psi["krylov"] = sd.time_evolve(model, psi["krylov"], dt, method="krylov", kry_m=15)
    for k in mags:
        mags[k][:, n + 1] = sd.magnetization_per_site(psi[k], model)
    for k in fid:
        fid[k][n + 1] = abs(np.vdot(psi["exact"], host(psi[k]))) ** 2
print("%d steps of both methods + observables: %.3f s" % (len(times) - 1, time.time() - t0))
print("worst fidelity: chebyshev %.12f   krylov %.12f" % (fid["cheb"].min(), fid["krylov"].min()))
print("max |<Sz_i>(t) - exact|: chebyshev %.2e   krylov %.2e" % (np.abs(mags["cheb"] - mags["exact"]).max(),
                                                                np.abs(mags["krylov"] - mags["exact"]).max()))
front = [int(np.argmax(np.abs(mags["exact"][:, n] - 0.5) > 1e-3)) + 1 for n in (0, 37, 74, 149)]
print("leftmost site the flipped spin has reached at t = 0, 1.24, 2.48, 5:", front)
assert fid["cheb"].min() > 1 - 1e-8 and fid["krylov"].min() > 1 - 1e-8
assert abs(mags["exact"].sum(axis=0) - (nup - L / 2)).max() < 1e-10          # total S^z is conserved
