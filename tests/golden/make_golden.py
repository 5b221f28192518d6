#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run from the repo root, CPU only).

The reference is Julia and cannot run in the build container (no Julia runtime), so the vectors
are produced by the INDEPENDENT numpy oracle (oracle/dense.py: Kronecker-product dense H projected on the
itertools.combinations order, scipy-free eigh/expm) -- not by the C restatement and not by the HIP code.
Both of those are then tested against these files.  Each file is < 100 kB.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dense as D  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, L, nup, Jxy, Jz, hz, boundary
    ("L2n1_open", 2, 1, 1.0, 1.0, 0.0, "open"),
    ("L4n2_open", 4, 2, 1.0, 1.0, 0.0, "open"),
    ("L6n3_open", 6, 3, 1.0, 1.0, 0.0, "open"),
    ("L6n3_per", 6, 3, 1.0, 1.0, 0.0, "periodic"),
    ("L8n3_field", 8, 3, 1.3, 0.7, 0.2, "open"),
    ("L10n5_open", 10, 5, 1.0, 1.0, 0.0, "open"),
    ("L10n4_per", 10, 4, 0.8, -1.1, 0.05, "periodic"),
    ("L12n6_open", 12, 6, 1.0, 1.0, 0.0, "open"),
    ("L7full", 7, None, 1.0, 0.6, 0.1, "open"),
]


def jackson(M):
    n = np.arange(M)
    return ((M - n + 1) * np.cos(np.pi * n / (M + 1)) + np.sin(np.pi * n / (M + 1)) / np.tan(np.pi / (M + 1))) / (M + 1)


def kpm_moments_dense(Ht, phi, M):
    mu = np.zeros(M)
    v0, v1 = phi.copy(), Ht @ phi
    mu[0] = np.vdot(phi, v0).real
    mu[1] = np.vdot(phi, v1).real
    for m in range(2, M):
        v2 = 2 * (Ht @ v1) - v0
        mu[m] = np.vdot(phi, v2).real
        v0, v1 = v1, v2
    return mu


def kpm_reconstruct(mu, omega, a, b, E0):
    M = len(mu)
    S = np.zeros(len(omega))
    for iw, w in enumerate(omega):
        x = (w + E0 - b) / a
        if abs(x) >= 1:
            continue
        T = np.cos(np.arange(M) * np.arccos(x))
        S[iw] = max(0.0, (mu[0] * T[0] + 2 * np.dot(mu[1:], T[1:])) / (a * np.pi * np.sqrt(1 - x * x)))
    return S


def main():
    for (name, L, nup, Jxy, Jz, hz, bc) in CASES:
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        hop, zz, f = D.xxz_lists(L, Jxy, Jz, hz, bc)
        H = D.dense_H(L, nup, hop, zz, f)
        N = H.shape[0]
        states = D.sector_states(L, nup) if nup is not None else np.arange(N, dtype=np.uint64)
        psi_c = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        psi_r = rng.standard_normal(N)
        out = dict(L=L, nup=-1 if nup is None else nup, Jxy=Jxy, Jz=Jz, hz=hz, periodic=int(bc == "periodic"),
                   states=states, psi_c=psi_c, psi_r=psi_r, Hpsi_c=H @ psi_c, Hpsi_r=H @ psi_r)
        if N <= 300:
            out["H"] = H
        w = np.linalg.eigvalsh(H)
        out["evals_minmax"] = np.array([w[0], w[-1]])
        qs = np.array([0.0, np.pi / 3, np.pi])
        out["q"] = qs
        out["szq_c"] = np.stack([D.szq_diag(L, states, q) * psi_c for q in qs])
        t = 0.3
        out["t"] = t
        psi0 = psi_c / np.linalg.norm(psi_c)
        out["psi0"] = psi0
        out["expm_psi0"] = D.expm_herm(H, t) @ psi0
        # KPM with explicit (a, b): moments, Jackson kernel, S(q, w) for q = pi
        Emin, Emax = w[0], w[-1]
        a, b = (Emax - Emin) / (2 * 0.99) if Emax > Emin else 1.0, (Emax + Emin) / 2
        M = 48
        Ht = (H - b * np.eye(N)) / a
        gs = np.linalg.eigh(H)[1][:, 0]
        E0 = w[0]
        phi = D.szq_diag(L, states, np.pi) * gs.astype(complex)
        nphi = np.linalg.norm(phi)
        out["gs"] = gs
        out["kpm_ab"] = np.array([a, b])
        out["kpm_M"] = M
        if nphi > 1e-12:
            mu = kpm_moments_dense(Ht, phi / nphi, M)
            omega = np.linspace(0.0, 4.0, 41)
            out["kpm_mu"] = mu
            out["kpm_omega"] = omega
            out["kpm_S_pi"] = nphi ** 2 * kpm_reconstruct(mu * jackson(M), omega, a, b, E0)
            out["kpm_norm_phi"] = nphi
        out["jackson"] = jackson(M)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, N, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
