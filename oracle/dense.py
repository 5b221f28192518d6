"""Independent numpy oracle -- TEST INFRASTRUCTURE ONLY.

Builds the dense Hamiltonian from Kronecker products of spin-1/2 matrices on
the full 2^L space (no bit tricks, no ranking), then restricts it to the
sector basis enumerated with itertools.combinations -- the order the
reference's Combinatorics.combinations(1:L, nup) produces
(src/Basis.jl:37-53).  It shares no code with spin_oracle.c, the HIP kernels
or the reference's apply_H!, so it cross-checks all three.

Convention (src/Hamiltonian.jl:19-29, src/Basis.jl:41-47): site i (1-based) is
bit i-1 of the integer state; bit 1 = up = +1/2.
"""
import itertools

import numpy as np

SZ = np.array([[-0.5, 0.0], [0.0, 0.5]])        # index 0 = bit 0 = down
SP = np.array([[0.0, 0.0], [1.0, 0.0]])         # S+ |down> = |up>
SM = SP.T.copy()
I2 = np.eye(2)


def site_op(op, site, L):
    """op acting on 1-based `site`; integer state index = sum bit_k 2^k, so
    site 1 is the LEAST significant factor => rightmost in the Kronecker chain."""
    mats = [I2] * L
    mats[L - site] = op
    out = mats[0]
    for m in mats[1:]:
        out = np.kron(out, m)
    return out


def dense_full_H(L, hopping, zz, field):
    """H = sum_i h_i Sz_i + sum Jz Sz_i Sz_j + sum Jxy (S+_i S-_j + S-_i S+_j)
    (the operator apply_H! implements, src/Hamiltonian.jl:228-267)."""
    dim = 1 << L
    H = np.zeros((dim, dim))
    for i in range(1, L + 1):
        if field[i - 1] != 0.0:
            H += field[i - 1] * site_op(SZ, i, L)
    for (i, j, Jz) in zz:
        H += Jz * site_op(SZ, i, L) @ site_op(SZ, j, L)
    for (i, j, J) in hopping:
        H += J * (site_op(SP, i, L) @ site_op(SM, j, L) + site_op(SM, i, L) @ site_op(SP, j, L))
    return H


def sector_states(L, nup):
    """States in the reference order: lexicographic combinations of 1-based sites."""
    out = []
    for comb in itertools.combinations(range(1, L + 1), nup):
        s = 0
        for i in comb:
            s |= 1 << (i - 1)
        out.append(s)
    return np.array(out, dtype=np.uint64)


def dense_H(L, nup, hopping, zz, field):
    H = dense_full_H(L, hopping, zz, field)
    if nup is None:
        return H
    st = sector_states(L, nup).astype(np.int64)
    return H[np.ix_(st, st)]


def xxz_lists(L, Jxy=1.0, Jz=1.0, hz=0.0, boundary="open"):
    hopping = [(i, i + 1, Jxy / 2) for i in range(1, L)]
    zz = [(i, i + 1, Jz) for i in range(1, L)]
    if boundary == "periodic" and L > 2:
        hopping.append((L, 1, Jxy / 2))
        zz.append((L, 1, Jz))
    return hopping, zz, [hz] * L


def szq_diag(L, states, q):
    """diag of S^z_q = L^{-1/2} sum_r e^{iqr} S^z_{r+1} on the given states."""
    d = np.zeros(len(states), dtype=np.complex128)
    for r in range(L):
        bits = (states.astype(np.uint64) >> np.uint64(r)) & np.uint64(1)
        d += np.exp(1j * q * r) * (bits.astype(np.float64) - 0.5)
    return d / np.sqrt(L)


def expm_herm(H, t):
    w, v = np.linalg.eigh(H)
    return (v * np.exp(-1j * w * t)) @ v.conj().T
