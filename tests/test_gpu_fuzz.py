"""Seeded random models against the oracle, bit for bit: random chain lengths and sectors, random tile sizes
(SD_SUFFIX_BITS), forced tile length classes, random extra bonds / couplings / fields (general-bond path, exact-order
diagonal, non-power-of-two hops -> no FMA), full 2^L basis on both of its kernels, both element types, plus the rescaled
epilogue and Sz_q on the same vectors."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_model(rng):
    full = rng.random() < 0.25
    L = int(rng.integers(2, 15 if full else 21))
    nup = None if full else int(rng.integers(0, L + 1))
    kind = rng.integers(0, 4)
    if kind == 0:       # plain XXZ chain, dyadic couplings (FMA path, closed-form diagonal)
        Jxy, Jz, hz = float(rng.choice([1.0, 2.0, 0.5])), float(rng.choice([1.0, 0.5, -1.0, 0.0])), 0.0
        hop = [(i, i + 1, Jxy / 2) for i in range(1, L)]
        zz = [(i, i + 1, Jz) for i in range(1, L)]
        f = np.full(L, hz)
    elif kind == 1:     # chain with irrational couplings and a field (no FMA, exact-order diagonal)
        hop = [(i, i + 1, float(rng.normal())) for i in range(1, L)]
        zz = [(i, i + 1, float(rng.normal())) for i in range(1, L)]
        f = rng.normal(size=L)
    elif kind == 2:     # chain + random further bonds (periodic / long range), listed after the chain
        hop = [(i, i + 1, 0.5) for i in range(1, L)]
        zz = [(i, i + 1, 0.75) for i in range(1, L)]
        for _ in range(int(rng.integers(1, 4))):
            i, j = sorted(rng.choice(np.arange(1, L + 1), size=2, replace=False).tolist()) if L >= 2 else (1, 1)
            if i != j:
                hop.append((int(i), int(j), float(rng.normal())))
                zz.append((int(i), int(j), float(rng.normal())))
        f = np.zeros(L)
    else:               # arbitrary bond list in random order (no leading chain: everything through the general path)
        pairs = [(i, j) for i in range(1, L + 1) for j in range(i + 1, L + 1)]
        rng.shuffle(pairs)
        pairs = pairs[: max(1, min(len(pairs), int(rng.integers(1, 2 * L))))]
        hop = [(i, j, float(rng.normal())) for (i, j) in pairs]
        zz = [(i, j, float(rng.normal())) for (i, j) in pairs[::2]]
        f = rng.normal(size=L)
    return L, nup, hop, zz, f


@pytest.mark.parametrize("seed", range(int(os.environ.get("SD_FUZZ_N", "40"))))   # SD_FUZZ_N=4000 for a long hunt (passes; 3 min)
def test_random_models_bit_exact(pkg, O, seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    L, nup, hop, zz, f = random_model(rng)
    monkeypatch.setenv("SD_SUFFIX_BITS", str(int(rng.integers(3, 14))))
    monkeypatch.setenv("SD_LEN_CLASSES", str(int(rng.choice([1, 2]))))
    m = pkg.build_model(L, nup=nup, hopping=hop, onsite_field=f, zz=zz)
    r = O.build_model(L, nup=nup, hopping=hop, onsite_field=f, zz=zz)
    assert m.N == r.N
    for cplx in (True, False):
        psi = rng.standard_normal(m.N) + (1j * rng.standard_normal(m.N) if cplx else 0)
        out = np.empty_like(psi)
        pkg.apply_H(out, psi, m)
        want = O.apply_H(r, psi)
        assert np.array_equal(out, want), (seed, L, nup, m.device_path, float(np.abs(out - want).max()))
        a, b = float(rng.uniform(1, 9)), float(rng.normal())
        pkg.apply_rescaled_H(out, psi, pkg.apply_H, m, a, b)
        assert np.array_equal(out, O.apply_rescaled_H(r, psi, a, b)), (seed, "rescaled")
    q = float(rng.uniform(0, 2 * np.pi))
    phi = pkg.Sz_q_vector(m, psi, q)
    want = O.Sz_q_vector(r, psi, q)
    assert np.abs(phi - want).max() <= 1e-15 * max(1.0, float(np.abs(want).max()))
