"""GPU parity of the orbit-group apply kernel (k_apply_orbit, kernels_orbit.hip): the default path of unsharded
open-chain sectors with at least 2^24 rows, forced here for small systems with SD_ORBIT=1 so that the CPU oracle can
check every row.  Same per-row operation order as the reference -> BIT-EXACT (np.array_equal)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rand_vec(n, seed, complex_=True):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(n)
    if complex_:
        v = v + 1j * rng.standard_normal(n)
    return v


CASES = [
    # (L, nup, Jxy, Jz, hz)   -- p = L - 10 prefix sites; orbits of 1..16 tiles occur from p >= 8
    (13, 6, 1.0, 1.0, 0.0),
    (14, 7, 1.0, 1.0, 0.0),
    (16, 8, 1.0, 1.0, 0.0),
    (17, 5, 1.0, 0.7, 0.0),          # list-order diagonal (Jz/4 multiples not exact), small sectors
    (18, 9, 0.9, 0.7, 0.3),          # fields, non-power-of-two hop amplitude (separate multiply-add)
    (19, 12, 1.0, 1.0, 0.0),         # suffix fillings up to 10: one-row tiles
    (20, 10, 1.0, 1.0, 0.0),
    (21, 10, 1.3, -0.4, 0.0),
    (22, 11, 1.0, 1.0, 0.0),
    (22, 3, 1.0, 1.0, 0.1),          # few up spins: many infeasible prefixes, short tiles only
]


@pytest.mark.parametrize("tb", ["8", "4"])
@pytest.mark.parametrize("L,nup,Jxy,Jz,hz", CASES)
def test_orbit_apply_bit_exact_vs_oracle(pkg, O, L, nup, Jxy, Jz, hz, tb, monkeypatch):
    monkeypatch.setenv("SD_ORBIT", "1")
    monkeypatch.setenv("SD_ORB_TB", tb)
    monkeypatch.setenv("SD_ORB_CHUNK", "3")
    m = pkg.XXZChain(L, Jxy=Jxy, Jz=Jz, hz=hz, nup=nup)
    r = O.XXZChain(L, Jxy=Jxy, Jz=Jz, hz=hz, nup=nup)
    assert m.device_path == "orbit"
    for cplx in (True, False):
        psi = rand_vec(m.N, 300 + L, cplx)
        out = np.empty_like(psi)
        pkg.apply_H(out, psi, m)
        want = O.apply_H(r, psi)
        assert np.array_equal(out, want), f"max diff {np.abs(out - want).max()}"
    a, b = 6.5, -0.25
    psi = rand_vec(m.N, 17)
    out = np.empty_like(psi)
    pkg.apply_rescaled_H(out, psi, pkg.apply_H, m, a, b)
    assert np.array_equal(out, O.apply_rescaled_H(r, psi, a, b))


def test_orbit_matches_tiled_kernel_and_recursions(pkg, O, monkeypatch):
    """Same model through both kernels: identical bits; the fused epilogues (KPM sums, Chebyshev pairs, Lanczos dot)
    run through the orbit kernel and agree with the oracle at the recursion tolerances."""
    L, nup = 20, 10
    r = O.XXZChain(L, nup=nup, Jz=0.8)
    monkeypatch.setenv("SD_ORBIT", "0")
    m_t = pkg.XXZChain(L, nup=nup, Jz=0.8)
    assert m_t.device_path == "tiled"
    monkeypatch.setenv("SD_ORBIT", "1")
    m_o = pkg.XXZChain(L, nup=nup, Jz=0.8)
    assert m_o.device_path == "orbit"
    psi = rand_vec(m_o.N, 5)
    o1, o2 = np.empty_like(psi), np.empty_like(psi)
    pkg.apply_H(o1, psi, m_t)
    pkg.apply_H(o2, psi, m_o)
    assert np.array_equal(o1, o2)
    phi = psi / np.linalg.norm(psi)
    mu = pkg.compute_chebyshev_moments(pkg.apply_H, phi, 24, 7.0, -0.3, m_o)
    assert np.abs(mu - O.compute_chebyshev_moments(r, phi, 24, 7.0, -0.3)).max() <= 1e-12
    got = pkg.time_evolve(m_o, phi, 0.3, method="chebyshev", cheb_n=40, Ebounds=(-9.0, 6.0))
    want = O.chebyshev_time_evolve(r, phi, 0.3, cheb_n=40, Ebounds=(-9.0, 6.0))
    assert np.abs(got - want).max() <= 1e-13
    got = pkg.time_evolve(m_o, phi, 0.2, method="krylov", kry_m=20)
    want = O.krylov_time_evolve(r, phi, 0.2, kry_m=20)
    assert np.abs(got - want).max() <= 1e-11
