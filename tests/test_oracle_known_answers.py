"""Pins the CPU oracle (oracle/spin_oracle.c) against every known-answer test the reference's own test suite
holds for the hot path (file:line under the reference's test/ directory), and against
the independent numpy dense oracle.  CPU only."""
import itertools

import numpy as np
import pytest


def dense_from_apply(O, m, dtype=float):
    """test/test_PublicAPI.jl:13-22: dense H column by column from apply_H! on unit vectors."""
    H = np.zeros((m.N, m.N), dtype=dtype)
    for j in range(m.N):
        e = np.zeros(m.N, dtype=dtype)
        e[j] = 1
        H[:, j] = O.apply_H(m, e)
    return H


def test_two_site_sector_matrix(O):
    # (1) test/test_PublicAPI.jl:5-28
    m = O.XXZChain(2, Jxy=1.0, Jz=1.0, nup=1)
    assert m.L == 2 and m.nup == 1 and m.mode == "sector" and m.N == 2
    H = dense_from_apply(O, m)
    assert np.allclose(H, [[-0.25, 0.5], [0.5, -0.25]], atol=0)
    assert np.allclose(np.linalg.eigvalsh(H), [-0.75, 0.25])


def test_momenta(O):
    # test/test_PublicAPI.jl:30-37
    q = O.momenta(O.XXZChain(6, nup=3))
    assert len(q) == 6 and np.allclose(q, 2 * np.pi * np.arange(6) / 6)


def test_groundstate_two_site(O):
    # (2) test/test_PublicAPI.jl:40-53
    m = O.XXZChain(2, nup=1)
    rng = np.random.default_rng(0)
    E0, psi = O.lanczos_groundstate(m, rng.standard_normal(2), lanc_m=2)
    assert abs(E0 + 0.75) <= 1e-12
    assert abs(np.linalg.norm(psi) - 1) <= 1e-12
    assert np.linalg.norm(O.apply_H(m, psi) - E0 * psi) < 1e-10


def test_krylov_two_site(O, D):
    # (3) test/test_PublicAPI.jl:56-93
    m = O.XXZChain(2, nup=1)
    H = np.array([[-0.25, 0.5], [0.5, -0.25]])
    psi0 = np.array([1.0, 0.0], dtype=complex)
    got = O.krylov_time_evolve(m, psi0, 0.3, kry_m=2)
    assert np.allclose(got, D.expm_herm(H, 0.3) @ psi0, atol=1e-10)
    assert abs(np.linalg.norm(got) - 1) <= 1e-12
    assert np.allclose(O.krylov_time_evolve(m, psi0, 0.0, kry_m=2), psi0, atol=1e-12)


def test_chebyshev_two_site(O, D):
    # (4) test/test_PublicAPI.jl:96-118
    m = O.XXZChain(2, nup=1)
    H = np.array([[-0.25, 0.5], [0.5, -0.25]])
    psi0 = np.array([1.0, 0.0], dtype=complex)
    got = O.chebyshev_time_evolve(m, psi0, 0.3, cheb_n=30, Ebounds=(-0.75, 0.25))
    assert np.allclose(got, D.expm_herm(H, 0.3) @ psi0, atol=1e-8)
    assert abs(np.linalg.norm(got) - 1) <= 1e-8


def test_lanczos_groundstate_L6(O):
    # (5) test/test_Lanczos.jl:29-53
    m = O.XXZChain(6, nup=3)
    H = dense_from_apply(O, m)
    w = np.linalg.eigvalsh(H)
    rng = np.random.default_rng(1)
    E0, psi = O.lanczos_groundstate(m, rng.standard_normal(m.N), lanc_m=m.N)
    assert abs(E0 - w[0]) <= 1e-12
    assert np.linalg.norm(O.apply_H(m, psi) - E0 * psi) < 1e-10


def test_lanczos_extremal_L4(O):
    # (6) test/test_Lanczos.jl:74-100
    m = O.XXZChain(4, nup=2)
    w = np.linalg.eigvalsh(dense_from_apply(O, m))
    rng = np.random.default_rng(2)
    lo, hi = O.lanczos_extremal(m, rng.standard_normal(m.N) + 1j * rng.standard_normal(m.N), lanc_m=m.N)
    assert abs(lo - w[0]) <= 1e-12 and abs(hi - w[-1]) <= 1e-12


def test_lanczos_tridiag_alpha1(O):
    # (7),(8) test/test_Lanczos.jl:6-26, 103-119
    m = O.XXZChain(2, nup=1)
    v = np.array([1.0, 1.0j]) / np.sqrt(2)
    alpha, beta, nv = O.lanczos_tridiag(m, v, lanc_m=2)
    H = np.array([[-0.25, 0.5], [0.5, -0.25]])
    assert abs(alpha[0] - np.real(np.vdot(v, H @ v))) <= 1e-14
    assert abs(nv - 1) <= 1e-14
    m6 = O.XXZChain(6, nup=3)
    rng = np.random.default_rng(3)
    a6, b6, _ = O.lanczos_tridiag(m6, rng.standard_normal(20) + 0j, lanc_m=50)
    assert len(b6) == len(a6) - 1 and len(a6) <= m6.N


def test_szq_vs_site_operators(O, D):
    # (9) test/test_Hamiltonian.jl:93-110: Sz_q psi0 == sum_r e^{iq(r-1)}/sqrt(L) S^z_r psi0, q = pi/3, L = 6
    L, q = 6, np.pi / 3
    m = O.XXZChain(L, nup=3)
    rng = np.random.default_rng(4)
    psi0 = rng.standard_normal(m.N) + 1j * rng.standard_normal(m.N)
    st = m.states
    want = np.zeros(m.N, dtype=complex)
    for r in range(1, L + 1):
        sz = np.array([0.5 if (int(s) >> (r - 1)) & 1 else -0.5 for s in st])
        want += np.exp(1j * q * (r - 1)) / np.sqrt(L) * sz * psi0
    assert np.allclose(O.Sz_q_vector(m, psi0, q), want, atol=1e-12)


def test_bit_helpers(O):
    # (10) test/test_Hamiltonian.jl:16-20
    l = O.lib()
    assert l.so_bit_at(0x05, 0) == 1 and l.so_bit_at(0x05, 1) == 0
    assert l.so_flip_bits(0x05, 0, 1) == 0x06
    assert l.so_sz_value(1) == 0.5 and l.so_sz_value(0) == -0.5


def test_kpm_rescaling(O):
    # (11) test/test_KPM.jl:27-41
    a, b = O.rescaling_from_bounds(-3.0, 5.0)
    assert abs((-3.0 - b) / a + 0.99) <= 1e-12 and abs((5.0 - b) / a - 0.99) <= 1e-12


def test_kpm_sum_rule_and_positivity(O):
    # (12),(13) test/test_KPM.jl:44-91: L=6, M=120, Jackson, w = 0:0.01:5, rtol 5e-3
    m = O.XXZChain(6, nup=3)
    rng = np.random.default_rng(5)
    E0, gs = O.lanczos_groundstate(m, rng.standard_normal(m.N), lanc_m=20)
    lo, hi = O.estimate_energy_bounds(m, rng.standard_normal(m.N) + 0j, rng.standard_normal(m.N) + 0j, lanc_m=20)
    a, b = O.rescaling_from_bounds(lo, hi)
    assert -1 < (lo - b) / a and (hi - b) / a < 1          # test/test_KPM.jl:4-24
    omega = np.arange(0.0, 5.0 + 1e-9, 0.01)
    S = O.kpm_sqw(m, gs, [np.pi], omega, a, b, kpm_m=120, kernel="jackson")
    phi = O.Sz_q_vector(m, gs, np.pi)
    assert np.all(np.isfinite(S)) and np.all(S >= 0)
    assert abs(S[0].sum() * 0.01 - np.linalg.norm(phi) ** 2) <= 5e-3 * np.linalg.norm(phi) ** 2
    S2 = O.kpm_sqw(m, gs, [np.pi], np.linspace(0, 5, 300), a, b, kpm_m=100)
    assert S2[:, -11:].max() < S2.max()


def test_sqw_shapes(O):
    # (14) test/test_PublicAPI.jl:154-203
    m = O.XXZChain(4, nup=2)
    rng = np.random.default_rng(6)
    _, gs = O.lanczos_groundstate(m, rng.standard_normal(m.N), lanc_m=6)
    q = O.momenta(m)
    S = O.lanczos_sqw(m, gs, q, np.linspace(0, 3, 40), lanc_m=6, eta=0.05)
    assert S.shape == (4, 40) and np.all(np.isfinite(S)) and np.all(S >= -1e-12)
    lo, hi = -1.0, 0.75
    a, b = O.rescaling_from_bounds(lo, hi)
    S = O.kpm_sqw(m, gs, q, np.linspace(-2, 2, 40), a, b, kpm_m=40)
    assert S.shape == (4, 40) and np.all(np.isfinite(S))


def test_sector_edges_and_validation(O):
    # (15) test/test_Basis.jl:4-19
    assert list(O.XXZChain(5, nup=0).states) == [0]
    s = O.XXZChain(5, nup=5).states
    assert len(s) == 1 and bin(int(s[0])).count("1") == 5
    for bad in [dict(L=0, nup=0), dict(L=64, nup=1), dict(L=4, nup=5)]:
        with pytest.raises(O.OracleError):
            O.Model(bad["L"], bad["nup"])


def test_popcount_and_plumbing(O):
    # test/test_SpinModel.jl:9-49
    m = O.XXZChain(6, Jxy=2.0, Jz=0.5, hz=0.1, nup=2)
    assert m.N == 15 and all(bin(int(s)).count("1") == 2 for s in m.states)
    assert m.hopping_list[0] == (1, 2, 1.0) and m.zz_list[-1] == (5, 6, 0.5) and np.allclose(m.onsite_field, 0.1)
    mp = O.XXZChain(6, nup=3, boundary="periodic")
    assert mp.hopping_list[-1] == (6, 1, 0.5) and len(mp.hopping_list) == 6
    assert len(O.XXZChain(2, nup=1, boundary="periodic").hopping_list) == 1   # periodic bond only when L > 2


@pytest.mark.parametrize("L,nup", [(4, 2), (6, 3), (7, 2), (9, 4), (10, 5), (5, 0), (5, 5)])
def test_basis_order_is_lex_combinations(O, L, nup):
    # F1: order of Combinatorics.combinations(1:L, nup) (src/Basis.jl:41) == itertools.combinations
    want = [sum(1 << (i - 1) for i in c) for c in itertools.combinations(range(1, L + 1), nup)]
    m = O.XXZChain(L, nup=nup)
    assert list(map(int, m.states)) == want
    assert [m.lookup(s) for s in want] == list(range(1, len(want) + 1))
    assert m.lookup(1 << L) == 0


@pytest.mark.parametrize("L,nup,bc", [(6, 3, "open"), (8, 3, "periodic"), (9, 4, "open"), (6, None, "open"), (5, None, "periodic")])
def test_apply_vs_independent_dense(O, D, L, nup, bc):
    m = O.XXZChain(L, Jxy=1.3, Jz=0.7, hz=0.2, nup=nup, boundary=bc)
    H = D.dense_H(L, nup, *D.xxz_lists(L, 1.3, 0.7, 0.2, bc))
    rng = np.random.default_rng(7)
    psi = rng.standard_normal(m.N) + 1j * rng.standard_normal(m.N)
    assert np.abs(O.apply_H(m, psi) - H @ psi).max() <= 1e-13
    assert np.abs(O.apply_H(m, psi.real) - H @ psi.real).max() <= 1e-13
    assert np.abs(O.apply_rescaled_H(m, psi, 3.0, -0.4) - (H @ psi + 0.4 * psi) / 3.0).max() <= 1e-13


def test_tridiag_eig_vs_numpy(O):
    rng = np.random.default_rng(8)
    for n in (1, 2, 5, 30, 120):
        d, e = rng.standard_normal(n), rng.standard_normal(max(n - 1, 0))
        w, z = O.symtridiag_eig(d, e)
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        assert np.abs(w - np.linalg.eigvalsh(T)).max() <= 1e-12
        assert np.abs(T @ z - z * w).max() <= 1e-12 and np.abs(z.T @ z - np.eye(n)).max() <= 1e-12


def test_chebyshev_coeffs_vs_scipy(O):
    from scipy.special import jv
    a, b, dt, n = 2.7, -0.3, 0.8, 40
    k = np.arange(n)
    want = np.where(k == 0, 1.0, 2.0) * (-1j) ** k * jv(k, a * dt) * np.exp(-1j * b * dt)
    assert np.abs(O.chebyshev_coeffs(n, a, b, dt) - want).max() <= 1e-15


def test_evolution_vs_dense_expm(O, D):
    L, nup = 8, 4
    m = O.XXZChain(L, Jz=0.8, nup=nup)
    H = D.dense_H(L, nup, *D.xxz_lists(L, 1.0, 0.8, 0.0))
    rng = np.random.default_rng(9)
    psi0 = rng.standard_normal(m.N) + 1j * rng.standard_normal(m.N)
    psi0 /= np.linalg.norm(psi0)
    want = D.expm_herm(H, 0.7) @ psi0
    assert np.abs(O.krylov_time_evolve(m, psi0, 0.7, kry_m=30) - want).max() <= 1e-10
    w = np.linalg.eigvalsh(H)
    assert np.abs(O.chebyshev_time_evolve(m, psi0, 0.7, cheb_n=60, Ebounds=(w[0], w[-1])) - want).max() <= 1e-10
    # a real psi0 goes through the same path (Krylov.jl:136 T<:Number)
    pr = psi0.real / np.linalg.norm(psi0.real)
    assert np.abs(O.krylov_time_evolve(m, pr, 0.7, kry_m=30) - D.expm_herm(H, 0.7) @ pr).max() <= 1e-10
