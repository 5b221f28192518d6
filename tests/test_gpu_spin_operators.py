"""create_spin_operator (src/Hamiltonian.jl:49-136) through sd_spin_operator: the reference's own assertions
(test/test_Hamiltonian.jl:27-44, 62-92, 95-113) plus random-vector parity with the scatter-form oracle.  Each result element
is one product (z, x, y) or a copy (plus, minus), so the comparison is bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_reference_known_answers_full_basis(pkg):
    L = 4
    m = pkg.build_model(L, hopping=[(1, 2, 1.0)], zz=[(1, 2, 0.5)])
    psi = np.zeros(1 << L, complex)
    psi[0] = 1.0
    assert pkg.create_spin_operator(1, "z")(psi, m)[0] == -0.5 + 0j
    assert pkg.create_spin_operator(1, "plus")(psi, m)[1] == 1.0 + 0j
    assert pkg.create_spin_operator(1, ":x")(psi, m)[1] == 0.5 + 0j
    assert pkg.create_spin_operator(1, "y")(psi, m)[1] == -0.5j


@pytest.mark.parametrize("L", [1, 5, 9])
@pytest.mark.parametrize("op", ["z", "plus", "minus", "x", "y"])
def test_full_basis_matches_oracle(pkg, O, L, op):
    m = pkg.build_model(L, hopping=[(i, i + 1, 0.5) for i in range(1, L)])
    r = O.build_model(L, hopping=[(i, i + 1, 0.5) for i in range(1, L)])
    rng = np.random.default_rng(L)
    for site in sorted({1, (L + 1) // 2, L}):
        for cplx in (True, False):
            if op == "y" and not cplx:
                continue
            psi = rng.standard_normal(m.N) + (1j * rng.standard_normal(m.N) if cplx else 0)
            got = pkg.create_spin_operator(site, op)(psi, m)
            assert got.dtype == psi.dtype
            assert np.array_equal(got, O.spin_operator(r, site, op, psi))


@pytest.mark.parametrize("L,nup", [(4, 2), (13, 6), (16, 8), (40, 2)])
def test_sz_in_sector_matches_oracle(pkg, O, L, nup):
    m = pkg.XXZChain(L, nup=nup)
    r = O.XXZChain(L, nup=nup)
    rng = np.random.default_rng(L)
    psi = rng.standard_normal(m.N) + 1j * rng.standard_normal(m.N)
    for site in (1, L // 2, L):
        assert np.array_equal(pkg.create_spin_operator(site, "z")(psi, m), O.spin_operator(r, site, "z", psi))


def test_validation(pkg):
    m = pkg.XXZChain(4, Jxy=1.0, Jz=1.0, nup=2)
    psi = np.zeros(m.N)
    with pytest.raises(pkg.ArgumentError):
        pkg.create_spin_operator(0, "z")
    with pytest.raises(pkg.ArgumentError):
        pkg.create_spin_operator(1, "foo")
    with pytest.raises(pkg.ArgumentError):
        pkg.create_spin_operator(5, "z")(psi, m)
    with pytest.raises(pkg.DimensionMismatch):
        pkg.create_spin_operator(1, "z")(np.zeros(3), m)
    psi[0] = 1.0
    assert abs(np.linalg.norm(pkg.create_spin_operator(1, "z")(psi, m)) - 0.5) < 1e-15
    for op in ("plus", "minus", "x", "y"):
        with pytest.raises(pkg.ArgumentError):
            pkg.create_spin_operator(1, op)(psi.astype(complex), m)
    full = pkg.build_model(3, hopping=[(1, 2, 1.0)])
    with pytest.raises(pkg.ArgumentError):                       # InexactError upstream
        pkg.create_spin_operator(1, "y")(np.ones(8), full)


def test_szq_is_the_sum_of_site_operators(pkg):
    m = pkg.XXZChain(6, Jxy=1.0, Jz=1.0, nup=3)
    _, psi0 = pkg.groundstate(m, lanc_m=20)
    q = np.pi / 3
    ref = np.zeros(m.N, complex)
    for r in range(1, m.L + 1):
        ref += np.exp(1j * q * (r - 1)) / np.sqrt(m.L) * pkg.create_spin_operator(r, "z")(psi0.astype(complex), m)
    assert np.abs(pkg.Sz_q_vector(m, psi0, q) - ref).max() < 1e-12
