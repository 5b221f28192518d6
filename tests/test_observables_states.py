"""§8f rows f2 (Observables) and f3 (InitialStates): oracle vs numpy and the reference's own tests on CPU; the HIP
reductions vs the oracle on the GPU."""
import numpy as np
import pytest


def dense_obs(O, m, psi):
    L = m.L
    st = m.states.astype(np.int64)
    bits = ((st[:, None] >> np.arange(L)) & 1) - 0.5
    prob = np.abs(psi) ** 2
    S = prob @ bits
    SS = np.einsum("n,ni,nj->ij", prob, bits, bits)
    Cr = np.array([sum(SS[i, (i + r) % L] - S[i] * S[(i + r) % L] for i in range(L)) / L for r in range(L)])
    return S, Cr


@pytest.mark.parametrize("L,nup", [(4, 2), (7, 3), (8, None), (10, 5)])
def test_oracle_observables_vs_numpy(O, L, nup):
    m = O.XXZChain(L, nup=nup)
    rng = np.random.default_rng(L)
    psi = rng.standard_normal(m.N) + 1j * rng.standard_normal(m.N)
    psi /= np.linalg.norm(psi)
    S, Cr = dense_obs(O, m, psi)
    assert np.abs(O.magnetization_per_site(psi, m) - S).max() <= 1e-14
    assert np.abs(O.connected_correlations(psi, m) - Cr).max() <= 1e-14
    Sq = O.structure_factor_Sq(psi, m)
    assert np.allclose(sorted(Sq), 2 * np.pi * np.arange(L) / L)
    assert np.abs(np.array([Sq[k] for k in sorted(Sq)]) - np.fft.fft(Cr).real).max() <= 1e-13


def test_reference_initial_state_tests(O):
    # test/test_InitialStates.jl:6-109 on the oracle
    mf, ms = O.XXZChain(4), O.XXZChain(4, nup=2)
    for psi, m in ((O.domain_wall_state(mf), mf), (O.domain_wall_state(ms), ms), (O.neel_state(mf), mf)):
        assert psi.dtype == np.float64 and len(psi) == m.N and abs((psi ** 2).sum() - 1) < 1e-15 and np.count_nonzero(psi) == 1
    assert O.neel_state(mf)[0b0101] == 1.0
    assert O.polarized_state(mf, up=True)[15] == 1.0 and O.polarized_state(mf, up=False)[0] == 1.0
    assert O.polarized_state_with_flips(mf, [1, 3])[0b1010] == 1.0
    with pytest.raises(O.OracleError):
        O.polarized_state(ms, up=True)                     # not in the nup=2 sector
    with pytest.raises(O.OracleError):
        O.polarized_state_with_flips(mf, [5])
    with pytest.raises(O.OracleError):
        O.neel_state(O.XXZChain(4, nup=1))


@pytest.mark.parametrize("L,nup", [(4, 2), (4, None), (9, 4), (12, 6), (30, 15), (36, 18)])
def test_host_initial_state_index(pkg, L, nup):
    m = pkg.XXZChain(L, nup=nup, ctx=None)
    from_bits = lambda s: int(m.rank(np.array([s], dtype=np.uint64))[0])
    ist = pkg.initial_states
    dw = (1 << (nup if nup is not None else (L + 1) // 2)) - 1
    assert ist.state_index(m, ist.DOMAIN_WALL) == from_bits(dw) == (0 if nup is not None else dw)
    neel = sum(1 << i for i in range(0, L, 2))
    if nup is None or nup == (L + 1) // 2:
        assert ist.state_index(m, ist.NEEL) == from_bits(neel)
    else:
        with pytest.raises(pkg.ArgumentError):
            ist.state_index(m, ist.NEEL)
    if nup is None:
        assert ist.state_index(m, ist.POLARIZED_UP) == (1 << L) - 1
        assert ist.state_index(m, ist.POLARIZED_FLIPS, [1, L]) == ((1 << L) - 1) ^ 1 ^ (1 << (L - 1))
    else:
        with pytest.raises(pkg.ArgumentError):
            ist.state_index(m, ist.POLARIZED_UP)
        flips = list(range(nup + 1, L + 1))                # flip the last L-nup sites down -> domain wall
        assert ist.state_index(m, ist.POLARIZED_FLIPS, flips) == 0
    with pytest.raises(pkg.ArgumentError):
        ist.state_index(m, ist.POLARIZED_FLIPS, [0])
    if L <= 12:
        assert np.count_nonzero(pkg.domain_wall_state(m)) == 1 and pkg.domain_wall_state(m).dtype == np.float64


@pytest.mark.gpu
@pytest.mark.parametrize("L,nup,bc", [(4, 2, "open"), (9, 4, "open"), (12, 6, "periodic"), (16, 8, "open"), (20, 7, "open"), (9, None, "open"), (40, 2, "open"),
                                      # one-pass kernels (k_obs2): odd L, full basis with 2^10-row tiles, L > 32 (64-bit rotations)
                                      (13, 6, "open"), (17, 5, "periodic"), (12, None, "open"), (15, None, "periodic"), (34, 3, "open")])
def test_hip_observables_vs_oracle(pkg, O, L, nup, bc):
    m = pkg.XXZChain(L, nup=nup, boundary=bc)
    r = O.XXZChain(L, nup=nup, boundary=bc)
    rng = np.random.default_rng(L)
    for cplx in (True, False):
        psi = rng.standard_normal(m.N) + (1j * rng.standard_normal(m.N) if cplx else 0)
        psi /= np.linalg.norm(psi)
        # tolerance: sums of N non-negative terms in a different (tree) order, plus R_r instead of the L x L matrix
        assert np.abs(pkg.magnetization_per_site(psi, m) - O.magnetization_per_site(psi, r)).max() <= 1e-13
        assert np.abs(pkg.connected_correlations(psi, m) - O.connected_correlations(psi, r)).max() <= 1e-13
        a, b = pkg.structure_factor_Sq(psi, m), O.structure_factor_Sq(psi, r)
        assert sorted(a) == sorted(b)
        assert max(abs(a[k] - b[k]) for k in a) <= 1e-12
        assert pkg.structure_factor(m, psi) == a               # test/test_PublicAPI.jl:136-151


@pytest.mark.gpu
def test_observables_on_device_after_evolution(pkg, O):
    """examples/example_time_evolution.jl pattern: domain wall -> time_evolve -> magnetization profile, on the GPU."""
    import torch
    L, nup = 14, 7
    m = pkg.XXZChain(L, nup=nup)
    r = O.XXZChain(L, nup=nup)
    psi0 = pkg.domain_wall_state(m)
    assert np.array_equal(psi0, O.domain_wall_state(r))
    psit = pkg.time_evolve(m, psi0.astype(complex), 1.5, method="chebyshev", cheb_n=60, Ebounds=(-7.0, 7.0))
    mags = pkg.magnetization_per_site(torch.from_numpy(psit).cuda(), m)
    assert np.abs(mags - O.magnetization_per_site(psit, r)).max() <= 1e-13
    assert abs(mags.sum()) <= 1e-12                         # total S^z = 0 is conserved
    d = pkg.domain_wall_state(m, device="cuda")
    assert d.is_cuda and float(d.sum()) == 1.0 and float(d[0]) == 1.0


def test_create_spin_operator_argument_checks_need_no_device(pkg):
    """src/Hamiltonian.jl:50-55: site and operator name are validated when the closure is created."""
    with pytest.raises(pkg.ArgumentError):
        pkg.create_spin_operator(0, "z")
    with pytest.raises(pkg.ArgumentError):
        pkg.create_spin_operator(1, "foo")
    assert callable(pkg.create_spin_operator(3, ":plus"))


def test_oracle_spin_operator_known_answers(O):
    """test/test_Hamiltonian.jl:27-44 on the oracle's scatter form."""
    r = O.build_model(4, hopping=[(1, 2, 1.0)])
    psi = np.zeros(16, complex)
    psi[0] = 1.0
    assert O.spin_operator(r, 1, "z", psi)[0] == -0.5
    assert O.spin_operator(r, 1, "plus", psi)[1] == 1.0
    assert O.spin_operator(r, 1, "x", psi)[1] == 0.5
    assert O.spin_operator(r, 1, "y", psi)[1] == -0.5j
    assert np.all(O.spin_operator(r, 1, "minus", psi) == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("L,nup", [(14, 7), (19, 9), (13, None), (34, 3)])
def test_one_pass_observables_equal_the_chunked_kernels(pkg, L, nup, monkeypatch):
    """k_obs2 (all sites / all lags in one pass over psi, uniform sites of a tile added once per tile, lags r and L-r shared)
    against the 16-accumulators-per-pass kernels it replaces (SD_OBS_CHUNKED=1): same sums in another order, 1e-13."""
    m = pkg.XXZChain(L, nup=nup)
    rng = np.random.default_rng(L)
    for cplx in (True, False):
        psi = rng.standard_normal(m.N) + (1j * rng.standard_normal(m.N) if cplx else 0)
        psi /= np.linalg.norm(psi)
        monkeypatch.delenv("SD_OBS_CHUNKED", raising=False)
        mag, cr = pkg.magnetization_per_site(psi, m), pkg.connected_correlations(psi, m)
        monkeypatch.setenv("SD_OBS_CHUNKED", "1")
        mag2, cr2 = pkg.magnetization_per_site(psi, m), pkg.connected_correlations(psi, m)
        assert np.abs(mag - mag2).max() <= 1e-13 and np.abs(cr - cr2).max() <= 1e-13
