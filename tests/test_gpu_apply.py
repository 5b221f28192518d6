"""GPU parity: the HIP apply (through the C ABI) against the CPU oracle on the same seeded inputs.
Plain applies follow the reference's per-row operation order without FMA contraction, so they are
compared BIT-EXACT (np.array_equal); the tolerance-based comparisons state their tolerance."""
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
    # (L, nup, Jxy, Jz, hz, boundary)
    (2, 1, 1.0, 1.0, 0.0, "open"),
    (4, 2, 1.0, 1.0, 0.0, "open"),
    (6, 3, 1.0, 1.0, 0.0, "open"),
    (9, 0, 1.0, 1.0, 0.0, "open"),
    (9, 9, 1.0, 1.0, 0.0, "open"),
    (9, 1, 1.0, 1.0, 0.0, "open"),
    (12, 6, 1.0, 1.0, 0.0, "open"),
    (12, 5, 1.3, 0.7, 0.2, "open"),
    (12, 6, 1.0, 1.0, 0.0, "periodic"),
    (13, 4, 0.37, -1.1, 0.05, "periodic"),
    (16, 8, 1.0, 1.0, 0.0, "open"),
    (17, 8, 1.0, 0.3, 0.0, "open"),
    (18, 9, 1.0, 1.0, 0.0, "open"),
    (20, 10, 1.0, 1.0, 0.0, "open"),
    (20, 3, 0.9, 1.7, 0.3, "open"),
    (21, 10, 1.0, 1.0, 0.0, "periodic"),
    (10, None, 1.0, 1.0, 0.0, "open"),
    (11, None, 0.8, 1.2, 0.1, "periodic"),
    # full 2^L basis, L >= 12: k_apply_fulltile (tiles of 2^10 consecutive rows)
    (12, None, 1.0, 1.0, 0.0, "open"),
    (13, None, 0.8, 1.2, 0.1, "periodic"),
    (15, None, 1.0, 0.4, -0.3, "open"),
    (19, None, 1.0, 1.0, 0.0, "open"),        # >= 512 tiles: XCD-chunked tile order
]


@pytest.mark.parametrize("L,nup,Jxy,Jz,hz,bc", CASES)
def test_apply_bit_exact_vs_oracle(pkg, O, L, nup, Jxy, Jz, hz, bc):
    m = pkg.XXZChain(L, Jxy=Jxy, Jz=Jz, hz=hz, nup=nup, boundary=bc)
    r = O.XXZChain(L, Jxy=Jxy, Jz=Jz, hz=hz, nup=nup, boundary=bc)
    assert m.N == r.N
    for cplx in (True, False):
        psi = rand_vec(m.N, 100 + L, cplx)
        out = np.empty_like(psi)
        pkg.apply_H(out, psi, m)
        want = O.apply_H(r, psi)
        assert np.array_equal(out, want), f"max diff {np.abs(out - want).max()}"


@pytest.mark.parametrize("L,nup", [(8, 4), (12, 6), (16, 7), (18, 9)])
def test_basis_indices_bit_exact(pkg, O, L, nup):
    m = pkg.XXZChain(L, nup=nup)
    r = O.XXZChain(L, nup=nup)
    st = r.states
    assert np.array_equal(m.states, st)
    assert np.array_equal(m.rank(st), np.arange(m.N))


def test_rescaled_bit_exact(pkg, O):
    m = pkg.XXZChain(14, nup=7, Jz=0.6)
    r = O.XXZChain(14, nup=7, Jz=0.6)
    psi = rand_vec(m.N, 3)
    out = np.empty_like(psi)
    pkg.apply_rescaled_H(out, psi, pkg.apply_H, m, 4.3, -0.7)
    assert np.array_equal(out, O.apply_rescaled_H(r, psi, 4.3, -0.7))


def test_long_range_and_general_bonds(pkg, O):
    L, nup = 10, 4
    hop = pkg.long_range_hopping(L, lambda i, j: 1.0 / (j - i) ** 2)
    zz = [(i, j, 0.3 / (j - i)) for i in range(1, L + 1) for j in range(i + 1, L + 1)]
    f = np.linspace(-0.5, 0.5, L)
    m = pkg.build_model(L, nup=nup, hopping=hop, onsite_field=f, zz=zz)
    r = O.build_model(L, nup=nup, hopping=hop, onsite_field=f, zz=zz)
    psi = rand_vec(m.N, 5)
    out = np.empty_like(psi)
    pkg.apply_H(out, psi, m)
    assert np.array_equal(out, O.apply_H(r, psi))


def test_szq_vs_oracle(pkg, O):
    # tolerance: phases come from the host libm on both sides; the device multiplies in the same order -> 1e-15 abs
    # full basis with L >= 12: k_szq_full (first k terms of the site sum once per thread, one add per remaining term)
    for (L, nup) in [(6, 3), (12, 6), (16, 8), (9, None), (12, None), (15, None), (17, None)]:
        m = pkg.XXZChain(L, nup=nup)
        r = O.XXZChain(L, nup=nup)
        for cplx in (True, False):
            psi = rand_vec(m.N, 11, cplx)
            for q in (0.0, np.pi / 3, np.pi):
                got = pkg.Sz_q_vector(m, psi, q)
                want = O.Sz_q_vector(r, psi, q)
                assert np.abs(got - want).max() <= 1e-15 * max(1.0, np.abs(want).max())


def test_szq_full_basis_kernel_equals_the_row_loop_bit_for_bit(pkg, monkeypatch):
    """k_szq_full shares the first k terms of the site sum among the rows with equal low index bits and adds +-(phase/2) where
    the row loop multiplies phase * (+-0.5): the same additions in the same order, so the same bits as k_szq_generic
    (SD_SZQ_FULL_GENERIC=1), for real and complex input, also when the vector is one rank's share of a sharded full basis."""
    import torch
    for L in (12, 14, 19):
        m = pkg.XXZChain(L)
        for cplx in (True, False):
            psi = rand_vec(m.N, 40 + L, cplx)
            for q in (0.3, 2 * np.pi * 5 / L):
                monkeypatch.delenv("SD_SZQ_FULL_GENERIC", raising=False)
                a = pkg.Sz_q_vector(m, psi, q)
                monkeypatch.setenv("SD_SZQ_FULL_GENERIC", "1")
                b = pkg.Sz_q_vector(m, psi, q)
                assert np.array_equal(a, b)
    monkeypatch.delenv("SD_SZQ_FULL_GENERIC", raising=False)
    L, P = 14, 4
    full = pkg.XXZChain(L)
    psi = rand_vec(full.N, 9, True)
    want = pkg.Sz_q_vector(full, psi, 1.1)
    for r in range(P):
        m = pkg.XXZChain(L)
        op = pkg.ShardedOperator(m, r, P, mode="range")
        rows = m.local_rows()
        got = op.Sz_q_vector(torch.from_numpy(psi[rows].copy()).cuda(), 1.1).cpu().numpy()
        assert np.array_equal(got, want[rows])


def test_dimension_and_argument_errors(pkg):
    m = pkg.XXZChain(6, nup=3)
    with pytest.raises(pkg.DimensionMismatch):
        pkg.apply_H(np.zeros(5), np.zeros(5), m)
    with pytest.raises(pkg.ArgumentError):
        pkg.XXZChain(6, nup=7)
    with pytest.raises(pkg.ArgumentError):
        pkg.XXZChain(64, nup=1)
    with pytest.raises(pkg.ArgumentError):
        pkg.XXZChain(6, nup=3, boundary="twisted")


@pytest.mark.parametrize("ls", ["11", "12", "13"])
def test_tile_length_classes_bit_exact(pkg, O, ls, monkeypatch):
    """Plans with >= 32768 tiles launch one kernel per tile length class (64/128/256-thread workgroups).  SD_LEN_CLASSES=2
    forces the split on plans small enough for the oracle: plain, rescaled and the reduction epilogues (KPM moments)."""
    monkeypatch.setenv("SD_LEN_CLASSES", "2")
    monkeypatch.setenv("SD_SUFFIX_BITS", ls)
    for (L, nup, bc) in [(16, 8, "open"), (19, 9, "periodic"), (20, 7, "open")]:
        m = pkg.XXZChain(L, nup=nup, Jz=0.7, hz=0.1, boundary=bc)
        r = O.XXZChain(L, nup=nup, Jz=0.7, hz=0.1, boundary=bc)
        for cplx in (True, False):
            psi = rand_vec(m.N, 31 + L, cplx)
            out = np.empty_like(psi)
            pkg.apply_H(out, psi, m)
            assert np.array_equal(out, O.apply_H(r, psi))
        a, b = 7.5, 0.25
        pkg.apply_rescaled_H(out, psi, pkg.apply_H, m, a, b)
        assert np.array_equal(out, O.apply_rescaled_H(r, psi, a, b))
        phi = rand_vec(m.N, 5)
        phi /= np.linalg.norm(phi)
        mu = pkg.compute_chebyshev_moments(pkg.apply_H, phi, 12, a, b, m)
        want = O.compute_chebyshev_moments(r, phi, 12, a, b)
        assert np.abs(mu - want).max() <= 1e-12            # fixed-order partial sums differ from the oracle's serial sum


def test_full_basis_tiled_path(pkg, O):
    """nup = nothing, L >= 12: general bond lists (long-range hops, fields, zz) and the fused epilogues on the full-basis
    tiled kernel; leading chain bonds take the LDS / stream path, the rest are gathers -- all in list order, bit-exact."""
    L = 13
    assert pkg.XXZChain(L).device_path == "full-tiled"
    chain = [(i, i + 1, 0.5) for i in range(1, L)]
    extra = [(1, L, 0.25), (2, 7, -0.3), (3, 12, 0.11)]
    zz = [(i, i + 1, 0.9) for i in range(1, L)] + [(1, L, 0.4)]
    f = np.linspace(-0.2, 0.3, L)
    for hop in (chain + extra, extra + chain, chain):
        m = pkg.build_model(L, hopping=hop, onsite_field=f, zz=zz)
        r = O.build_model(L, hopping=hop, onsite_field=f, zz=zz)
        for cplx in (True, False):
            psi = rand_vec(m.N, 9, cplx)
            out = np.empty_like(psi)
            pkg.apply_H(out, psi, m)
            assert np.array_equal(out, O.apply_H(r, psi))
        pkg.apply_rescaled_H(out, psi, pkg.apply_H, m, 6.0, -0.5)
        assert np.array_equal(out, O.apply_rescaled_H(r, psi, 6.0, -0.5))
    phi = rand_vec(m.N, 5)
    phi /= np.linalg.norm(phi)
    mu = pkg.compute_chebyshev_moments(pkg.apply_H, phi, 10, 6.0, -0.5, m)
    assert np.abs(mu - O.compute_chebyshev_moments(r, phi, 10, 6.0, -0.5)).max() <= 1e-12
    E0, psi0 = pkg.groundstate(m, lanc_m=80)                # DOT epilogue inside the Lanczos recursion
    want = O.apply_H(r, psi0)
    assert np.linalg.norm(want - E0 * psi0) < 1e-8          # a converged eigenpair of the oracle's operator


@pytest.mark.parametrize("cache", ["0", "1"])
def test_diagonal_cache_on_off_bit_exact(pkg, O, cache, monkeypatch):
    """Couplings without an exact shortcut for the diagonal: the list-order sum is evaluated once per model and cached
    (default) or evaluated inside every apply (SD_DIAG_CACHE=0).  Both are the reference's sequential sum, bit for bit; short
    tiles (SD_SUFFIX_BITS=8) so that several length classes and many tiles are in play."""
    monkeypatch.setenv("SD_DIAG_CACHE", cache)
    monkeypatch.setenv("SD_SUFFIX_BITS", "8")
    for (L, nup, kw) in [(18, 9, dict(Jz=0.7)), (20, 8, dict(Jz=0.7, hz=0.3)), (19, 10, dict(Jxy=0.9, Jz=0.7, hz=0.3)),
                         (18, 9, dict(Jz=0.7, hz=0.3, boundary="periodic"))]:
        m = pkg.XXZChain(L, nup=nup, **kw)
        r = O.XXZChain(L, nup=nup, **kw)
        for cplx in (True, False):
            psi = rand_vec(m.N, 77 + L, cplx)
            out = np.empty_like(psi)
            pkg.apply_H(out, psi, m)
            assert np.array_equal(out, O.apply_H(r, psi))
        psi = rand_vec(m.N, 79 + L)
        a, b = 5.5, 0.25
        out = np.empty_like(psi)
        pkg.apply_rescaled_H(out, psi, pkg.apply_H, m, a, b)
        assert np.array_equal(out, O.apply_rescaled_H(r, psi, a, b))


def _bond_lists(L, kind):
    """hopping / zz lists of build_model (src/SpinModel.jl:23-46) beyond the chain"""
    hop, zz = [], []
    if kind == "all-pairs":            # long_range_hopping: every pair, J(i, j) = 1 / (j - i)^2, in the reference's order
        for i in range(1, L + 1):
            for j in range(i + 1, L + 1):
                hop.append((i, j, 0.5 / (j - i) ** 2)); zz.append((i, j, 1.0 / (j - i) ** 2))
        return hop, zz
    ranges = {"j1j2": ((1, 1.0), (2, 0.5)), "j1j2j3": ((1, 1.0), (2, 0.5), (3, -0.3)), "j1j2-periodic": ((1, 1.0), (2, 0.37))}[kind]
    for d, J in ranges:
        for i in range(1, L + 1):
            j = i + d
            if j > L:
                if "periodic" not in kind:
                    continue
                j -= L
            hop.append((i, j, 0.5 * J)); zz.append((i, j, J))
    return hop, zz


@pytest.mark.parametrize("L,nup,kind,ls", [(16, 8, "j1j2", 12), (20, 10, "j1j2", 12), (20, 9, "j1j2j3", 10), (19, 9, "j1j2-periodic", 12),
                                            (18, 9, "all-pairs", 12),      # 153 bonds: three blocks of 64, 55 suffix-suffix bonds in 5 table chunks
                                            (20, 10, "all-pairs", 8), (14, 7, "all-pairs", 12), (22, 11, "j1j2-periodic", 12)])
def test_models_with_longer_bonds_bit_exact_vs_oracle(pkg, O, L, nup, kind, ls, monkeypatch):
    """Second / third neighbours and all-pairs lists through the general-bond plan of k_apply_tiled (prefix-prefix streams,
    suffix-suffix partners from the packed table, mixed bonds through the second LDS image): bit-exact against the oracle, and equal
    to the per-row form the plan replaces (SD_GEN_PLAN=0)."""
    monkeypatch.setenv("SD_SUFFIX_BITS", str(ls))
    hop, zz = _bond_lists(L, kind)
    m = pkg.build_model(L, nup=nup, hopping=hop, zz=zz)
    r = O.build_model(L, nup=nup, hopping=hop, zz=zz)
    monkeypatch.setenv("SD_GEN_PLAN", "0")
    m0 = pkg.build_model(L, nup=nup, hopping=hop, zz=zz)
    for cplx in (True, False):
        psi = rand_vec(m.N, 7 + L, cplx)
        out, out0 = np.empty_like(psi), np.empty_like(psi)
        pkg.apply_H(out, psi, m)
        pkg.apply_H(out0, psi, m0)
        want = O.apply_H(r, psi)
        assert np.array_equal(out, want), f"max diff {np.abs(out - want).max()}"
        assert np.array_equal(out0, want)
