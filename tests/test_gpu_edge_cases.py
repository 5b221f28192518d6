"""Edge cases of the boundary on the GPU: tiny and degenerate sectors, very long chains with few up spins (generic
combinadic path, 64-bit states), aliasing / null / dtype errors, torch device tensors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("L,nup", [(1, 0), (1, 1), (1, None), (2, 0), (2, 2), (2, None), (3, 1), (40, 2), (63, 1), (50, 3), (33, 0)])
def test_small_and_long_chains_bit_exact(pkg, O, L, nup):
    m = pkg.XXZChain(L, Jxy=0.9, Jz=1.1, hz=0.3, nup=nup)
    r = O.XXZChain(L, Jxy=0.9, Jz=1.1, hz=0.3, nup=nup)
    assert m.N == r.N and np.array_equal(m.states, r.states)
    rng = np.random.default_rng(L)
    for cplx in (True, False):
        psi = rng.standard_normal(m.N) + (1j * rng.standard_normal(m.N) if cplx else 0)
        out = np.empty_like(psi)
        pkg.apply_H(out, psi, m)
        assert np.array_equal(out, O.apply_H(r, psi))
    phi = pkg.Sz_q_vector(m, psi, 0.7)
    assert np.abs(phi - O.Sz_q_vector(r, psi, 0.7)).max() <= 1e-15 * max(1.0, np.abs(phi).max())


def test_device_paths_reported(pkg):
    assert pkg.XXZChain(12, nup=6).device_path == "tiled"
    assert pkg.XXZChain(8, nup=None).device_path == "generic"
    assert pkg.XXZChain(14, nup=None).device_path == "full-tiled"   # 2^10-row tiles of the full basis
    assert pkg.XXZChain(50, nup=3).device_path == "generic"     # 2^38 prefix tiles would not fit a table
    assert pkg.XXZChain(30, nup=6).device_path == "generic"     # small and very dilute (12 rows per tile, 0.6 M rows): one launch of the per-row kernel
    assert pkg.XXZChain(30, nup=24).device_path == "generic"    # ... the same sector seen from the other side
    assert pkg.XXZChain(32, nup=8).device_path == "tiled"       # dilute but large: tiles, the short ones through k_apply_short


def test_boundary_errors(pkg):
    m = pkg.XXZChain(8, nup=4)
    psi = np.zeros(m.N)
    with pytest.raises(pkg.ArgumentError):
        pkg.apply_H(psi, psi, m)                               # out must not alias psi
    with pytest.raises(pkg.ArgumentError):
        pkg.apply_H(np.zeros(m.N, np.float32), np.zeros(m.N, np.float32), m)
    with pytest.raises(pkg.ArgumentError):
        pkg.apply_H(np.zeros(m.N), np.zeros(m.N, complex), m)  # mixed element types
    with pytest.raises(pkg.DimensionMismatch):
        pkg.apply_H(np.zeros(m.N + 1), np.zeros(m.N + 1), m)
    with pytest.raises(pkg.DimensionMismatch):
        pkg.Sz_q_vector(m, np.zeros(m.N - 1), 0.1)
    with pytest.raises(pkg.ArgumentError):
        pkg.compute_chebyshev_moments(pkg.apply_H, np.ones(m.N, complex), 1, 1.0, 0.0, m)   # kpm_m >= 2
    with pytest.raises(pkg.ArgumentError):
        pkg.lanczos_extremal(None, m)                          # applyH must be apply_H or a callable (applyH(out, psi, model))


def test_torch_device_tensors_and_streams(pkg, O):
    import torch
    m = pkg.XXZChain(14, nup=7)
    r = O.XXZChain(14, nup=7)
    psi = np.random.default_rng(1).standard_normal(m.N)
    want = O.apply_H(r, psi)
    d = torch.from_numpy(psi).cuda()
    out = torch.empty_like(d)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):                                 # the launch follows torch's current stream
        pkg.apply_H(out, d, m)
    s.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    phi = pkg.Sz_q_vector(m, d, 1.1)
    assert phi.is_cuda and np.abs(phi.cpu().numpy() - O.Sz_q_vector(r, psi, 1.1)).max() <= 1e-15


def test_kpm_sum_rule_medium_size(pkg):
    """Size-independent property (test/test_KPM.jl:67-91 at L=20): integral of S(q,w) over the whole band equals
    |S^z_q psi0|^2 for ANY psi0 when (a,b) cover the spectrum; Jackson kernel, M=160: 1 % (kernel resolution)."""
    L = 20
    m = pkg.XXZChain(L, nup=L // 2)
    rng = np.random.default_rng(5)
    psi0 = rng.standard_normal(m.N)
    psi0 /= np.linalg.norm(psi0)
    lo, hi = pkg.estimate_energy_bounds(pkg.apply_H, m, lanc_m=60, seed=1)
    a, b = pkg.rescaling_from_bounds(lo, hi)
    Hpsi = np.empty_like(psi0)
    pkg.apply_H(Hpsi, psi0, m)
    E0 = float(psi0 @ Hpsi)
    omega = np.arange(lo - E0 - 0.5, hi - E0 + 0.5, 0.01)
    q = np.array([np.pi / 2, np.pi])
    S = pkg.kpm_sqw(psi0, m, q, omega, a=a, b=b, kpm_m=160)
    for iq in range(2):
        w = np.linalg.norm(pkg.Sz_q_vector(m, psi0, float(q[iq]))) ** 2
        assert abs(S[iq].sum() * 0.01 - w) <= 1e-2 * w


def test_host_pointer_applies_reuse_and_release_staging(pkg, O):
    """sd_apply keeps its two device staging vectors in the context between calls (grow-only); sizes going up and down and
    an explicit release must not change results."""
    ctx = pkg.XXZChain(4, nup=2).ctx
    for (L, nup) in [(12, 6), (16, 8), (10, 5), (16, 7)]:
        m = pkg.XXZChain(L, nup=nup, Jz=0.4)
        r = O.XXZChain(L, nup=nup, Jz=0.4)
        psi = np.random.default_rng(L + nup).standard_normal(m.N) + 0j
        out = np.empty_like(psi)
        for _ in range(2):
            pkg.apply_H(out, psi, m)
            assert np.array_equal(out, O.apply_H(r, psi))
        if L == 16 and nup == 8:
            ctx.release_scratch()


def test_recursion_work_vectors_are_pooled_and_released(pkg, monkeypatch):
    """The recursion-level calls take their device vectors from the context's pool (blocks >= 64 MiB are kept between
    calls): repeated calls, calls of other sizes and an explicit release in between must give identical results."""
    L, nup = 24, 12                      # 2.7 M rows: 43 MB complex vectors (below the pooling threshold) ...
    m = pkg.XXZChain(L, nup=nup)
    psi0 = np.random.default_rng(3).standard_normal(m.N) + 0j
    psi0 /= np.linalg.norm(psi0)
    a = pkg.time_evolve(m, psi0, 0.2, method="chebyshev", cheb_n=12, Ebounds=(-11.0, 6.5))
    big = pkg.XXZChain(26, nup=13)        # ... and 10.4 M rows: 166 MB vectors, pooled
    phi = np.random.default_rng(4).standard_normal(big.N) + 0j
    phi /= np.linalg.norm(phi)
    r1 = pkg.time_evolve(big, phi, 0.2, method="chebyshev", cheb_n=6, Ebounds=(-12.0, 7.0))
    r2 = pkg.time_evolve(big, phi, 0.2, method="chebyshev", cheb_n=6, Ebounds=(-12.0, 7.0))      # pooled vectors reused
    k1 = pkg.time_evolve(big, phi, 0.2, method="krylov", kry_m=5)
    m.ctx.release_scratch()
    r3 = pkg.time_evolve(big, phi, 0.2, method="chebyshev", cheb_n=6, Ebounds=(-12.0, 7.0))
    k2 = pkg.time_evolve(big, phi, 0.2, method="krylov", kry_m=5)
    b = pkg.time_evolve(m, psi0, 0.2, method="chebyshev", cheb_n=12, Ebounds=(-11.0, 6.5))
    assert np.array_equal(r1, r2) and np.array_equal(r1, r3) and np.array_equal(k1, k2) and np.array_equal(a, b)
    assert abs(np.linalg.norm(r1) - 1) < 1e-3          # six Chebyshev terms: truncation, not rounding


@pytest.mark.parametrize("ls", ["13", "14", "15"])
def test_large_suffix_tiles(pkg, O, ls, monkeypatch):
    """SD_SUFFIX_BITS up to 15: 512- and 1024-thread workgroups; LS is lowered on the host when the longest tile (C(15,7) =
    6435 rows) would not fit the largest workgroup (4096 rows)."""
    monkeypatch.setenv("SD_SUFFIX_BITS", ls)
    for (L, nup) in [(16, 8), (18, 2), (17, 8)]:
        m = pkg.XXZChain(L, nup=nup, Jz=0.3)
        r = O.XXZChain(L, nup=nup, Jz=0.3)
        psi = np.random.default_rng(L).standard_normal(m.N) + 1j * np.random.default_rng(L + 1).standard_normal(m.N)
        out = np.empty_like(psi)
        pkg.apply_H(out, psi, m)
        assert np.array_equal(out, O.apply_H(r, psi))


def test_degenerate_bonds_i_equals_j_are_accepted_as_the_reference_accepts_them(pkg, O):
    """build_model takes any (i, j, J) tuple (src/SpinModel.jl:23-38).  A hop with i == j never fires (bit_i != bit_j is
    false, src/Hamiltonian.jl:248-252) and a zz term with i == j adds J/4 to every diagonal element: both must be accepted
    and reproduce the oracle bit for bit, in a sector and in the full basis."""
    hop = [(1, 2, 0.5), (3, 3, 0.7), (2, 3, 0.5), (3, 4, 0.5), (5, 5, -1.3), (4, 5, 0.5), (5, 6, 0.25)]
    zz = [(1, 2, 1.0), (2, 2, 0.5), (2, 3, 1.0), (3, 4, 1.0), (4, 5, 1.0), (5, 6, 1.0)]
    for nup in (3, None):
        m = pkg.build_model(6, nup=nup, hopping=hop, zz=zz)
        r = O.build_model(6, nup=nup, hopping=hop, zz=zz)
        rng = np.random.default_rng(4)
        psi = rng.standard_normal(m.N) + 1j * rng.standard_normal(m.N)
        out = np.empty_like(psi)
        pkg.apply_H(out, psi, m)
        assert np.array_equal(out, O.apply_H(r, psi))


@pytest.mark.parametrize("cplx", [True, False])
def test_host_transfer_modes_move_the_same_bytes(pkg, monkeypatch, cplx):
    """csrc/xfer.cpp: the host-pointer entries copy large vectors through a ring of pinned chunks filled / drained by a team of
    host threads (staged), or from registered caller pages (register), or with one plain hipMemcpy.  Every mode must hand
    the kernel, and the caller, the same bytes: 11 chunks through a ring of 4 (wrap-around), a ragged last chunk, one thread and
    many, fresh and reused result arrays."""
    L, nup = 22, 11
    m = pkg.XXZChain(L, nup=nup)
    rng = np.random.default_rng(5)
    psi = rng.standard_normal(m.N) + (1j * rng.standard_normal(m.N) if cplx else 0.0)
    if not cplx:
        psi = np.ascontiguousarray(psi.real)
    monkeypatch.setenv("SD_XFER", "plain")
    want = np.empty_like(psi)
    pkg.apply_H(want, psi, m)
    monkeypatch.setenv("SD_XFER_MIN_MB", "1")
    for mode, extra in (("staged", {"SD_XFER_CHUNK_MB": "1"}), ("staged", {"SD_XFER_CHUNK_MB": "3"}), ("register", {}), ("auto", {})):
        monkeypatch.setenv("SD_XFER", mode)
        for k, v in extra.items():
            monkeypatch.setenv(k, v)
        for _ in range(2):
            out = np.empty_like(psi)                  # fresh pages: first touched by the copy team
            pkg.apply_H(out, psi, m)
            assert np.array_equal(out, want), mode
        pkg.apply_H(out, psi, m)                      # reused pages
        assert np.array_equal(out, want), mode
    # the recursion-level entries use the same path: psi0 in, psi(t) out (Chebyshev takes ComplexF64 only, as the reference)
    psic = psi.astype(np.complex128)
    monkeypatch.setenv("SD_XFER", "plain")
    a = pkg.time_evolve(m, psic, 0.1, method="chebyshev", cheb_n=6, Ebounds=(-12.0, 7.0))
    monkeypatch.setenv("SD_XFER", "staged")
    b = pkg.time_evolve(m, psic, 0.1, method="chebyshev", cheb_n=6, Ebounds=(-12.0, 7.0))
    assert np.array_equal(a, b)
