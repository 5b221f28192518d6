"""GPU parity of the on-device recursions (Lanczos / Krylov / Chebyshev / KPM / Lanczos-S(q,w)) against the CPU
oracle on the same injected inputs, plus the reference's own PublicAPI known-answer tests run through the HIP path.
Reductions (dot/norm) are summed in a different order on the device, so these are tolerance-based; the tolerance
is stated at each assert."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def cvec(n, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(n) + 1j * rng.standard_normal(n)


# ---- the reference's own tests (test/test_PublicAPI.jl, test/test_Lanczos.jl, test/test_KPM.jl) on the HIP path ----

def test_reference_public_api_two_site(pkg, D):
    m = pkg.XXZChain(2, Jxy=1.0, Jz=1.0, nup=1)
    assert (m.L, m.nup, m.mode, m.N) == (2, 1, "sector", 2)
    H = np.zeros((2, 2))
    for j in range(2):
        e = np.zeros(2); e[j] = 1.0
        out = np.zeros(2)
        pkg.apply_H(out, e, m)
        H[:, j] = out
    assert np.array_equal(H, [[-0.25, 0.5], [0.5, -0.25]])                  # test_PublicAPI.jl:5-28
    E0, psi = pkg.groundstate(m, lanc_m=2, seed=3)
    assert abs(E0 + 0.75) <= 1e-12 and abs(np.linalg.norm(psi) - 1) <= 1e-12  # :40-53
    Hpsi = np.empty_like(psi)
    pkg.apply_H(Hpsi, psi, m)
    assert np.linalg.norm(Hpsi - E0 * psi) < 1e-10
    psi0 = np.array([1.0, 0.0], dtype=complex)
    exact = D.expm_herm(H, 0.3) @ psi0
    got = pkg.time_evolve(m, psi0, 0.3, method="krylov", kry_m=2)
    assert np.allclose(got, exact, atol=1e-10) and abs(np.linalg.norm(got) - 1) <= 1e-12   # :56-85
    assert np.allclose(pkg.time_evolve(m, psi0, 0.0, method="krylov", kry_m=2), psi0, atol=1e-12)
    got = pkg.time_evolve(m, psi0, 0.3, method="chebyshev", cheb_n=30, Ebounds=(-0.75, 0.25))
    assert np.allclose(got, exact, atol=1e-8) and abs(np.linalg.norm(got) - 1) <= 1e-8      # :96-118
    got = pkg.time_evolve(m, psi0, 0.1, method="chebyshev", cheb_n=20)                      # :121-134 auto bounds
    assert abs(np.linalg.norm(got) - 1) <= 1e-6


def test_reference_lanczos_tests(pkg, D):
    m = pkg.XXZChain(6, nup=3)                                                # test_Lanczos.jl:29-53
    H = D.dense_H(6, 3, *D.xxz_lists(6))
    w = np.linalg.eigvalsh(H)
    E0, psi = pkg.groundstate(m, lanc_m=m.N, seed=1)
    assert abs(E0 - w[0]) <= 1e-12
    assert np.linalg.norm(H @ psi - E0 * psi) < 1e-10
    m4 = pkg.XXZChain(4, nup=2)                                               # :74-100
    w4 = np.linalg.eigvalsh(D.dense_H(4, 2, *D.xxz_lists(4)))
    lo, hi = pkg.lanczos_extremal(pkg.apply_H, m4, lanc_m=m4.N, seed=5)
    assert abs(lo - w4[0]) <= 1e-12 and abs(hi - w4[-1]) <= 1e-12
    m2 = pkg.XXZChain(2, nup=1)                                               # :6-26
    v = np.array([1.0, 1.0j]) / np.sqrt(2)
    alpha, beta, nv = pkg.lanczos_tridiag(pkg.apply_H, m2, v, lanc_m=2)
    assert abs(alpha[0] - (-0.25)) <= 1e-14 and abs(nv - 1) <= 1e-14
    a6, b6, _ = pkg.lanczos_tridiag(pkg.apply_H, m, cvec(m.N, 2), lanc_m=50)  # :103-119
    assert len(b6) == len(a6) - 1 and len(a6) <= m.N
    with pytest.raises(pkg.ZeroNormError):                                    # src/Lanczos.jl:210-212
        pkg.lanczos_tridiag(pkg.apply_H, m, np.zeros(m.N, complex))
    e1 = pkg.groundstate(m, lanc_m=10, seed=9)[0]                             # :122-166 seeded reproducibility
    assert e1 == pkg.groundstate(m, lanc_m=10, seed=9)[0]


def test_reference_kpm_tests(pkg):
    m = pkg.XXZChain(6, nup=3)
    E0, gs = pkg.groundstate(m, lanc_m=20, seed=4)
    lo, hi = pkg.estimate_energy_bounds(pkg.apply_H, m, lanc_m=20, seed=11)   # test_KPM.jl:4-24
    a, b = pkg.get_rescaling_params(pkg.apply_H, m, lanc_m=20, seed=11)
    assert -1 < (lo - b) / a and (hi - b) / a < 1
    omega = np.arange(0.0, 5.0 + 1e-9, 0.01)                                  # :67-91 sum rule
    S = pkg.dynamical_structure_factor(m, gs, [np.pi], omega, method="kpm", kpm_m=120, kernel="jackson", seed=2)
    phi = pkg.Sz_q_vector(m, gs, np.pi)
    w_exact = np.linalg.norm(phi) ** 2
    assert np.all(np.isfinite(S)) and np.all(S >= 0)
    assert abs(S[0].sum() * 0.01 - w_exact) <= 5e-3 * w_exact
    S2 = pkg.dynamical_structure_factor(m, gs, [np.pi], np.linspace(0, 5, 300), method="kpm", kpm_m=100, seed=2)
    assert S2[:, -11:].max() < S2.max()                                       # :44-65
    m4 = pkg.XXZChain(4, nup=2)                                               # test_PublicAPI.jl:154-203
    _, g4 = pkg.groundstate(m4, lanc_m=6, seed=1)
    q = pkg.momenta(m4)
    S = pkg.dynamical_structure_factor(m4, g4, q, np.linspace(0, 3, 40), method="lanczos", lanc_m=6, eta=0.05)
    assert S.shape == (4, 40) and np.all(np.isfinite(S)) and np.all(S >= -1e-12)
    S = pkg.dynamical_structure_factor(m4, g4, q, np.linspace(-2, 2, 40), method="kpm", kpm_m=40, seed=3)
    assert S.shape == (4, 40) and np.all(np.isfinite(S))


# ---- HIP vs oracle on identical injected inputs ----

MODELS = [(10, 5, 1.0, 1.0, "open"), (12, 6, 1.0, 0.6, "open"), (14, 7, 1.0, 1.0, "open"), (11, 4, 0.8, 1.2, "periodic"),
          (16, 8, 1.0, 1.0, "open")]


@pytest.mark.parametrize("L,nup,Jxy,Jz,bc", MODELS)
def test_lanczos_family_vs_oracle(pkg, O, L, nup, Jxy, Jz, bc):
    m = pkg.XXZChain(L, Jxy=Jxy, Jz=Jz, nup=nup, boundary=bc)
    r = O.XXZChain(L, Jxy=Jxy, Jz=Jz, nup=nup, boundary=bc)
    p0 = cvec(m.N, 1)
    for neg in (False, True):
        got = pkg.lanczos_extremal(pkg.apply_H, m, lanc_m=40, psi0=p0, negate=neg)
        want = O.lanczos_extremal(r, p0, lanc_m=40, negate=neg)
        assert np.allclose(got, want, atol=1e-10)          # Ritz values after 40 steps; reduction-order noise only
    pa, pb = cvec(m.N, 2), cvec(m.N, 3)
    assert np.allclose(pkg.estimate_energy_bounds(pkg.apply_H, m, lanc_m=30, psi0_a=pa, psi0_b=pb),
                       O.estimate_energy_bounds(r, pa, pb, lanc_m=30), atol=1e-10)
    al, be, nv = pkg.lanczos_tridiag(pkg.apply_H, m, p0, lanc_m=15)
    al2, be2, nv2 = O.lanczos_tridiag(r, p0, lanc_m=15)
    assert len(al) == len(al2) and abs(nv - nv2) <= 1e-12 * nv2
    assert np.abs(al - al2).max() <= 1e-9 and np.abs(be - be2).max() <= 1e-9   # Lanczos amplifies rounding noise
    x0 = np.random.default_rng(4).standard_normal(m.N)
    E2, gs2 = O.lanczos_groundstate(r, x0, lanc_m=60)
    # full re-orthogonalisation in blocks of 8 columns (default) and column by column (the reference's order):
    # both within the bars of the oracle, and within 1e-12 / 1e-8 of each other
    res = {}
    try:
        for blocked in (True, False):
            m.ctx.set_gs_blocked(blocked)
            E, gs = pkg.lanczos_groundstate(pkg.apply_H, m, lanc_m=60, psi0=x0)
            assert abs(E - E2) <= 1e-10
            assert min(np.abs(gs - gs2).max(), np.abs(gs + gs2).max()) <= 1e-6     # eigenvector up to sign, converged part
            res[blocked] = (E, gs)
    finally:
        m.ctx.set_gs_blocked(True)
    assert abs(res[True][0] - res[False][0]) <= 1e-12
    assert min(np.abs(res[True][1] - res[False][1]).max(), np.abs(res[True][1] + res[False][1]).max()) <= 1e-8


@pytest.mark.parametrize("L,nup,Jxy,Jz,bc", MODELS)
def test_time_evolution_vs_oracle(pkg, O, L, nup, Jxy, Jz, bc):
    m = pkg.XXZChain(L, Jxy=Jxy, Jz=Jz, nup=nup, boundary=bc)
    r = O.XXZChain(L, Jxy=Jxy, Jz=Jz, nup=nup, boundary=bc)
    psi0 = cvec(m.N, 5)
    psi0 /= np.linalg.norm(psi0)
    got = pkg.krylov_time_evolve(psi0, 0.4, pkg.apply_H, m, kry_m=30)
    want = O.krylov_time_evolve(r, psi0, 0.4, kry_m=30)
    assert np.abs(got - want).max() <= 1e-11                                   # psi tolerance (fp64, 30 steps)
    pr = psi0.real.copy()
    assert np.abs(pkg.krylov_time_evolve(pr, 0.4, pkg.apply_H, m, kry_m=20) - O.krylov_time_evolve(r, pr, 0.4, kry_m=20)).max() <= 1e-11
    got = pkg.chebyshev_time_evolve(psi0, 0.4, pkg.apply_H, m, cheb_n=80, Ebounds=(-L / 2, L / 2))
    want = O.chebyshev_time_evolve(r, psi0, 0.4, cheb_n=80, Ebounds=(-L / 2, L / 2))
    # the fused device step performs the reference's arithmetic in the same order: agreement to a few ulp
    assert np.abs(got - want).max() <= 1e-14
    with pytest.raises(pkg.ArgumentError):
        pkg.chebyshev_time_evolve(pr, 0.4, pkg.apply_H, m)                     # needs ComplexF64 (Chebyshev.jl:36,98)
    with pytest.raises(pkg.DimensionMismatch):
        pkg.krylov_time_evolve(psi0[:-1], 0.4, pkg.apply_H, m)


@pytest.mark.parametrize("L,nup,Jxy,Jz,bc", MODELS[:4])
def test_kpm_and_sqw_vs_oracle(pkg, O, L, nup, Jxy, Jz, bc):
    m = pkg.XXZChain(L, Jxy=Jxy, Jz=Jz, nup=nup, boundary=bc)
    r = O.XXZChain(L, Jxy=Jxy, Jz=Jz, nup=nup, boundary=bc)
    x0 = np.random.default_rng(6).standard_normal(m.N)
    _, gs = O.lanczos_groundstate(r, x0, lanc_m=80)
    a, b = O.rescaling_from_bounds(-L / 2, L / 2)
    phi = O.Sz_q_vector(r, gs, np.pi)
    phi /= np.linalg.norm(phi)
    mu = pkg.compute_chebyshev_moments(pkg.apply_H, phi, 200, a, b, m)
    mu2 = O.compute_chebyshev_moments(r, phi, 200, a, b)
    assert np.abs(mu - mu2).max() <= 1e-12
    q = pkg.momenta(m)
    omega = np.arange(0.0, 4.0, 0.05)
    S = pkg.kpm_sqw(gs, m, q, omega, a=a, b=b, kpm_m=128)
    S2 = O.kpm_sqw(r, gs, q, omega, a, b, kpm_m=128)
    assert np.abs(S - S2).max() <= 1e-8 * max(1.0, np.abs(S2).max())           # BASELINE: 1e-8 rel on S(q,w)
    for kern in ("lorentz",):
        assert np.abs(pkg.kpm_sqw(gs, m, q[:2], omega, a=a, b=b, kpm_m=64, kernel=kern)
                      - O.kpm_sqw(r, gs, q[:2], omega, a, b, kpm_m=64, kernel=kern)).max() <= 1e-8
    # short recursion: orthogonality still holds, agreement to reduction-order noise
    Sl = pkg.lanczos_sqw(gs, m, q[1:4], omega, lanc_m=12, eta=0.05)
    Sl2 = O.lanczos_sqw(r, gs, q[1:4], omega, lanc_m=12, eta=0.05)
    assert np.abs(Sl - Sl2).max() <= 1e-8 * max(1.0, np.abs(Sl2).max())
    # long recursion without re-orthogonalisation (src/Lanczos.jl:196-246): rounding differences in dot/norm are
    # amplified chaotically once orthogonality is lost (ghost Ritz values), on the reference as much as here, so only
    # the broadened spectrum is comparable, to ~1e-3 ("parity unpinned" for un-converged Lanczos)
    Sl = pkg.lanczos_sqw(gs, m, q[1:4], omega, lanc_m=40, eta=0.05)
    Sl2 = O.lanczos_sqw(r, gs, q[1:4], omega, lanc_m=40, eta=0.05)
    assert np.abs(Sl - Sl2).max() <= 2e-3 * max(1.0, np.abs(Sl2).max())
    Sg = pkg.lanczos_sqw(gs, m, q[1:2], omega, lanc_m=12, eta=0.1, broaden="gauss")
    assert np.abs(Sg - O.lanczos_sqw(r, gs, q[1:2], omega, lanc_m=12, eta=0.1, broaden="gauss")).max() <= 1e-8


def test_kpm_sqw_momenta_in_one_batch_equal_one_momentum_at_a_time(pkg, O):
    """At launch-bound sizes the momenta's vectors share every launch of the moment recursion (sd_ctx_set_q_batch, default on;
    the reference threads over q, src/KPM_Sqw.jl:218).  Each vector sees exactly the arithmetic of a recursion of its own, so S(q, w)
    must be EQUAL -- to the bit -- to the one-momentum-at-a-time loop: for both moment routes, a real psi0 (paired momenta), a complex
    one (every q on its own, q = 0 dropping out of the batch as a zero vector), and bounds that make the reference's overflow guard
    fire for every momentum (the batch then hands each vector to the one-at-a-time recursion)."""
    L, nup = 14, 7
    m = pkg.XXZChain(L, nup=nup, Jz=0.9)
    r = O.XXZChain(L, nup=nup, Jz=0.9)
    x0 = np.random.default_rng(12).standard_normal(m.N)
    _, gs = O.lanczos_groundstate(r, x0, lanc_m=60)
    q = pkg.momenta(m)
    omega = np.arange(0.0, 4.0, 0.05)
    a, b = O.rescaling_from_bounds(-L / 2, L / 2)
    cases = [(gs, a, b, 65), (cvec(m.N, 3), a, b, 32), (gs, 0.6, 0.0, 24)]
    try:
        for doubling in (True, False):
            m.ctx.set_kpm_doubling(doubling)
            for psi0, aa, bb, M in cases:
                m.ctx.set_q_batch(True)
                n0 = m.ctx.apply_count()
                S_batch = pkg.kpm_sqw(psi0, m, q, omega, a=aa, b=bb, kpm_m=M)
                n_batch = m.ctx.apply_count() - n0
                m.ctx.set_q_batch(False)
                n0 = m.ctx.apply_count()
                S_serial = pkg.kpm_sqw(psi0, m, q, omega, a=aa, b=bb, kpm_m=M)
                n_serial = m.ctx.apply_count() - n0
                assert np.array_equal(S_batch, S_serial), (doubling, M)
                if aa == a:
                    assert n_batch == n_serial                       # the same operator applications, in fewer launches
                    assert np.abs(S_batch - O.kpm_sqw(r, psi0, q, omega, aa, bb, kpm_m=M)).max() <= 1e-8 * max(1.0, np.abs(S_batch).max())
    finally:
        m.ctx.set_kpm_doubling(True)
        m.ctx.set_q_batch(True)


def test_lanczos_sqw_momenta_in_one_batch_equal_one_momentum_at_a_time(pkg, O):
    """lanczos_sqw (src/LanczosSqw.jl:49-80; the reference threads over q, :65): at launch-bound sizes all momenta run in ONE
    recursion of two launches per step (a batched apply + a batched update pass that sums the partial lists itself).  Each
    vector's coefficients are those of a recursion of its own -- S equal to the bit -- and agree with the oracle."""
    L, nup = 14, 7
    m = pkg.XXZChain(L, nup=nup, Jz=1.1)
    r = O.XXZChain(L, nup=nup, Jz=1.1)
    x0 = np.random.default_rng(13).standard_normal(m.N)
    _, gs = O.lanczos_groundstate(r, x0, lanc_m=60)
    q = pkg.momenta(m)
    omega = np.arange(0.0, 4.0, 0.05)
    try:
        for psi0 in (gs, cvec(m.N, 4)):
            m.ctx.set_q_batch(True)
            n0 = m.ctx.apply_count()
            S_batch = pkg.lanczos_sqw(psi0, m, q, omega, lanc_m=12, eta=0.05)
            n_batch = m.ctx.apply_count() - n0
            m.ctx.set_q_batch(False)
            n0 = m.ctx.apply_count()
            S_serial = pkg.lanczos_sqw(psi0, m, q, omega, lanc_m=12, eta=0.05)
            assert np.array_equal(S_batch, S_serial)
            assert n_batch == m.ctx.apply_count() - n0
            S2 = O.lanczos_sqw(r, psi0, q, omega, lanc_m=12, eta=0.05)
            assert np.abs(S_batch - S2).max() <= 1e-8 * max(1.0, np.abs(S2).max())
    finally:
        m.ctx.set_q_batch(True)


@pytest.mark.parametrize("M", [2, 3, 4, 7, 64, 201])
def test_kpm_moment_doubling_equals_reference_loop(pkg, O, M):
    """Default: two moments per apply (mu_2n = 2<v_n|v_n> - mu_0, mu_2n+1 = 2Re<v_n|v_n+1> - mu_1).  It must give the
    moments of the reference's one-per-apply loop (src/KPM_Sqw.jl:103-124, = the oracle) to rounding: tolerance 1e-13
    absolute on moments of a normalised phi, for even and odd M."""
    L, nup = 14, 7
    m = pkg.XXZChain(L, Jz=0.8, nup=nup)
    r = O.XXZChain(L, Jz=0.8, nup=nup)
    rng = np.random.default_rng(M)
    phi = rng.standard_normal(m.N) + 1j * rng.standard_normal(m.N)
    phi /= np.linalg.norm(phi)
    a, b = O.rescaling_from_bounds(-L / 2, L / 2)
    want = O.compute_chebyshev_moments(r, phi, M, a, b)
    try:
        m.ctx.set_kpm_doubling(True)
        mu_d = pkg.compute_chebyshev_moments(pkg.apply_H, phi, M, a, b, m)
        m.ctx.set_kpm_doubling(False)
        mu_r = pkg.compute_chebyshev_moments(pkg.apply_H, phi, M, a, b, m)
    finally:
        m.ctx.set_kpm_doubling(True)
    assert mu_d.shape == mu_r.shape == (M,)
    assert np.abs(mu_r - want).max() <= 1e-13
    assert np.abs(mu_d - want).max() <= 1e-13


def test_kpm_doubling_falls_back_when_the_reference_guard_fires(pkg, O):
    """Bounds that do not contain the spectrum: |v_k| grows past 1e3 and the reference renormalises v_next
    (src/KPM_Sqw.jl:118-121).  The doubling identity does not describe that sequence, so the library reruns the
    reference loop; the moments then follow the oracle's (they grow like cosh, so the comparison is relative)."""
    L, nup = 12, 6
    m = pkg.XXZChain(L, nup=nup)
    r = O.XXZChain(L, nup=nup)
    phi = np.random.default_rng(1).standard_normal(m.N) + 0j
    phi /= np.linalg.norm(phi)
    a, b = 0.6, 0.0                                   # spectrum is ~[-5.4, 2.8] / 0.6: far outside [-1, 1]
    mu = pkg.compute_chebyshev_moments(pkg.apply_H, phi, 24, a, b, m)
    want = O.compute_chebyshev_moments(r, phi, 24, a, b)
    assert np.abs(want).max() > 100.0                 # far outside a Chebyshev moment's range [-1, 1]: the guard fired
    assert np.abs(mu - want).max() <= 1e-9 * np.abs(want).max()


def test_fused_epilogues_vs_unfused_device_ops(pkg, O):
    """apply_rescaled / Chebyshev step on torch device tensors equal the oracle's un-fused passes bit for bit."""
    import torch
    m = pkg.XXZChain(15, nup=7, Jz=0.9)
    r = O.XXZChain(15, nup=7, Jz=0.9)
    v = cvec(m.N, 8)
    u = cvec(m.N, 9)
    t = cvec(m.N, 10)
    a, b, c = 4.1, -0.35, 0.3 - 0.2j
    dv, du, dt = (torch.from_numpy(x).cuda() for x in (v, u, t))
    dn = torch.empty_like(dv)
    pkg.cheb_step(dn, dv, du, dt, m, a, b, c)
    want_next = 2 * O.apply_rescaled_H(r, v, a, b) - u
    assert np.array_equal(dn.cpu().numpy(), want_next)
    cr, ci = c.real, c.imag
    want_t = t + (cr * want_next.real - ci * want_next.imag) + 1j * (cr * want_next.imag + ci * want_next.real)
    assert np.array_equal(dt.cpu().numpy(), want_t)
    out = torch.empty_like(dv)
    pkg.apply_H(out, dv, m)
    assert np.array_equal(out.cpu().numpy(), O.apply_H(r, v))


@pytest.mark.parametrize("cheb_n", [1, 2, 3, 4, 5, 8, 13])
def test_chebyshev_pairing_is_bit_identical_to_one_term_per_pass(pkg, cheb_n):
    """sd_chebyshev_evolve and the sharded driver take the terms in pairs (RECUR + CHEB2 epilogues: psi_t is read and
    written once per two terms).  Both must give the same BITS as one fused cheb_step per term (the reference's loop,
    src/TimeEvolution/Chebyshev.jl:110-121), for even and odd numbers of terms."""
    import torch
    L, nup = 14, 7
    m = pkg.XXZChain(L, nup=nup, Jz=0.7)
    psi0 = cvec(m.N, 21)
    psi0 /= np.linalg.norm(psi0)
    dt, Eb = 0.37, (-7.5, 4.0)
    a, b = (Eb[1] - Eb[0]) / (2 * 0.9999), (Eb[1] + Eb[0]) / 2
    c = pkg.chebyshev_coeffs(cheb_n, a, b, dt)
    prev = torch.from_numpy(psi0).cuda()
    cur, nxt = torch.empty_like(prev), torch.empty_like(prev)
    pkg.apply_rescaled_H(cur, prev, pkg.apply_H, m, a, b)
    acc = torch.zeros_like(prev)
    acc += complex(c[0]) * prev
    if cheb_n >= 2:
        acc += complex(c[1]) * cur
    for k in range(2, cheb_n):
        pkg.cheb_step(nxt, cur, prev, acc, m, a, b, complex(c[k]))
        prev, cur, nxt = cur, nxt, prev
    want = acc.cpu().numpy()
    got = pkg.chebyshev_time_evolve(psi0, dt, pkg.apply_H, m, cheb_n=cheb_n, Ebounds=Eb)
    got_sharded = pkg.ShardedOperator(m, 0, 1).chebyshev_time_evolve(torch.from_numpy(psi0).cuda(), dt, cheb_n=cheb_n,
                                                                   Ebounds=Eb).cpu().numpy()
    # the C recursion forms the first two terms in k_cheb_init (unfused multiply-add; torch may contract its complex
    # product), so psi_t can start an ulp away from the torch expression above: 1e-15 absolute (|psi_t| <= 1) ...
    assert np.abs(got - want).max() <= 1e-15
    # ... and EXACTLY equal when the one-term-per-pass loop starts from the recursion's own first two terms: this is what
    # pins the pairing (RECUR + CHEB2 epilogues) to the bits of one accumulation per term
    acc2 = torch.from_numpy(pkg.chebyshev_time_evolve(psi0, dt, pkg.apply_H, m, cheb_n=min(cheb_n, 2), Ebounds=Eb)).cuda()
    prev = torch.from_numpy(psi0).cuda()
    cur, nxt = torch.empty_like(prev), torch.empty_like(prev)
    pkg.apply_rescaled_H(cur, prev, pkg.apply_H, m, a, b)
    for k in range(2, cheb_n):
        pkg.cheb_step(nxt, cur, prev, acc2, m, a, b, complex(c[k]))
        prev, cur, nxt = cur, nxt, prev
    assert np.array_equal(got, acc2.cpu().numpy())
    # the sharded driver (one rank) runs the same C recursion through sd_chebyshev_evolve_sharded
    assert np.array_equal(got_sharded, got)


def test_chebyshev_evolve_device_resident(pkg):
    """sd_chebyshev_evolve_dev: the state stays on the GPU between the steps of a time evolution.  Same bits as the
    host-pointer call; in place (psit == psi0) allowed at the C level."""
    import torch
    L, nup = 14, 7
    m = pkg.XXZChain(L, nup=nup, Jz=0.6)
    psi0 = cvec(m.N, 33)
    psi0 /= np.linalg.norm(psi0)
    Eb = (-7.0, 4.0)
    host = psi0.copy()
    dev = torch.from_numpy(psi0).cuda()
    for _ in range(3):                                     # three steps of dt = 0.2
        host = pkg.chebyshev_time_evolve(host, 0.2, pkg.apply_H, m, cheb_n=25, Ebounds=Eb)
        dev = pkg.chebyshev_time_evolve(dev, 0.2, pkg.apply_H, m, cheb_n=25, Ebounds=Eb)
    assert isinstance(dev, torch.Tensor) and dev.is_cuda
    assert np.array_equal(dev.cpu().numpy(), host)
    # in place
    z = torch.from_numpy(psi0).cuda()
    m.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    pkg.check(pkg.lib().sd_chebyshev_evolve_dev(m.ctx.h, m.h, z.data_ptr(), m.N, 0.2, 25, Eb[0], Eb[1], z.data_ptr()), m.ctx.h)
    assert np.array_equal(z.cpu().numpy(), pkg.chebyshev_time_evolve(psi0, 0.2, pkg.apply_H, m, cheb_n=25, Ebounds=Eb))
    with pytest.raises(pkg.ArgumentError):
        pkg.chebyshev_time_evolve(torch.ones(m.N, dtype=torch.float64, device="cuda"), 0.1, pkg.apply_H, m, Ebounds=Eb)


def test_krylov_evolve_device_resident(pkg):
    """sd_krylov_evolve_dev: same bits as the host-pointer call, for real and complex device states, over several steps."""
    import torch
    m = pkg.XXZChain(14, nup=7, Jz=0.6)
    psi0 = cvec(m.N, 34)
    psi0 /= np.linalg.norm(psi0)
    host, dev = psi0.copy(), torch.from_numpy(psi0).cuda()
    for _ in range(3):
        host = pkg.krylov_time_evolve(host, 0.3, pkg.apply_H, m, kry_m=12)
        dev = pkg.krylov_time_evolve(dev, 0.3, pkg.apply_H, m, kry_m=12)
    assert np.array_equal(dev.cpu().numpy(), host)
    real0 = np.random.default_rng(2).standard_normal(m.N)
    a = pkg.krylov_time_evolve(real0, 0.3, pkg.apply_H, m, kry_m=8)
    b = pkg.krylov_time_evolve(torch.from_numpy(real0).cuda(), 0.3, pkg.apply_H, m, kry_m=8)
    assert b.dtype == torch.complex128 and np.array_equal(b.cpu().numpy(), a)
    z = torch.zeros(m.N, dtype=torch.complex128, device="cuda")          # zero state comes back unchanged (Krylov.jl:145-147)
    assert float(pkg.krylov_time_evolve(z, 0.3, pkg.apply_H, m, kry_m=4).abs().max()) == 0.0


def test_lanczos_tridiag_breakdown_is_applied_after_the_queued_recursion(pkg, O, D):
    """The device recursion queues every step without a host round trip and applies the reference's break on beta_j < tol
    (src/Lanczos.jl:228-231) afterwards.  An exact eigenvector as start vector breaks down at the first step: alpha = [E],
    beta = [] (or one tiny entry), nothing of the discarded steps (inf / NaN by then) leaks out."""
    L, nup = 8, 4
    m = pkg.XXZChain(L, nup=nup)
    r = O.XXZChain(L, nup=nup)
    v = np.ones(m.N, dtype=complex)                 # |F>: H|F> = (L-1)/4 |F> at the Heisenberg point
    al, be, nv = pkg.lanczos_tridiag(pkg.apply_H, m, v, lanc_m=12)
    assert len(al) == 1 and abs(al[0] - (L - 1) / 4) < 1e-14 and np.all(np.isfinite(al)) and np.all(np.isfinite(be))
    assert abs(nv - np.sqrt(m.N)) < 1e-12
    # breakdown in the middle: a start vector inside a 3-dimensional invariant subspace
    hop, zz, f = D.xxz_lists(L)
    w, U = np.linalg.eigh(D.dense_H(L, nup, hop, zz, f))
    pick = [0, 5, 11]                                # three non-degenerate levels
    assert min(abs(w[i] - w[j]) for i in pick for j in range(len(w)) if j != i) > 1e-6
    v3 = (U[:, pick[0]] + 0.5 * U[:, pick[1]] - 0.25 * U[:, pick[2]]).astype(complex)
    al, be, _ = pkg.lanczos_tridiag(pkg.apply_H, m, v3, lanc_m=12, tol=1e-9)
    al_o, be_o, _ = O.lanczos_tridiag(r, v3, lanc_m=12, tol=1e-9)
    assert len(al) == len(al_o) == 3 and np.all(np.isfinite(al)) and np.all(np.isfinite(be))
    T = np.diag(al) + np.diag(be[:2], 1) + np.diag(be[:2], -1)
    assert np.abs(np.linalg.eigvalsh(T) - np.sort(w[pick])).max() < 1e-9


def test_long_queued_lanczos_stops_soon_after_a_breakdown(pkg):
    """A breakdown at step 1 with lanc_m = 12000 (capped at N = 12870): the queued recursion looks at the betas every 32 steps
    (SD_BREAK_PEEK, csrc/recur.cpp) and stops; the result is the reference's (alpha = [E], m_eff = 1) and the call queues at
    most two peek intervals of garbage steps, counted by the library itself (sd_ctx_apply_count) -- no wall-clock bound."""
    L, nup = 16, 8
    m = pkg.XXZChain(L, nup=nup)
    v = np.ones(m.N, dtype=complex)
    n0 = m.ctx.apply_count()
    pkg.lanczos_tridiag(pkg.apply_H, m, v, lanc_m=4)
    assert m.ctx.apply_count() - n0 == 4              # the counter counts recursion steps
    n0 = m.ctx.apply_count()
    al, be, _ = pkg.lanczos_tridiag(pkg.apply_H, m, v, lanc_m=12000)
    steps = m.ctx.apply_count() - n0
    assert len(al) == 1 and abs(al[0] - (L - 1) / 4) < 1e-13
    assert 1 <= steps <= 64, steps                    # not 12000: stopped at the first or second look at the betas
    n0 = m.ctx.apply_count()
    lo, hi = pkg.lanczos_extremal(pkg.apply_H, m, lanc_m=300, psi0=v)
    assert abs(lo - (L - 1) / 4) < 1e-13 and abs(hi - (L - 1) / 4) < 1e-13
    assert 1 <= m.ctx.apply_count() - n0 <= 64


def test_user_operator_at_recursion_level(pkg, O):
    """The reference's solvers take the operator as a callable (src/Lanczos.jl:27-29, src/TimeEvolution/Chebyshev.jl:61-64,
    src/KPM_Sqw.jl:95-98).  A user callable applyH(out, psi, model) on device tensors replaces the built-in kernel inside every
    recursion (sd_ctx_set_apply_callback): here 2*H written as a closure over the library's own apply, against the built-in
    operator of the model with doubled couplings (a power-of-two scale: every intermediate doubles exactly)."""
    L, nup = 12, 6
    m = pkg.XXZChain(L, nup=nup, Jz=0.7, hz=0.1)
    m2 = pkg.XXZChain(L, nup=nup, Jxy=2.0, Jz=1.4, hz=0.2)
    calls = []

    def twice(out, psi, model):
        assert out.is_cuda and psi.is_cuda and out.shape == psi.shape == (model.N,)
        calls.append(psi.dtype)
        pkg.apply_H(out, psi, model)
        out.mul_(2.0)

    psi0 = cvec(m.N, 5)
    psi0 /= np.linalg.norm(psi0)
    lo, hi = pkg.lanczos_extremal(twice, m, lanc_m=60, psi0=psi0)
    lo2, hi2 = pkg.lanczos_extremal(pkg.apply_H, m2, lanc_m=60, psi0=psi0)
    assert abs(lo - lo2) < 1e-10 and abs(hi - hi2) < 1e-10 and len(calls) >= 60
    al, be, nv = pkg.lanczos_tridiag(twice, m, psi0, lanc_m=20)
    al2, be2, nv2 = pkg.lanczos_tridiag(pkg.apply_H, m2, psi0, lanc_m=20)
    assert np.allclose(al, al2, atol=1e-11) and np.allclose(be, be2, atol=1e-11)
    a = pkg.chebyshev_time_evolve(psi0, 0.3, twice, m, cheb_n=40, Ebounds=(2 * lo2 / 2 - 1.0, hi2 + 1.0))
    b = pkg.chebyshev_time_evolve(psi0, 0.3, pkg.apply_H, m2, cheb_n=40, Ebounds=(2 * lo2 / 2 - 1.0, hi2 + 1.0))
    assert np.abs(a - b).max() < 1e-13
    a = pkg.krylov_time_evolve(psi0, 0.3, twice, m, kry_m=20)
    b = pkg.krylov_time_evolve(psi0, 0.3, pkg.apply_H, m2, kry_m=20)
    assert np.abs(a - b).max() < 1e-12
    aa, bb = pkg.rescaling_from_bounds(lo2, hi2)
    mu = pkg.compute_chebyshev_moments(twice, psi0, 33, aa, bb, m)
    mu2 = pkg.compute_chebyshev_moments(pkg.apply_H, psi0, 33, aa, bb, m2)
    assert np.abs(mu - mu2).max() < 1e-13
    x0 = np.random.default_rng(3).standard_normal(m.N)
    calls.clear()
    E, gs = pkg.lanczos_groundstate(twice, m, lanc_m=50, psi0=x0)
    E2, gs2 = pkg.lanczos_groundstate(pkg.apply_H, m2, lanc_m=50, psi0=x0)
    assert abs(E - E2) < 1e-10 and calls and all(str(d) == "torch.float64" for d in calls)
    # against the oracle's operator too: E is the ground energy of 2 H
    r2 = O.XXZChain(L, nup=nup, Jxy=2.0, Jz=1.4, hz=0.2)
    assert np.linalg.norm(O.apply_H(r2, gs) - E * gs) < 1e-6

    # the caller's exception is what comes back, and the built-in operator is in place again afterwards
    def bad(out, psi, model):
        raise ValueError("boom")

    with pytest.raises(ValueError, match="boom"):
        pkg.lanczos_extremal(bad, m, lanc_m=10, psi0=psi0)
    n_before = len(calls)
    lo3, hi3 = pkg.lanczos_extremal(pkg.apply_H, m, lanc_m=60, psi0=psi0)
    assert len(calls) == n_before and abs(2 * lo3 - lo) < 1e-10
    with pytest.raises(pkg.ArgumentError):
        pkg.lanczos_extremal("not callable", m)


def test_set_apply_belongs_to_the_context_and_dies_with_its_model(pkg):
    """Model.set_apply installs the operator on the model's CONTEXT (sd_ctx_set_apply_callback).  Models built without ctx=
    share the default context, so: (i) a second model may not replace an installed operator silently; (ii) when the
    installing model is garbage-collected the context goes back to the built-in H instead of keeping a pointer to a freed
    thunk; (iii) the solvers' applyH= argument restores a persistently installed operator; (iv) the cached Chebyshev
    bounds are not reused across operators."""
    import gc
    L, nup = 12, 6
    m1 = pkg.XXZChain(L, nup=nup)
    m2 = pkg.XXZChain(L, nup=nup, Jz=0.5)
    assert m1.ctx is m2.ctx
    psi0 = cvec(m2.N, 3)
    psi0 /= np.linalg.norm(psi0)
    want = pkg.lanczos_extremal(pkg.apply_H, m2, lanc_m=30, psi0=psi0)
    calls = []

    def thrice(out, psi, model):
        calls.append(id(model))                # (no reference to the model itself: it must be collectable below)
        pkg.apply_H(out, psi, model)
        out.mul_(3.0)

    m1.set_apply(thrice)
    lo, hi = pkg.lanczos_extremal(pkg.apply_H, m1, lanc_m=30, psi0=psi0)       # "built-in" = whatever the context holds
    assert calls and all(c == id(m1) for c in calls)
    lo1, hi1 = 3 * np.array(pkg.lanczos_extremal(pkg.apply_H, pkg.XXZChain(L, nup=nup, ctx=pkg.Context(0)), lanc_m=30, psi0=psi0))
    assert abs(lo - lo1) < 1e-9 and abs(hi - hi1) < 1e-9
    with pytest.raises(pkg.ArgumentError):                                    # (i)
        m2.set_apply(thrice)
    n = len(calls)                                                            # (iii): a one-call operator, then thrice is back
    pkg.lanczos_extremal(lambda o, p, mm: pkg.apply_H(o, p, mm), m1, lanc_m=5, psi0=psi0)
    assert len(calls) == n
    pkg.lanczos_extremal(pkg.apply_H, m1, lanc_m=5, psi0=psi0)
    assert len(calls) > n
    pkg.time_evolve(m1, psi0, 0.05, method="chebyshev", cheb_n=10)            # (iv): estimated with thrice installed, not cached
    assert "_energy_bounds" not in m1.__dict__ or not m1.__dict__["_energy_bounds"]
    del m1                                                                    # (ii)
    gc.collect()
    assert m2.ctx._apply_cb is None and m2.ctx._apply_owner is None
    n = len(calls)
    got = pkg.lanczos_extremal(pkg.apply_H, m2, lanc_m=30, psi0=psi0)
    assert len(calls) == n and got == want


def test_groundstate_deferred_orthogonality_check_and_its_fallback(pkg, O):
    """lanczos_groundstate runs the reference's orthogonality check of step j (src/Lanczos.jl:142-153) together with the Gram-Schmidt
    passes of step j + 1 and corrects the step before, the reference's way, when a check fires.  With the default tolerance the
    check never fires; with orthogonalize_tol = 1e-18 it fires on every step (rounding-level overlaps), so every step takes the
    correction-and-redo path: E0 and the vector must still agree with the oracle, which runs the reference's loop literally."""
    for (L, nup, lm) in ((12, 6, 30), (14, 7, 41)):
        m = pkg.XXZChain(L, Jz=0.9, nup=nup)
        r = O.XXZChain(L, Jz=0.9, nup=nup)
        x0 = np.random.default_rng(50 + L).standard_normal(m.N)
        for otol in (1e-10, 1e-18):
            E2, gs2 = O.lanczos_groundstate(r, x0, lanc_m=lm, orthogonalize_tol=otol)
            E, gs = pkg.lanczos_groundstate(pkg.apply_H, m, lanc_m=lm, psi0=x0, orthogonalize_tol=otol)
            assert abs(E - E2) <= 1e-10, (L, otol, E - E2)
            assert min(np.abs(gs - gs2).max(), np.abs(gs + gs2).max()) <= 1e-6
    # a start vector that is an exact eigenvector (uniform state at the Heisenberg point): beta_1 = 0 -> breakdown after one step,
    # as the reference breaks (:136-139); the steps queued behind it are discarded
    mh = pkg.XXZChain(12, nup=6)
    E, gs = pkg.lanczos_groundstate(pkg.apply_H, mh, lanc_m=10, psi0=np.ones(mh.N))
    assert abs(E - 11 / 4) <= 1e-12 and np.abs(np.abs(gs) - 1 / np.sqrt(mh.N)).max() <= 1e-12


def test_blocked_gram_schmidt_chain_sizes(pkg, O):
    """lanczos_groundstate with the blocked re-orthogonalisation across the block boundaries (1, 8, 9, 16, 17, 25 columns) and
    an odd dimension (scalar tail of the 16-byte loop): E0 against the oracle and against the column-by-column chain."""
    for (L, nup, lm) in ((10, 5, 2), (10, 5, 9), (10, 5, 10), (12, 6, 17), (12, 6, 18), (11, 5, 26), (13, 6, 40), (7, 3, 10), (15, 7, 12)):   # C(7,3), C(15,7) odd
        m = pkg.XXZChain(L, Jz=0.8, nup=nup)
        r = O.XXZChain(L, Jz=0.8, nup=nup)
        x0 = np.random.default_rng(L + lm).standard_normal(m.N)
        E2, gs2 = O.lanczos_groundstate(r, x0, lanc_m=lm)
        try:
            m.ctx.set_gs_blocked(True)
            Eb, gb = pkg.lanczos_groundstate(pkg.apply_H, m, lanc_m=lm, psi0=x0)
            m.ctx.set_gs_blocked(False)
            Es, gss = pkg.lanczos_groundstate(pkg.apply_H, m, lanc_m=lm, psi0=x0)
        finally:
            m.ctx.set_gs_blocked(True)
        assert abs(Eb - E2) <= 1e-10 and abs(Es - E2) <= 1e-10 and abs(Eb - Es) <= 1e-12, (L, nup, lm)
        assert min(np.abs(gb - gss).max(), np.abs(gb + gss).max()) <= 1e-8, (L, nup, lm)


def test_kpm_sqw_pairs_q_with_2pi_minus_q_for_a_real_psi0(pkg, O):
    """H is real, so for a real psi0 phi_{2pi-q} = conj(phi_q) and the moments of q and 2pi - q agree: sd_kpm_sqw computes
    each pair of momenta(model) once and copies the row (the reference recomputes it, src/KPM_Sqw.jl:218-252).
    Pairing on == pairing off to 1e-12 (rounding of exp(iqr)), both within the 1e-8 bar of the oracle; a complex psi0, or a
    Float64 one passed as ComplexF64 with a non-zero imaginary part somewhere, is never paired."""
    L, nup = 12, 6
    m = pkg.XXZChain(L, Jz=0.9, nup=nup)
    r = O.XXZChain(L, Jz=0.9, nup=nup)
    rng = np.random.default_rng(3)
    psi = rng.standard_normal(m.N)
    psi /= np.linalg.norm(psi)
    q, omega = pkg.momenta(m), np.arange(-1.0, 4.0, 0.1)
    a, b = O.rescaling_from_bounds(-L / 2, L / 2)
    want = O.kpm_sqw(r, psi, q, omega, a, b, kpm_m=96)
    try:
        for vec in (psi, psi.astype(np.complex128)):
            m.ctx.set_kpm_pair_q(True)
            S_on = pkg.kpm_sqw(vec, m, q, omega, a=a, b=b, kpm_m=96)
            m.ctx.set_kpm_pair_q(False)
            S_off = pkg.kpm_sqw(vec, m, q, omega, a=a, b=b, kpm_m=96)
            for n in range(1, L):
                assert np.array_equal(S_on[n], S_on[L - n])                    # the row was copied
            assert np.abs(S_on - S_off).max() <= 1e-12 * max(1.0, np.abs(S_off).max())
            assert np.abs(S_on - want).max() <= 1e-8 * max(1.0, np.abs(want).max())
            assert np.abs(S_off - want).max() <= 1e-8 * max(1.0, np.abs(want).max())
        # complex psi0: S(q) != S(2pi - q) in general, nothing may be copied
        m.ctx.set_kpm_pair_q(True)
        psic = psi + 1j * rng.standard_normal(m.N) * 0.3
        psic /= np.linalg.norm(psic)
        Sc = pkg.kpm_sqw(psic, m, q, omega, a=a, b=b, kpm_m=96)
        wantc = O.kpm_sqw(r, psic, q, omega, a, b, kpm_m=96)
        assert np.abs(Sc - wantc).max() <= 1e-8 * max(1.0, np.abs(wantc).max())
        assert not np.array_equal(Sc[1], Sc[L - 1])
        # the Lanczos S(q, w) pairs the same way (equal alpha_j, beta_j for phi and conj(phi))
        m.ctx.set_kpm_pair_q(True)
        Sl_on = pkg.lanczos_sqw(psi, m, q, omega, lanc_m=12, eta=0.05)
        m.ctx.set_kpm_pair_q(False)
        Sl_off = pkg.lanczos_sqw(psi, m, q, omega, lanc_m=12, eta=0.05)
        m.ctx.set_kpm_pair_q(True)
        for n in range(1, L):
            assert np.array_equal(Sl_on[n], Sl_on[L - n])
        assert np.abs(Sl_on - Sl_off).max() <= 1e-9 * max(1.0, np.abs(Sl_off).max())
        wl = O.lanczos_sqw(r, psi, q, omega, lanc_m=12, eta=0.05)
        assert np.abs(Sl_on - wl).max() <= 1e-8 * max(1.0, np.abs(wl).max())
        # a list that holds q but not 2pi - q is computed as it stands
        S3 = pkg.kpm_sqw(psi, m, q[:3], omega, a=a, b=b, kpm_m=96)
        assert np.abs(S3 - want[:3]).max() <= 1e-8 * max(1.0, np.abs(want).max())
    finally:
        m.ctx.set_kpm_pair_q(True)


def test_time_evolve_chebyshev_estimates_its_bounds_once_per_model(pkg):
    """time_evolve(:chebyshev) without Ebounds (src/PublicAPI.jl:68-75): the bounds come from two 80-step Lanczos runs on start
    vectors of the counter-based generator, i.e. they are the same for every call with the same (model, seed) -- so the mirror
    computes them once per model: the second call returns the same bits and queues only the recursion's own applies."""
    m = pkg.XXZChain(12, nup=6, Jz=0.7)
    psi0 = np.random.default_rng(2).standard_normal(m.N) + 1j * np.random.default_rng(3).standard_normal(m.N)
    psi0 /= np.linalg.norm(psi0)
    n0 = m.ctx.apply_count()
    a = pkg.time_evolve(m, psi0, 0.2, method="chebyshev", cheb_n=20)
    n1 = m.ctx.apply_count()
    b = pkg.time_evolve(m, psi0, 0.2, method="chebyshev", cheb_n=20)
    n2 = m.ctx.apply_count()
    assert np.array_equal(a, b)
    assert n1 - n0 > 100 and n2 - n1 <= 20          # first call: 2 x 80 Lanczos steps + 19 terms; second: the terms alone
    bounds = pkg.estimate_energy_bounds(pkg.apply_H, m, seed=0)
    c = pkg.time_evolve(m, psi0, 0.2, method="chebyshev", cheb_n=20, Ebounds=bounds)
    assert np.array_equal(a, c)                     # the cached estimate is the one an explicit call gives


@pytest.mark.parametrize("L,nup,periodic", [(14, 7, False), (16, 8, True)])
def test_recursions_on_a_chain_with_second_neighbour_bonds_vs_oracle(pkg, O, L, nup, periodic):
    """build_model with J1-J2 bonds (the general-bond plan of k_apply_tiled) under every fused store of the recursions: the Lanczos dot,
    the Chebyshev term, the KPM step with two moments per apply, and the momenta of S(q,w) in one batch."""
    hop, zz = [], []
    for d, J in ((1, 1.0), (2, 0.45)):
        for i in range(1, L + 1):
            j = i + d
            if j > L:
                if not periodic:
                    continue
                j -= L
            hop.append((i, j, 0.5 * J)); zz.append((i, j, J))
    m = pkg.build_model(L, nup=nup, hopping=hop, zz=zz)
    r = O.build_model(L, nup=nup, hopping=hop, zz=zz)
    p0 = cvec(m.N, 1)
    al, be, nv = pkg.lanczos_tridiag(pkg.apply_H, m, p0, lanc_m=15)
    al2, be2, nv2 = O.lanczos_tridiag(r, p0, lanc_m=15)
    assert np.abs(al - al2).max() <= 1e-9 and np.abs(be - be2).max() <= 1e-9 and abs(nv - nv2) <= 1e-12 * nv2
    x0 = np.random.default_rng(4).standard_normal(m.N)
    E, gs = pkg.lanczos_groundstate(pkg.apply_H, m, lanc_m=60, psi0=x0)
    E2, gs2 = O.lanczos_groundstate(r, x0, lanc_m=60)
    assert abs(E - E2) <= 1e-10
    psi0 = cvec(m.N, 5)
    psi0 /= np.linalg.norm(psi0)
    assert np.abs(pkg.krylov_time_evolve(psi0, 0.4, pkg.apply_H, m, kry_m=30) - O.krylov_time_evolve(r, psi0, 0.4, kry_m=30)).max() <= 1e-11
    got = pkg.chebyshev_time_evolve(psi0, 0.4, pkg.apply_H, m, cheb_n=80, Ebounds=(-L / 2, L / 2))
    assert np.abs(got - O.chebyshev_time_evolve(r, psi0, 0.4, cheb_n=80, Ebounds=(-L / 2, L / 2))).max() <= 1e-14
    a, b = O.rescaling_from_bounds(-L / 2, L / 2)
    phi = O.Sz_q_vector(r, gs2, np.pi)
    phi /= np.linalg.norm(phi)
    assert np.abs(pkg.compute_chebyshev_moments(pkg.apply_H, phi, 120, a, b, m) - O.compute_chebyshev_moments(r, phi, 120, a, b)).max() <= 1e-12
    q, omega = pkg.momenta(m), np.arange(0.0, 4.0, 0.05)
    S = pkg.kpm_sqw(gs2, m, q, omega, a=a, b=b, kpm_m=96)
    assert np.abs(S - O.kpm_sqw(r, gs2, q, omega, a, b, kpm_m=96)).max() <= 1e-8 * max(1.0, np.abs(S).max())


@pytest.mark.parametrize("L,nup,bc", [(20, 3, "open"), (22, 19, "periodic")])
def test_recursions_on_the_per_row_path_vs_oracle(pkg, O, L, nup, bc):
    """A small, very dilute sector runs through the per-row kernel (closed-form chain partners; every fused store of the recursions goes
    through its epilogue too): Lanczos, ground state, Krylov, Chebyshev, KPM moments and S(q,w) against the oracle."""
    m = pkg.XXZChain(L, Jxy=0.9, Jz=1.3, nup=nup, boundary=bc)
    r = O.XXZChain(L, Jxy=0.9, Jz=1.3, nup=nup, boundary=bc)
    assert m.device_path == "generic"
    p0 = cvec(m.N, 1)
    al, be, nv = pkg.lanczos_tridiag(pkg.apply_H, m, p0, lanc_m=15)
    al2, be2, nv2 = O.lanczos_tridiag(r, p0, lanc_m=15)
    assert np.abs(al - al2).max() <= 1e-9 and np.abs(be - be2).max() <= 1e-9 and abs(nv - nv2) <= 1e-12 * nv2
    x0 = np.random.default_rng(4).standard_normal(m.N)
    E, gs = pkg.lanczos_groundstate(pkg.apply_H, m, lanc_m=60, psi0=x0)
    E2, gs2 = O.lanczos_groundstate(r, x0, lanc_m=60)
    assert abs(E - E2) <= 1e-10
    psi0 = cvec(m.N, 5)
    psi0 /= np.linalg.norm(psi0)
    assert np.abs(pkg.krylov_time_evolve(psi0, 0.4, pkg.apply_H, m, kry_m=30) - O.krylov_time_evolve(r, psi0, 0.4, kry_m=30)).max() <= 1e-11
    got = pkg.chebyshev_time_evolve(psi0, 0.4, pkg.apply_H, m, cheb_n=80, Ebounds=(-L / 2, L / 2))
    assert np.abs(got - O.chebyshev_time_evolve(r, psi0, 0.4, cheb_n=80, Ebounds=(-L / 2, L / 2))).max() <= 1e-14
    a, b = O.rescaling_from_bounds(-L / 2, L / 2)
    phi = O.Sz_q_vector(r, gs2, np.pi)
    phi /= np.linalg.norm(phi)
    assert np.abs(pkg.compute_chebyshev_moments(pkg.apply_H, phi, 120, a, b, m) - O.compute_chebyshev_moments(r, phi, 120, a, b)).max() <= 1e-12
    q, omega = pkg.momenta(m), np.arange(0.0, 4.0, 0.05)
    S = pkg.kpm_sqw(gs2, m, q, omega, a=a, b=b, kpm_m=96)
    assert np.abs(S - O.kpm_sqw(r, gs2, q, omega, a, b, kpm_m=96)).max() <= 1e-8 * max(1.0, np.abs(S).max())


def test_recursions_with_the_short_tile_kernel_vs_oracle(pkg, O, monkeypatch):
    """SD_SHORT_TILES=1 moves every tile of at most 16 rows to k_apply_short whatever their number: its per-tile partial sums feed the
    Lanczos dot, the KPM step and the batched momenta exactly like the tiles' own."""
    monkeypatch.setenv("SD_SHORT_TILES", "1")
    monkeypatch.setenv("SD_SUFFIX_BITS", "10")
    L, nup = 18, 5
    m = pkg.XXZChain(L, Jxy=1.0, Jz=0.8, nup=nup)
    r = O.XXZChain(L, Jxy=1.0, Jz=0.8, nup=nup)
    assert m.device_path == "tiled"
    psi = cvec(m.N, 3)
    out = np.empty_like(psi)
    pkg.apply_H(out, psi, m)
    assert np.array_equal(out, O.apply_H(r, psi))
    al, be, _ = pkg.lanczos_tridiag(pkg.apply_H, m, psi, lanc_m=20)
    al2, be2, _ = O.lanczos_tridiag(r, psi, lanc_m=20)
    assert np.abs(al - al2).max() <= 1e-9 and np.abs(be - be2).max() <= 1e-9
    x0 = np.random.default_rng(4).standard_normal(m.N)
    _, gs = O.lanczos_groundstate(r, x0, lanc_m=60)
    a, b = O.rescaling_from_bounds(-L / 2, L / 2)
    q, omega = pkg.momenta(m), np.arange(0.0, 4.0, 0.05)
    S = pkg.kpm_sqw(gs, m, q, omega, a=a, b=b, kpm_m=96)                      # the momenta in one batch
    assert np.abs(S - O.kpm_sqw(r, gs, q, omega, a, b, kpm_m=96)).max() <= 1e-8 * max(1.0, np.abs(S).max())
    Sl = pkg.lanczos_sqw(gs, m, q[1:5], omega, lanc_m=12, eta=0.05)
    assert np.abs(Sl - O.lanczos_sqw(r, gs, q[1:5], omega, lanc_m=12, eta=0.05)).max() <= 1e-8 * max(1.0, np.abs(Sl).max())
