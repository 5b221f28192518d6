"""Size-independent properties at (and near) BASELINE.json's full sizes, where the CPU oracle cannot follow.

For the isotropic open chain (Jxy = Jz = 1) H = sum_<ij> (P_ij/2 - 1/4) with P_ij the site exchange.  With
|F> the uniform superposition of the sector and D_w = sum_r w_r S^z_r (diagonal), P_ij D_w |F> = D_{w o (ij)} |F>,
so   H D_w |F> = D_{w''} |F>,   w'' = sum_<ij> (w o (ij))/2 - w/4   -- an L x L computation.
psi = D_w|F> with complex w_r = e^{iqr}/sqrt(L) is exactly Sz_q_vector(model, ones, q), so the test runs the
Sz_q kernel and the apply kernel on every one of the N rows and checks a large sample of rows (all tile-boundary
regions included) against the closed form.  A wrong partner index on any bond changes the result at O(1)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def closed_form(model, rows, q):
    L = model.L
    w = np.exp(1j * q * np.arange(L)) / np.sqrt(L)
    w2 = -0.25 * (L - 1) * w
    for i in range(L - 1):
        ws = w.copy()
        ws[i], ws[i + 1] = w[i + 1], w[i]
        w2 = w2 + 0.5 * ws
    st = np.concatenate([model.states_range(int(r0), int(c)) for (r0, c) in rows])
    bits = ((st[:, None] >> np.arange(L, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.float64) - 0.5
    return bits @ w, bits @ w2


@pytest.mark.parametrize("L", [24, 28, 32, 34])     # L=34: N = 2.33e9 rows, past 32-bit row indices
def test_heisenberg_closed_form_full_size(pkg, L):
    import torch
    nup = L // 2
    model = pkg.XXZChain(L, nup=nup)
    N = model.N
    import gc
    gc.collect()
    torch.cuda.empty_cache()                 # what earlier tests left in torch's caching allocator is not "in use" ...
    pkg.default_context().release_scratch()  # ... nor are the work vectors the library's context keeps between calls
    free, _ = torch.cuda.mem_get_info()
    if free < 3 * 16 * N + (2 << 30):
        pytest.skip("not enough device memory")
    q = 2 * np.pi * 3 / L
    ones = torch.ones(N, dtype=torch.float64, device="cuda")
    psi = pkg.Sz_q_vector(model, ones, q)
    del ones
    out = torch.empty_like(psi)
    pkg.apply_H(out, psi, model)
    torch.cuda.synchronize()
    rng = np.random.default_rng(L)
    starts = np.unique(np.concatenate([[0, N - 4096], rng.integers(0, N - 4096, 60)]))
    rows = [(int(s), 4096) for s in starts]
    want_psi, want_out = closed_form(model, rows, q)
    idx = np.concatenate([np.arange(s, s + c) for (s, c) in rows])
    tidx = torch.from_numpy(idx).cuda()
    got_psi = psi[tidx].cpu().numpy()
    got_out = out[tidx].cpu().numpy()
    assert np.abs(got_psi - want_psi).max() <= 1e-13          # Sz_q kernel
    assert np.abs(got_out - want_out).max() <= 1e-12          # apply kernel (sum of <= L terms of size <= 1)
    # global checks over all N rows: <psi|H|psi> is real, and equals sum conj(psi) * closed form on the sample ratio
    vdot = pkg.ShardedOperator(model, 0, 1).dot       # the library's fixed-order reduction (torch.vdot stops at 2^31 elements)
    e = vdot(psi, out)
    assert abs(e.imag) <= 1e-9 * abs(e.real)
    # hermiticity with a second, random vector: <x|H psi> == conj(<psi|H x>)
    x = torch.empty_like(psi)
    model.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    pkg.lib().sd_fill_randn_dev(model.ctx.h, x.data_ptr(), 2 * N, 7, 0)
    lhs = vdot(x, out)
    pkg.apply_H(out, x, model)
    rhs = vdot(psi, out).conjugate()
    assert abs(lhs - rhs) <= 1e-9 * max(1.0, abs(lhs))
    if N < 2 ** 31:
        assert abs(torch.vdot(psi, out).item() - vdot(psi, out)) <= 1e-9 * max(1.0, abs(lhs))


def test_uniform_state_is_exact_eigenvector(pkg):
    """H |F> = (L-1)/4 |F> exactly in floating point (all partial sums are small dyadic rationals): every row."""
    import torch
    L = 28
    model = pkg.XXZChain(L, nup=L // 2)
    psi = torch.ones(model.N, dtype=torch.complex128, device="cuda")
    out = torch.empty_like(psi)
    pkg.apply_H(out, psi, model)
    assert bool((out == (L - 1) / 4).all())


@pytest.mark.parametrize("rank", [0, 3, 7])
def test_config5_shard_of_L36_exact(pkg, rank):
    """BASELINE config 5 (L=36, nup=18, N = 9.08e9, 8 ranks): one rank's shard at full size on one GPU.  For the uniform
    state every imported row is 1 too, so the halo can be filled without peers and H|F> = (L-1)/4 |F> must hold bit for
    bit on every owned row -- through the sharded tile tables, 64-bit global bases and the interior/boundary split."""
    import torch
    L, world = 36, 8
    model = pkg.XXZChain(L, nup=L // 2)
    assert model.N == 9075135300

    def fill_halo(op, psi, halo):
        halo.fill_(1.0)

    op = pkg.ShardedOperator(model, rank, world, exchange_fn=fill_halo)
    import gc
    gc.collect()
    torch.cuda.empty_cache()                 # what earlier tests left in torch's caching allocator is not "in use" ...
    pkg.default_context().release_scratch()  # ... nor are the work vectors the library's context keeps between calls
    free, _ = torch.cuda.mem_get_info()
    if free < 16 * (2 * op.n_local + op.n_halo) + (4 << 30):
        pytest.skip("not enough device memory")
    psi = op.empty(torch.complex128, "cuda")
    out = op.empty(torch.complex128, "cuda")
    psi.fill_(1.0)
    out.zero_()
    op.apply(out, psi)
    assert bool((out == (L - 1) / 4).all())
    # interior tiles alone never touch the halo: poison it and run part 1 + part 2 separately
    halo = op.halo(psi)
    halo.fill_(float("nan"))
    out.zero_()
    op._launch(out, psi, halo, 0, part=1)
    assert not bool(torch.isnan(out.real).any())
    halo.fill_(1.0)
    op._launch(out, psi, halo, 0, part=2)
    assert bool((out == (L - 1) / 4).all())
    if rank == 3:
        # one KPM moment step of config 5 on this shard (sd_kpm_step_sharded_dev: fused rescale + <phi|v> + |v|^2) with the
        # uniform state: v_next = ((L-1)/4 - b)/a exactly on every row, and both sums are exact multiples of n_local
        import ctypes as C
        a, b = 2.0, 0.25
        val = ((L - 1) / 4 - b) / a
        sums = (C.c_double * 2)()
        model.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        pkg.check(pkg.lib().sd_kpm_step_sharded_dev(model.ctx.h, model.h, out.data_ptr(), psi.data_ptr(), halo.data_ptr(), None,
                                                    psi.data_ptr(), op.n_local, a, b, 1, sums), model.ctx.h)
        assert bool((out == val).all())
        assert sums[0] == val * op.n_local and sums[1] == val * val * op.n_local
        # second form of the step: v_next = 2 (H v - b v)/a - v_prev with v_prev = v_curr = |F>
        pkg.check(pkg.lib().sd_kpm_step_sharded_dev(model.ctx.h, model.h, out.data_ptr(), psi.data_ptr(), halo.data_ptr(),
                                                    psi.data_ptr(), psi.data_ptr(), op.n_local, a, b, 0, sums), model.ctx.h)
        assert bool((out == 2 * val - 1).all())
        assert sums[0] == (2 * val - 1) * op.n_local and sums[1] == (2 * val - 1) ** 2 * op.n_local


def _random_state(pkg, model, seed):
    """normalised ComplexF64 state on the device from the library's counter-based generator"""
    import torch
    op = pkg.ShardedOperator(model, 0, 1)
    x = op.empty(torch.complex128, "cuda")
    op.fill_randn(x, seed)
    x /= op.norm(x)
    return op, x


def _energy(pkg, op, model, x, scratch):
    pkg.apply_H(scratch, x, model)
    return op.dot(x, scratch).real


def test_config2_krylov_L28_full_size(pkg):
    """BASELINE config 2 at its real size (L=28, nup=14, kry_m=30, one MI355X): the CPU oracle cannot follow, so the
    recursion is checked through what it must conserve -- the norm (1e-12) and <H> (1e-10: V'HV = T makes
    <psi(t)|H|psi(t)> = T_11 = <psi0|H|psi0> up to the loss of orthogonality of 30 Lanczos vectors)."""
    import torch
    L = 28
    model = pkg.XXZChain(L, nup=L // 2)
    op, x = _random_state(pkg, model, 11)
    scratch = torch.empty_like(x)
    e0 = _energy(pkg, op, model, x, scratch)
    y = pkg.krylov_time_evolve(x, 0.1, pkg.apply_H, model, kry_m=30)
    assert abs(op.norm(y) - 1.0) <= 1e-12
    assert abs(_energy(pkg, op, model, y, scratch) - e0) <= 1e-10
    assert float((y - x).abs().max()) > 1e-6          # it did move
    # two steps of dt/2 == one step of dt to the accuracy of the 30-dimensional Krylov space (|H| dt ~ 1.5 here)
    h = pkg.krylov_time_evolve(pkg.krylov_time_evolve(x, 0.05, pkg.apply_H, model, kry_m=30), 0.05, pkg.apply_H, model, kry_m=30)
    assert float((h - y).abs().max()) <= 1e-10


def test_config3_kpm_moments_L30_full_size(pkg):
    """BASELINE config 3 at its real size (L=30, nup=15, 1024 moments, one momentum): mu_0 = 1 (1e-12), |mu_n| <= 1, the
    two-moments-per-apply recursion against the library's own one-moment-per-apply mode on the first 64 moments (1e-12 --
    a SELF-COMPARISON of two routes through the same kernels, not an oracle check: the CPU oracle cannot follow L=30), and
    the sum rule of the reconstructed S(q,w) (the reference's own test: rtol 5e-3, test/test_KPM.jl:67-91).
    The oracle comparison at this recursion LENGTH (M = 1024, both routes) is tests/test_gpu_baseline_lengths.py at L=18/20."""
    import torch
    L, M = 30, 1024
    model = pkg.XXZChain(L, nup=L // 2)
    op, psi0 = _random_state(pkg, model, 5)
    scratch = torch.empty_like(psi0)
    E0 = _energy(pkg, op, model, psi0, scratch)
    del scratch
    phi = pkg.Sz_q_vector(model, psi0, float(pkg.momenta(model)[7]))
    n2 = op.norm(phi) ** 2
    phi /= n2 ** 0.5
    del psi0
    a, b = pkg.rescaling_from_bounds(-13.9, 7.6)        # open L=30 chain: E0 = -13.11..., Emax = (L-1)/4 = 7.25
    mu = op.kpm_moments(phi, M, a, b)
    assert abs(mu[0] - 1.0) <= 1e-12
    assert np.abs(mu).max() <= 1.0 + 1e-12
    mu_ref = op.kpm_moments(phi, 64, a, b, doubling=False)
    assert np.abs(mu[:64] - mu_ref).max() <= 1e-12
    omega = np.linspace(b - a - E0, b + a - E0, 6001)
    S = n2 * pkg.kpm_reconstruct(mu * pkg.get_kernel(M, "jackson"), omega, a, b, E0)
    assert (S >= 0).all() and np.isfinite(S).all()
    assert abs(S.sum() * (omega[1] - omega[0]) - n2) <= 5e-3 * n2


def test_config4_chebyshev_L32_full_size(pkg):
    """BASELINE config 4 at its real size (L=32, nup=16, cheb_n=100, explicit bounds, one rank): the evolution is unitary
    (norm to 1e-10; the reference does not renormalise, src/TimeEvolution/Chebyshev.jl:123) and two steps of dt/2 equal one
    step of dt (1e-9 on every element: both are converged, a dt = 4.5 << cheb_n)."""
    import torch
    L = 32
    model = pkg.XXZChain(L, nup=L // 2)
    import gc
    gc.collect()
    torch.cuda.empty_cache()                 # what earlier tests left in torch's caching allocator is not "in use" ...
    pkg.default_context().release_scratch()  # ... nor are the work vectors the library's context keeps between calls
    free, _ = torch.cuda.mem_get_info()
    if free < 8 * 16 * model.N + (4 << 30):
        pytest.skip("not enough device memory")
    op, x = _random_state(pkg, model, 3)
    Eb = (-14.6, 8.0)                                 # open L=32 chain: E0 = -13.99..., Emax = 7.75
    y = pkg.chebyshev_time_evolve(x, 0.4, pkg.apply_H, model, cheb_n=100, Ebounds=Eb)
    assert abs(op.norm(y) - 1.0) <= 1e-10
    h = pkg.chebyshev_time_evolve(x, 0.2, pkg.apply_H, model, cheb_n=100, Ebounds=Eb)
    h = pkg.chebyshev_time_evolve(h, 0.2, pkg.apply_H, model, cheb_n=100, Ebounds=Eb)
    d = 0.0
    for k in range(0, model.N, 1 << 27):              # elementwise difference in slices (no third 9.6 GB temporary)
        d = max(d, float((h[k:k + (1 << 27)] - y[k:k + (1 << 27)]).abs().max()))
    assert d <= 1e-9
    assert d < float(y[:1 << 20].abs().max())         # and the state did move away from psi0
    assert float((y[:1 << 20] - x[:1 << 20]).abs().max()) > 1e-6


@pytest.mark.parametrize("mode", ["class", "range"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_plans_of_L32_every_rank_exact(pkg, world, mode):
    """The plans `bench.py --gpus 2|4|8` runs (L=32, popcount-cell ownership; index ranges are its fall-back): every rank's shard, one after the other on
    this GPU, with the halo filled locally (uniform state) -- H|F> = (L-1)/4 |F> bit for bit on every owned row, in one
    launch and as interior + boundary launches."""
    import torch
    L = 32
    total = 0
    for rank in range(world):
        model = pkg.XXZChain(L, nup=L // 2)
        op = pkg.ShardedOperator(model, rank, world, exchange_fn=lambda o, p, h: h.fill_(1.0), mode=mode)
        psi = op.empty(torch.complex128, "cuda")
        out = op.empty(torch.complex128, "cuda")
        psi.fill_(1.0)
        out.zero_()
        op.apply(out, psi)
        assert bool((out == (L - 1) / 4).all()), (world, rank)
        halo = op.halo(psi)
        halo.fill_(float("nan"))
        out.zero_()
        op._launch(out, psi, halo, 0, part=1)                  # interior tiles never touch the halo
        assert not bool(torch.isnan(out.real).any())
        halo.fill_(1.0)
        op._launch(out, psi, halo, 0, part=2)
        assert bool((out == (L - 1) / 4).all()), (world, rank, "parts")
        total += op.n_local
        del psi, out, halo, op, model
        torch.cuda.empty_cache()
    assert total == 601080390


class _UniformStateComm:
    """A callback communicator that stands in for the peers of ONE rank when every vector of a recursion is uniform (the uniform
    state |F> is an eigenvector of H, so T_n(H)|F>, exp(-iHt)|F> are): the exchange fills the halo with the value the peers
    would have sent -- the first element of what this rank sends --, the all-reduce scales the local sums by N / n_local."""

    def __init__(self, pkg, op, N):
        import ctypes as C
        import torch
        from spindynamics_jl_amd import _lib
        self.counts = {"start": 0, "reduce": 0}
        dev = torch.device("cuda")
        scale = float(N) / float(op.n_local)

        def ex_start(_u, dtype, src_ptr, halo_ptr):
            try:
                self.counts["start"] += 1
                per = 2 if dtype == _lib.SD_C128 else 1
                if op.n_halo:
                    src = _lib.dev_tensor(src_ptr, per, dev)
                    halo = _lib.dev_tensor(halo_ptr, per * op.n_halo, dev).view(-1, per)
                    halo.copy_(src.view(1, per).expand_as(halo))
                return 0
            except Exception:
                return 1

        def allreduce(_u, vals, count):
            self.counts["reduce"] += 1
            for i in range(count):
                vals[i] = vals[i] * scale
            return 0

        self._cbs = _lib.sd_comm_callbacks(None, _lib.EXCHANGE_START_FN(ex_start), _lib.EXCHANGE_WAIT_FN(lambda _u: 0),
                                           _lib.ALLREDUCE_FN(allreduce))
        self.h = C.c_void_p()
        self._pkg = pkg
        pkg.check(pkg.lib().sd_comm_from_callbacks(C.byref(self._cbs), op.rank, op.world, C.byref(self.h)))

    def close(self):
        if self.h:
            self._pkg.lib().sd_comm_destroy(self.h)
            self.h = None


@pytest.mark.parametrize("world", [2, 4, 8])
def test_config4_sharded_recursions_every_rank_exact(pkg, world):
    """BASELINE config 4 (L=32 Chebyshev evolution, state sharded over 2 / 4 / 8 ranks) and the sharded KPM moments, through the C
    recursions (sd_chebyshev_evolve_sharded, sd_kpm_moments_sharded) on EVERY rank of the plans bench.py runs, one rank after
    the other on this GPU: for the uniform state the peers' halo values and the global sums are known (_UniformStateComm), so
    psi(t) = exp(-i (L-1) t / 4) |F> and mu_n = T_n(x) must hold on every owned row of every rank."""
    import ctypes as C
    import torch
    L, M, dt = 32, 16, 0.25
    a, b = L / 2 + 1.0, 0.0
    x = ((L - 1) / 4 - b) / a
    want = np.cos(np.arange(M) * np.arccos(x))
    total = 0
    for rank in range(world):
        model = pkg.XXZChain(L, nup=L // 2)
        op = pkg.ShardedOperator(model, rank, world, exchange_fn=lambda o, p, h: None)
        comm = _UniformStateComm(pkg, op, model.N)
        try:
            phi = op.empty(torch.complex128, "cuda")
            phi.fill_(1.0 / np.sqrt(float(model.N)))
            model.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            mu = np.zeros(M)
            pkg.check(pkg.lib().sd_kpm_moments_sharded(model.ctx.h, model.h, comm.h, phi.data_ptr(), op.n_local, M, a, b,
                                                       mu.ctypes.data_as(C.POINTER(C.c_double))), model.ctx.h)
            assert np.abs(mu - want).max() <= 1e-12, (world, rank, np.abs(mu - want).max())
            psit = op.empty(torch.complex128, "cuda")
            pkg.check(pkg.lib().sd_chebyshev_evolve_sharded(model.ctx.h, model.h, comm.h, phi.data_ptr(), op.n_local, dt, 30, -a, a,
                                                            psit.data_ptr()), model.ctx.h)
            torch.cuda.synchronize()
            ref = np.exp(-1j * (L - 1) / 4 * dt) / np.sqrt(float(model.N))
            assert float((psit - ref).abs().max()) <= 1e-12 * abs(ref), (world, rank)
            total += op.n_local
        finally:
            comm.close()
        del phi, psit, op, model
        torch.cuda.empty_cache()
    assert total == 601080390


def test_config5_recursions_on_one_rank_full_size(pkg):
    """BASELINE config 5 (L=36, 8 ranks): the whole sharded moment recursion sd_kpm_moments_sharded at full size on the largest
    rank (1.26 G owned rows), every piece real -- model, pack, vectors, halo, fused KPM steps, reductions -- except the wire:
    for the uniform state every vector of the recursion is uniform, v_n = T_n(x) |F> with x = ((L-1)/4 - b)/a, so the exchange
    callback can fill the halo with the value the peers would have sent (the first element of what this rank sends) and the
    all-reduce callback can scale the local sums by N / n_local.  The moments must be the Chebyshev polynomials T_n(x)."""
    import ctypes as C
    import torch
    from spindynamics_jl_amd import _lib
    L, world, rank, M = 36, 8, 3, 32
    model = pkg.XXZChain(L, nup=L // 2)
    op = pkg.ShardedOperator(model, rank, world, exchange_fn=lambda o, p, h: None)
    import gc
    gc.collect()
    torch.cuda.empty_cache()                 # what earlier tests left in torch's caching allocator is not "in use" ...
    pkg.default_context().release_scratch()  # ... nor are the work vectors the library's context keeps between calls
    free, _ = torch.cuda.mem_get_info()
    if free < 16 * (5 * op.n_local + op.n_halo + op.n_send) + (6 << 30):
        pytest.skip("not enough device memory")
    dev = torch.device("cuda")
    N, nl = model.N, op.n_local
    phi = op.empty(torch.complex128, dev)
    phi.fill_(1.0 / np.sqrt(float(N)))
    comm = _UniformStateComm(pkg, op, N)
    h, counts = comm.h, comm.counts
    try:
        a, b = L / 2 + 1.0, 0.0
        x = ((L - 1) / 4 - b) / a
        want = np.cos(np.arange(M) * np.arccos(x))
        model.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        for doubling in (True, False):
            model.ctx.set_kpm_doubling(doubling)
            mu = np.zeros(M)
            pkg.check(pkg.lib().sd_kpm_moments_sharded(model.ctx.h, model.h, h, phi.data_ptr(), nl, M, a, b,
                                                       mu.ctypes.data_as(C.POINTER(C.c_double))), model.ctx.h)
            assert np.abs(mu - want).max() <= 1e-12, (doubling, np.abs(mu - want).max())
        assert counts["start"] >= M // 2 + M - 1 and counts["reduce"] >= M
        # the sharded Chebyshev evolution (config 4's recursion at config 5's size) on the same rank: |F> is an eigenvector,
        # psi(t) = exp(-i (L-1)/4 t) |F> on every owned row
        dt, E = 0.3, (L - 1) / 4
        psit = op.empty(torch.complex128, dev)
        pkg.check(pkg.lib().sd_chebyshev_evolve_sharded(model.ctx.h, model.h, h, phi.data_ptr(), nl, dt, 40, -(L / 2 + 1.0), L / 2 + 1.0,
                                                        psit.data_ptr()), model.ctx.h)
        torch.cuda.synchronize()
        ref = np.exp(-1j * E * dt) / np.sqrt(float(N))
        assert float((psit - ref).abs().max()) <= 1e-12 * abs(ref)            # relative to an element (1/sqrt(N) = 1e-5)
        del psit
    finally:
        model.ctx.set_kpm_doubling(True)
        comm.close()
