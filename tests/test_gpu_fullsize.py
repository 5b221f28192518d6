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
