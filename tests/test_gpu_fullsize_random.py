"""Bit-exact row samples of H|psi> on a RANDOM vector at BASELINE's full sizes, where the C oracle cannot follow (its states[] and
hash map alone are 23 GB at L=32).

The reference's row loop (src/Hamiltonian.jl:211-273) needs, for one row, nothing but the row's configuration and psi at <= L-1 partner
rows.  For a few ten thousand sampled rows -- the first and last rows, rows around tile boundaries, random rows -- this test restates
that loop in numpy on its own: configuration from the row index by the combinadic unranking of the lexicographic-combination order
(src/Basis.jl:37-53), the diagonal as the reference's sequential sum (fields i = 1..L, then zz in list order), the
partner index by ranking the flipped configuration, psi at the partner rows gathered from the device vector, `value += J * psi'` in
list order.  Nothing of the library takes part in the expected values except the random vector itself; every sampled row must match the
device result to the bit (real and imaginary parts as separate IEEE doubles; the reference has no fused multiply-add, and the kernel's
fma for power-of-two amplitudes is exact)."""
from math import comb

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def unrank(rows, L, nup):
    """configurations (uint64, site i = bit i-1) of 0-based rows in the reference order: site 1 up first"""
    idx = rows.astype(np.int64).copy()
    r = np.full(idx.shape, nup, dtype=np.int64)
    s = np.zeros(idx.shape, dtype=np.uint64)
    table = np.array([[comb(n, k) for k in range(nup + 1)] for n in range(L + 1)], dtype=np.int64)
    for k in range(1, L + 1):
        c = np.where(r > 0, table[L - k, np.maximum(r - 1, 0)], 0)
        up = (r > 0) & (idx < c)
        s |= np.where(up, np.uint64(1) << np.uint64(k - 1), np.uint64(0))
        idx = np.where(up | (r == 0), idx, idx - c)
        r = r - up
    assert (r == 0).all()
    return s


def rank(s, L, nup):
    idx = np.zeros(s.shape, dtype=np.int64)
    r = np.full(s.shape, nup, dtype=np.int64)
    table = np.array([[comb(n, k) for k in range(nup + 1)] for n in range(L + 1)], dtype=np.int64)
    for k in range(1, L + 1):
        bit = ((s >> np.uint64(k - 1)) & np.uint64(1)).astype(bool)
        add = np.where(~bit & (r > 0), table[L - k, np.maximum(r - 1, 0)], 0)
        idx += add
        r = r - bit
    assert (r == 0).all()
    return idx


def sample_rows(model, n_random, seed):
    N = model.N
    rng = np.random.default_rng(seed)
    parts = [np.arange(0, min(N, 2048)), np.arange(max(0, N - 2048), N), rng.integers(0, N, n_random)]
    gb = []
    if model.device_path == "tiled":                      # tile boundaries of the device plan: rows on both sides of a few hundred of them
        _lb, gb, _ln = model.local_tiles()
    pick = rng.choice(len(gb), size=min(len(gb), 300), replace=False) if len(gb) else []     # (the per-row path has no tiles)
    for t in pick:
        parts.append(np.arange(max(0, gb[t] - 3), min(N, gb[t] + 3)))
    return np.unique(np.concatenate(parts).astype(np.int64))


CASES = [
    # L, nup, kwargs, dtype, random rows
    (32, 16, {}, "c128", 20000),                                                     # the headline workload
    (32, 16, {}, "f64", 20000),
    (30, 15, {"Jxy": 0.7, "Jz": -0.4, "hz": 0.3}, "c128", 12000),                    # amplitudes that are no powers of two, fields
    (28, 14, {"boundary": "periodic"}, "c128", 12000),                               # the wrap bond (L, 1)
    (28, 12, {"Jz": 0.37, "boundary": "periodic"}, "f64", 12000),
    (34, 17, {}, "f64", 12000),                                                      # rows past 2^31
    (28, 14, {"ranges": ((1, 1.0), (2, 0.5))}, "c128", 12000),                       # J1-J2: the general-bond plan (streams, packed table, second LDS image)
    (28, 13, {"ranges": ((1, 1.0), (2, 0.5), (3, -0.3)), "boundary": "periodic"}, "f64", 12000),
    (26, 13, {"ranges": tuple((d, 1.0 / d ** 2) for d in range(1, 26))}, "c128", 6000),   # every pair, 1/r^2: 325 bonds
    (40, 10, {}, "c128", 8000),                                                      # 2^28 prefixes: the per-row path (closed-form chain partners), N = 8.5e8
    (44, 8, {"Jxy": 0.8, "Jz": 1.3, "hz": 0.1}, "f64", 8000),
    (39, 7, {"boundary": "periodic"}, "c128", 8000),                                 # ... with the wrap bond through the rank walk
    (36, 18, {}, "f64", 8000),                                                       # config 5's sector whole on one GPU: N = 9.08e9 rows, past 2^32
]


def bond_lists(L, kw):
    """(hopping, zz, field) of the model: XXZChain's lists (src/SpinModel.jl:63-90) or, with kw["ranges"], a build_model with bonds
    (i, i + d) of strength J for every (d, J) of the ranges, distance by distance (periodic: wrapped)"""
    hop, zz = [], []
    if "lists" in kw:
        return kw["lists"]
    if "ranges" in kw:
        for d, J in kw["ranges"]:
            for i in range(1, L + 1):
                j = i + d
                if j > L:
                    if kw.get("boundary") != "periodic":
                        continue
                    j -= L
                hop.append((i, j, 0.5 * J)); zz.append((i, j, J))
        return hop, zz, [0.0] * L
    Jxy, Jz, hz = kw.get("Jxy", 1.0), kw.get("Jz", 1.0), kw.get("hz", 0.0)
    bonds = [(i, i + 1) for i in range(1, L)]
    if kw.get("boundary") == "periodic" and L > 2:
        bonds.append((L, 1))                                                           # src/SpinModel.jl:74-78
    return [(i, j, Jxy / 2) for (i, j) in bonds], [(i, j, Jz) for (i, j) in bonds], [hz] * L      # src/SpinModel.jl:71


def make_model(pkg, L, nup, kw):
    if "ranges" in kw or "lists" in kw:
        hop, zz, f = bond_lists(L, kw)
        return pkg.build_model(L, nup=nup, hopping=hop, zz=zz, onsite_field=np.asarray(f, dtype=float))
    return pkg.XXZChain(L, nup=nup, **kw)


def reference_rows(psi, rows, L, nup, kw, dtype):
    """(own, H psi) at `rows` by the reference's row loop in numpy, real and imaginary parts as separate float64 arrays
    (imaginary None for Float64); psi is the device vector, read only at the rows and their partner rows."""
    import torch
    s = unrank(rows, L, nup)
    assert (rank(s, L, nup) == rows).all()                                             # the two restatements agree with each other
    hops, zzs, field = bond_lists(L, kw)
    sz = lambda site: np.where((s >> np.uint64(site - 1)) & np.uint64(1), 0.5, -0.5)   # noqa: E731

    d = np.zeros(len(rows))
    for i in range(1, L + 1):                                                          # src/Hamiltonian.jl:228-233
        d = d + field[i - 1] * sz(i)
    for (i, j, Jz) in zzs:                                                             # :235-241
        d = d + (Jz * sz(i)) * sz(j)

    def gather(idx):
        v = psi[torch.from_numpy(idx).cuda()].cpu().numpy()
        return (v.real.copy(), v.imag.copy()) if dtype == "c128" else (v.copy(), None)

    own_re, own_im = gather(rows)
    val_re = d * own_re                                                                # :243
    val_im = d * own_im if own_im is not None else None
    for (i, j, hop) in hops:                                                           # :248-267
        bi = (s >> np.uint64(i - 1)) & np.uint64(1)
        bj = (s >> np.uint64(j - 1)) & np.uint64(1)
        fl = bi != bj
        if not fl.any():
            continue
        s2 = s[fl] ^ ((np.uint64(1) << np.uint64(i - 1)) | (np.uint64(1) << np.uint64(j - 1)))
        p_re, p_im = gather(rank(s2, L, nup))
        val_re[fl] = val_re[fl] + hop * p_re
        if val_im is not None:
            val_im[fl] = val_im[fl] + hop * p_im
    return own_re, own_im, val_re, val_im


def random_vector(pkg, model, N, dtype, seed):
    import torch
    tdt = torch.complex128 if dtype == "c128" else torch.float64
    psi = torch.empty(N, dtype=tdt, device="cuda")
    model.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    pkg.check(pkg.lib().sd_fill_randn_dev(model.ctx.h, psi.data_ptr(), (2 if dtype == "c128" else 1) * N, seed, 0), model.ctx.h)
    return psi


@pytest.mark.parametrize("L,nup,kw,dtype,n_random", CASES)
def test_random_vector_sampled_rows_bit_exact_full_size(pkg, L, nup, kw, dtype, n_random):
    import torch
    model = make_model(pkg, L, nup, kw)
    N = model.N
    esz = 16 if dtype == "c128" else 8
    import gc
    gc.collect()
    torch.cuda.empty_cache()                 # what earlier tests left in torch's caching allocator is not "in use" ...
    pkg.default_context().release_scratch()  # ... nor are the work vectors the library's context keeps between calls
    free, _ = torch.cuda.mem_get_info()
    if free < 2 * esz * N + (3 << 30):
        pytest.skip("not enough device memory")
    psi = random_vector(pkg, model, N, dtype, 20260821)
    out = torch.empty_like(psi)
    pkg.apply_H(out, psi, model)
    torch.cuda.synchronize()
    rows = sample_rows(model, n_random, seed=L * 1000 + nup)
    _own_re, _own_im, val_re, val_im = reference_rows(psi, rows, L, nup, kw, dtype)
    got = out[torch.from_numpy(rows).cuda()].cpu().numpy()
    if dtype == "c128":
        assert np.array_equal(got.real, val_re) and np.array_equal(got.imag, val_im)
    else:
        assert np.array_equal(got, val_re)
    assert np.isfinite(val_re).all() and np.abs(val_re).max() > 0.1                    # a real comparison, not zeros against zeros


def test_fused_rescale_and_chebyshev_term_sampled_rows_bit_exact_at_L32(pkg):
    """The fused stores of the recursions on the headline workload, random vectors, sampled rows, to the bit:
    apply_rescaled_H! = (H psi - b psi) / a with a true division (src/Hamiltonian.jl:286-301), and one Chebyshev term
    phi_next = 2 H~ phi_curr - phi_prev; psi_t += c phi_next (src/TimeEvolution/Chebyshev.jl:110-121) with Julia's complex product."""
    import torch
    L, nup, kw = 32, 16, {}
    model = pkg.XXZChain(L, nup=nup)
    N = model.N
    import gc
    gc.collect()
    torch.cuda.empty_cache()                 # what earlier tests left in torch's caching allocator is not "in use" ...
    pkg.default_context().release_scratch()  # ... nor are the work vectors the library's context keeps between calls
    free, _ = torch.cuda.mem_get_info()
    if free < 5 * 16 * N + (3 << 30):
        pytest.skip("not enough device memory")
    a, b, c = 8.3172, -2.71, complex(0.3125, -0.77)
    phi = random_vector(pkg, model, N, "c128", 11)
    prev = random_vector(pkg, model, N, "c128", 12)
    psit = random_vector(pkg, model, N, "c128", 13)
    rows = sample_rows(model, 12000, seed=77)
    tr = torch.from_numpy(rows).cuda()
    own_re, own_im, val_re, val_im = reference_rows(phi, rows, L, nup, kw, "c128")
    prev_s, psit_s = prev[tr].cpu().numpy(), psit[tr].cpu().numpy()
    out = torch.empty_like(phi)

    pkg.apply_rescaled_H(out, phi, pkg.apply_H, model, a, b)
    got = out[tr].cpu().numpy()
    r_re, r_im = (val_re - b * own_re) / a, (val_im - b * own_im) / a
    assert np.array_equal(got.real, r_re) and np.array_equal(got.imag, r_im)

    pkg.cheb_step(out, phi, prev, psit, model, a, b, c)
    torch.cuda.synchronize()
    o_re, o_im = 2.0 * r_re - prev_s.real, 2.0 * r_im - prev_s.imag
    got = out[tr].cpu().numpy()
    assert np.array_equal(got.real, o_re) and np.array_equal(got.imag, o_im)
    t_re = psit_s.real + (c.real * o_re - c.imag * o_im)
    t_im = psit_s.imag + (c.real * o_im + c.imag * o_re)
    got = psit[tr].cpu().numpy()
    assert np.array_equal(got.real, t_re) and np.array_equal(got.imag, t_im)


@pytest.mark.parametrize("world,mode,rank", [(8, "class", 2), (8, "class", 5), (8, "range", 3), (4, "class", 1), (2, "class", 1)])
def test_one_rank_of_the_sharded_L32_apply_on_a_random_vector_equals_the_unsharded_rows(pkg, world, mode, rank):
    """BASELINE's sharded workload (L=32 over 2 / 4 / 8 ranks) on ONE GPU, one receiving rank at a time, on the random vector of
    bench.py: every peer's shard is materialised in turn (its owned rows of the global vector, packed by the HIP pack kernel where
    the plan packs), its send slabs are copied into the receiver's halo exactly where a real exchange would put them, and the
    receiver's sharded apply (interior launch, then boundary launch) must equal the rows it owns of the unsharded H psi, bit for
    bit, all of them.  The unsharded apply itself is pinned row by row above."""
    import torch
    L, nup = 32, 16
    full = pkg.XXZChain(L, nup=nup)
    N = full.N
    import gc
    gc.collect()
    torch.cuda.empty_cache()                 # what earlier tests left in torch's caching allocator is not "in use" ...
    pkg.default_context().release_scratch()  # ... nor are the work vectors the library's context keeps between calls
    free, _ = torch.cuda.mem_get_info()
    if free < 2 * 16 * N + (12 << 30):
        pytest.skip("not enough device memory")
    fop = pkg.ShardedOperator(full, 0, 1)
    x = fop.fill_randn(fop.empty(torch.complex128, "cuda"), 20260821)
    y = torch.empty_like(x)
    pkg.apply_H(y, x, full)

    m = pkg.XXZChain(L, nup=nup)
    op = pkg.ShardedOperator(m, rank, world, mode=mode)
    rows = torch.from_numpy(m.local_rows()).cuda()
    mine = x[rows]
    halo = op.halo(mine)
    halo.fill_(float("nan"))
    peers = sorted({s[0] for s in op.recv_slabs})
    assert peers and rank not in peers
    for q in peers:
        mq = pkg.XXZChain(L, nup=nup)
        oq = pkg.ShardedOperator(mq, q, world, mode=mode)
        xq = x[torch.from_numpy(mq.local_rows()).cuda()]
        src = oq.pack(xq) if oq.packed else xq
        sends = [s for s in oq.send_slabs if s[0] == rank]
        recvs = [s for s in op.recv_slabs if s[0] == q]
        assert len(sends) == len(recvs)
        for (_p, so, cnt, _g), (_p2, ro, cnt2, _g2) in zip(sends, recvs):
            assert cnt == cnt2
            halo[ro - op.n_local:ro - op.n_local + cnt] = src[so:so + cnt]
        del xq, src, oq, mq
    assert not bool(torch.isnan(halo.real).any())                       # every halo element was delivered
    out = torch.full_like(mine, float("nan"))
    op._launch(out, mine, halo, 0, part=1)
    op._launch(out, mine, halo, 0, part=2)
    torch.cuda.synchronize()
    assert bool(torch.equal(out, y[rows]))
    assert op.n_halo > 0 and (op.n_interior_tiles > 0 or mode == "range")      # (an inner index range of eight has no interior tile)


def _random_big_model(rng):
    """a sector of 10^6..3*10^8 rows with random filling, couplings and extra bonds: whatever plan the library picks for it"""
    from math import comb as C
    while True:
        L = int(rng.integers(22, 37))
        nup = int(rng.integers(2, L - 1))
        if 1e6 <= C(L, nup) <= 3e8:
            break
    kind = int(rng.integers(0, 4))
    Jz = float(rng.choice([1.0, 0.5, float(rng.normal())]))
    hz = float(rng.choice([0.0, 0.0, 0.25]))
    hop = [(i, i + 1, 0.5) for i in range(1, L)] if kind != 1 else [(i, i + 1, float(rng.normal())) for i in range(1, L)]
    zz = [(i, i + 1, Jz) for i in range(1, L)]
    if kind == 2:                                   # periodic chain
        hop.append((L, 1, hop[0][2])); zz.append((L, 1, Jz))
    if kind == 3:                                   # further bonds: a second-neighbour ladder and a few random pairs
        hop += [(i, i + 2, 0.2) for i in range(1, L - 1)]
        zz += [(i, i + 2, 0.4) for i in range(1, L - 1)]
        for _ in range(int(rng.integers(0, 4))):
            i, j = sorted(rng.choice(np.arange(1, L + 1), size=2, replace=False).tolist())
            hop.append((int(i), int(j), float(rng.normal()))); zz.append((int(i), int(j), float(rng.normal())))
    return L, nup, {"lists": (hop, zz, [hz] * L)}


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SD_BIG_FUZZ_N", "10"))))
def test_random_large_sectors_sampled_rows_bit_exact(pkg, seed):
    """Seeded random sectors of 10^6 .. 3*10^8 rows (any filling: tiled plans with long and with short tiles, the per-row path, the
    general-bond plan), random couplings, both element types: sampled rows against the numpy row loop, to the bit."""
    import torch
    rng = np.random.default_rng(9000 + seed)
    L, nup, kw = _random_big_model(rng)
    dtype = str(rng.choice(["c128", "f64"]))
    model = make_model(pkg, L, nup, kw)
    psi = random_vector(pkg, model, model.N, dtype, 77 + seed)
    out = torch.empty_like(psi)
    pkg.apply_H(out, psi, model)
    torch.cuda.synchronize()
    rows = sample_rows(model, 3000, seed=seed)
    _own_re, _own_im, val_re, val_im = reference_rows(psi, rows, L, nup, kw, dtype)
    got = out[torch.from_numpy(rows).cuda()].cpu().numpy()
    if dtype == "c128":
        assert np.array_equal(got.real, val_re) and np.array_equal(got.imag, val_im), (L, nup, model.device_path)
    else:
        assert np.array_equal(got, val_re), (L, nup, model.device_path)
