"""Sharded (multi-GPU layout) apply on ONE GPU: P virtual shards in one process, the halo exchange emulated with
device copies between the shard buffers following exactly the slab plan that dist.py sends over RCCL.  The
concatenated result must be bit-identical to the unsharded apply (SURVEY.md 4 take-away iii)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["range", "class"])
@pytest.mark.parametrize("L,nup,P,ls", [(16, 8, 2, 8), (16, 8, 3, 8), (18, 9, 8, 9), (20, 10, 4, 12), (17, 6, 5, 8), (14, 7, 8, 13),
                                        (20, 10, 8, 8), (22, 11, 4, 10),
                                        # negative ls: tile length classes forced (SD_LEN_CLASSES=2), so the interior and the
                                        # boundary part are each several launches with 64/128/256-thread workgroups
                                        (20, 10, 4, -12), (22, 9, 3, -11), (21, 10, 8, -12)])
def test_virtual_shards_bit_identical(pkg, L, nup, P, ls, mode, monkeypatch):
    if ls < 0:
        monkeypatch.setenv("SD_LEN_CLASSES", "2")
        ls = -ls
    monkeypatch.setenv("SD_SUFFIX_BITS", str(ls))
    check_virtual_shards(pkg, L, nup, P, mode, {})


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("SD_SHARD_FUZZ_N", "24"))))
def test_virtual_shards_random(pkg, seed, monkeypatch):
    """Seeded random sector, rank count, tile size, ownership mode, boundary condition and couplings."""
    rng = np.random.default_rng(500 + seed)
    L = int(rng.integers(10, 21))
    nup = int(rng.integers(max(1, L // 2 - 3), min(L - 1, L // 2 + 3) + 1))
    P = int(rng.integers(2, 9))
    monkeypatch.setenv("SD_SUFFIX_BITS", str(int(rng.integers(4, 13))))
    monkeypatch.setenv("SD_LEN_CLASSES", str(int(rng.choice([1, 2]))))
    pack = str(rng.choice(["auto", "0", "1"]))      # cell mode: packed send buffer or contiguous runs of the vector itself
    if pack != "auto":
        monkeypatch.setenv("SD_SHARD_PACK", pack)
    kw = {"Jxy": float(rng.choice([1.0, 0.7])), "Jz": float(rng.normal()), "hz": float(rng.choice([0.0, 0.3])),
          "boundary": str(rng.choice(["open", "periodic"]))}
    make = None
    if rng.random() < 0.4:
        # the chain followed by a few further bonds anywhere (second neighbours, long range, wraps): the general-bond plan, whose
        # prefix-prefix and mixed partner tiles may be another rank's
        hop = [(i, i + 1, 0.5) for i in range(1, L)]
        zz = [(i, i + 1, float(kw["Jz"])) for i in range(1, L)]
        for _ in range(int(rng.integers(1, 6))):
            i, j = sorted(rng.choice(np.arange(1, L + 1), size=2, replace=False).tolist())
            hop.append((int(i), int(j), float(rng.normal()))); zz.append((int(i), int(j), float(rng.normal())))
        for d in (2,) if rng.random() < 0.5 else ():
            hop += [(i, i + d, 0.2) for i in range(1, L - d + 1)]
        make = lambda: pkg.build_model(L, nup=nup, hopping=hop, zz=zz)      # noqa: E731
    check_virtual_shards(pkg, L, nup, P, str(rng.choice(["range", "class"])), kw, need_interior=False, make=make)


@pytest.mark.parametrize("L,P", [(13, 2), (14, 4), (15, 8), (16, 4), (17, 2)])
@pytest.mark.parametrize("kw", [{}, {"Jxy": 0.7, "Jz": -0.4, "hz": 0.3}])
def test_virtual_shards_full_basis_by_top_bits(pkg, L, P, kw):
    """nup = nothing (src/Hamiltonian.jl:223,255-257: idx = state): the 2^L basis sharded by its top index bits.  A rank owns
    the contiguous rows [r N/P, (r+1) N/P); the bond that straddles the cut imports the matching half of rank r^1, a bond
    between two rank bits that differ the whole vector of rank r ^ (3 << k).  Bit-identical to the unsharded apply."""
    check_virtual_shards(pkg, L, None, P, "range", kw, need_interior=False)


def j1j2_model(pkg, L, nup, J2=0.5, J3=0.0, periodic=False):
    """build_model (src/SpinModel.jl:23-38) with second (and third) neighbour bonds behind the chain bonds"""
    hop, zz = [], []
    for d, J in ((1, 1.0), (2, J2), (3, J3)):
        if J == 0.0:
            continue
        for i in range(1, L + 1):
            j = i + d
            if j > L:
                if not periodic:
                    continue
                j -= L
            hop.append((i, j, 0.5 * J)); zz.append((i, j, J))
    return pkg.build_model(L, nup=nup, hopping=hop, zz=zz)


@pytest.mark.parametrize("mode", ["range", "class"])
@pytest.mark.parametrize("L,nup,P,ls,J3,periodic", [(18, 9, 2, 9, 0.0, False), (20, 10, 4, 10, 0.25, False), (20, 9, 8, 8, 0.0, True),
                                                   (22, 11, 3, 12, 0.3, True)])
def test_virtual_shards_with_second_neighbour_bonds_bit_identical(pkg, L, nup, P, ls, J3, periodic, mode, monkeypatch):
    """The general-bond plan of a sharded model (k_apply_tiled GEN): prefix-prefix partner tiles and the partner tile of a mixed bond may
    live on another rank, i.e. in the halo."""
    monkeypatch.setenv("SD_SUFFIX_BITS", str(ls))
    check_virtual_shards(pkg, L, nup, P, mode, {}, need_interior=False, make=lambda: j1j2_model(pkg, L, nup, 0.5, J3, periodic))


def check_virtual_shards(pkg, L, nup, P, mode, kw, need_interior=True, make=None):
    import torch
    if make is None:
        make = lambda: pkg.XXZChain(L, nup=nup, **kw)       # noqa: E731
    full = make()
    rng = np.random.default_rng(L * P)
    psi = rng.standard_normal(full.N) + 1j * rng.standard_normal(full.N)
    want = np.empty_like(psi)
    pkg.apply_H(want, psi, full)
    ops, bufs = [], []
    for r in range(P):
        m = make()
        op = pkg.ShardedOperator(m, r, P, mode=mode)
        buf = torch.from_numpy(psi[m.local_rows()].copy()).cuda()
        op.halo(buf).fill_(float("nan"))
        ops.append(op); bufs.append(buf)
    # class mode: the owner packs the requested tiles with the HIP pack kernel, the send slabs index that buffer
    outs = [op.pack(b) if op.packed else b for op, b in zip(ops, bufs)]
    # emulate the grouped send/recv: k-th slab r->q pairs with the k-th slab q receives from r
    for q in range(P):
        hq = ops[q].halo(bufs[q])
        for r in range(P):
            sends = [s for s in ops[r].send_slabs if s[0] == q]
            recvs = [s for s in ops[q].recv_slabs if s[0] == r]
            assert len(sends) == len(recvs)
            for (_, so, cnt, _g), (_, ro, cnt2, _g2) in zip(sends, recvs):
                assert cnt == cnt2
                hq[ro - ops[q].n_local:ro - ops[q].n_local + cnt] = outs[r][so:so + cnt]
    got = np.empty_like(psi)
    n_int = 0
    for r in range(P):
        out = torch.full_like(bufs[r], float("nan"))
        if r % 2 == 0:
            ops[r].apply(out, bufs[r], exchange=False)
        else:   # the overlapped form: interior tiles (no halo read) and boundary tiles as separate launches
            ops[r]._launch(out, bufs[r], ops[r].halo(bufs[r]), 0, part=1)
            ops[r]._launch(out, bufs[r], ops[r].halo(bufs[r]), 0, part=2)
        n_int += ops[r].n_interior_tiles
        got[ops[r].model.local_rows()] = out.cpu().numpy()
    assert n_int > 0 or L <= 14 or not need_interior          # (two-tile plans have no interior tile)
    assert np.array_equal(got, want)
    assert sum(o.n_local for o in ops) == full.N


def test_sharded_chebyshev_matches_single_gpu(pkg, O, monkeypatch):
    """config 4 in miniature: Chebyshev evolution with the state sharded over P virtual ranks (sequentially
    stepped in one process is impossible for a collective recursion, so P=1 with the sharded driver and P=1 via the
    C-ABI recursion are compared; the multi-rank exchange itself is covered by the bit-identical apply test above)."""
    import torch
    monkeypatch.setenv("SD_SUFFIX_BITS", "8")
    L, nup = 16, 8
    m = pkg.XXZChain(L, nup=nup)
    r = O.XXZChain(L, nup=nup)
    rng = np.random.default_rng(3)
    psi0 = rng.standard_normal(m.N) + 1j * rng.standard_normal(m.N)
    psi0 /= np.linalg.norm(psi0)
    op = pkg.ShardedOperator(m, 0, 1)
    got = op.chebyshev_time_evolve(torch.from_numpy(psi0).cuda(), 0.3, cheb_n=40, Ebounds=(-8.0, 8.0))
    want = O.chebyshev_time_evolve(r, psi0, 0.3, cheb_n=40, Ebounds=(-8.0, 8.0))
    assert np.abs(got.cpu().numpy() - want).max() <= 1e-14


@pytest.mark.parametrize("mode", ["range", "class"])
def test_sharded_fill_randn_is_sharding_independent(pkg, mode, monkeypatch):
    """bench.py's synthetic psi is keyed by the GLOBAL element index: every sharding produces the same state."""
    import torch
    monkeypatch.setenv("SD_SUFFIX_BITS", "8")
    L, nup, P = 18, 9, 4
    full = pkg.XXZChain(L, nup=nup)
    ref = torch.empty(full.N, dtype=torch.complex128, device="cuda")
    pkg.ShardedOperator(full, 0, 1).fill_randn(ref, 1234)
    ref = ref.cpu().numpy()
    host = np.empty(2 * 64)
    import ctypes as C
    pkg.lib().sd_fill_randn_host(host.ctypes.data_as(C.POINTER(C.c_double)), len(host), 1234, 0)
    assert np.abs(ref[:64].view(np.float64) - host).max() <= 1e-15     # same stream on host and device (libm ulp)
    got = np.empty_like(ref)
    for r in range(P):
        m = pkg.XXZChain(L, nup=nup)
        op = pkg.ShardedOperator(m, r, P, mode=mode)
        x = op.empty(torch.complex128, "cuda")
        op.fill_randn(x, 1234)
        got[m.local_rows()] = x.cpu().numpy()
    assert np.array_equal(got, ref)


def test_sharded_kpm_driver_matches_oracle(pkg, O, monkeypatch):
    """config 5 in miniature (driver logic with one rank; the multi-rank exchange is covered above): S(q,w) via the
    sharded KPM driver == the oracle within BASELINE's 1e-8."""
    import torch
    monkeypatch.setenv("SD_SUFFIX_BITS", "8")
    L, nup = 14, 7
    m = pkg.XXZChain(L, nup=nup)
    r = O.XXZChain(L, nup=nup)
    rng = np.random.default_rng(8)
    psi0 = rng.standard_normal(m.N)
    psi0 /= np.linalg.norm(psi0)
    a, b = O.rescaling_from_bounds(-L / 2, L / 2)
    q = pkg.momenta(m)[:4]
    omega = np.arange(-3.0, 5.0, 0.1)
    op = pkg.ShardedOperator(m, 0, 1)
    S = op.kpm_sqw(torch.from_numpy(psi0).cuda(), q, omega, a, b, kpm_m=96)
    S2 = O.kpm_sqw(r, psi0, q, omega, a, b, kpm_m=96)
    assert np.abs(S - S2).max() <= 1e-8 * max(1.0, np.abs(S2).max())
    # the moment loop itself, doubling (default) and the reference's one-moment-per-apply form, odd and even M
    phi = np.asarray(O.Sz_q_vector(r, psi0, float(q[1])))
    phi /= np.linalg.norm(phi)
    tphi = torch.from_numpy(phi).cuda()
    for M in (2, 3, 8, 33):
        want = O.compute_chebyshev_moments(r, phi, M, a, b)
        assert np.abs(op.kpm_moments(tphi, M, a, b) - want).max() <= 1e-13
        assert np.abs(op.kpm_moments(tphi, M, a, b, doubling=False) - want).max() <= 1e-13


def test_kpm_q_replicas_two_processes_one_gpu():
    """kpm_sqw_replicas (momenta dealt over the ranks, no data-path communication): two real processes sharing this GPU,
    gloo for the final all-reduce; every rank must reproduce the single-process S(q,w) bit for bit."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "profiles", "replicas_rehearsal.py")],
                       cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("replicas == single: True") == 2


def test_sharded_lanczos_bounds_and_kpm_without_explicit_rescaling(pkg, O, monkeypatch):
    """Sharded lanczos_extremal / estimate_energy_bounds (driver logic with one rank) against the C-ABI recursion on the same
    generated start vector, and kpm_sqw estimating (a, b) itself as the reference does (src/KPM_Sqw.jl:212-214)."""
    import torch
    monkeypatch.setenv("SD_SUFFIX_BITS", "8")
    L, nup = 14, 7
    m = pkg.XXZChain(L, nup=nup, Jz=0.8)
    r = O.XXZChain(L, nup=nup, Jz=0.8)
    op = pkg.ShardedOperator(m, 0, 1)
    start = op.fill_randn(op.empty(torch.complex128, "cuda"), 5)
    lo, hi = op.lanczos_extremal(lanc_m=60, psi0=start)
    lo2, hi2 = pkg.lanczos_extremal(pkg.apply_H, m, lanc_m=60, psi0=start.cpu().numpy())
    assert abs(lo - lo2) < 1e-10 and abs(hi - hi2) < 1e-10            # converged extremal Ritz values
    nlo, nhi = op.lanczos_extremal(lanc_m=60, psi0=start, negate=True)
    assert abs(nhi + lo) < 1e-9 and abs(nlo + hi) < 1e-9               # spectrum of -H
    Emin, Emax = op.estimate_energy_bounds(lanc_m=60)
    w = np.linalg.eigvalsh(np.asarray([O.apply_H(r, e) for e in np.eye(m.N)]).T.real) if m.N <= 4000 else None
    if w is not None:
        assert abs(Emin - w[0]) < 1e-8 and abs(Emax - w[-1]) < 1e-6
    with pytest.raises(pkg.ZeroNormError):
        op.lanczos_extremal(lanc_m=5, psi0=torch.zeros(m.N, dtype=torch.complex128, device="cuda"))
    psi0 = np.random.default_rng(8).standard_normal(m.N)
    psi0 /= np.linalg.norm(psi0)
    q, omega = pkg.momenta(m)[:2], np.arange(-2.0, 4.0, 0.2)
    S = op.kpm_sqw(torch.from_numpy(psi0).cuda(), q, omega, kpm_m=64)
    a, b = pkg.rescaling_from_bounds(Emin, Emax) if hasattr(pkg, "rescaling_from_bounds") else O.rescaling_from_bounds(Emin, Emax)
    S2 = O.kpm_sqw(r, psi0, q, omega, a, b, kpm_m=64)
    assert np.isfinite(S).all() and np.abs(S - S2).max() <= 1e-6 * max(1.0, np.abs(S2).max())   # bounds agree to 1e-8, not to the bit


@pytest.mark.parametrize("relay", [False, True])
def test_sharded_drivers_three_processes_one_gpu(relay):
    """Every sharded driver (apply with the overlapped exchange, Chebyshev pairs, KPM moments both ways, Lanczos bounds,
    S(q,w)) with three real processes sharing this GPU, in both ownership modes; gloo carries the halo messages through the
    host (dist.py stages them), which exercises the same request / wait / part-1 / part-2 sequence as RCCL does.
    relay=True: the halo messages of the popcount-cell mode take the two-hop routes of dist.relay_routes (SD_RELAY=1)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "profiles", "sharded_rehearsal.py")],
                       cwd=root, capture_output=True, text=True, timeout=400,
                       env=dict(os.environ, **({"SD_RELAY": "2", "SD_RELAY_MIN": "0", "SD_RELAY_CHUNKS": "4"} if relay else {})))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("sharded == single: True") == 6
    if relay:
        assert "relayed elements per exchange:" in r.stdout


def test_rccl_communicator_selftest_one_rank(pkg):
    """sd_comm_rccl_create on this one GPU (nranks = 1): librccl is found at run time, ncclCommInitRank succeeds, and the
    entry points the sharded recursions use (ncclAllReduce on the compute stream, grouped ncclSend/ncclRecv on the
    communication stream fenced by events) move the right bytes.  What needs peers is covered by the gloo rehearsals."""
    import ctypes as C
    import torch
    ctx = pkg.default_context()
    idbuf = (C.c_ubyte * 128)()
    pkg.check(pkg.lib().sd_comm_rccl_unique_id(idbuf))
    h = C.c_void_p()
    pkg.check(pkg.lib().sd_comm_rccl_create(ctx.h, 0, 1, idbuf, C.byref(h)), ctx.h)
    try:
        pkg.check(pkg.lib().sd_comm_selftest(ctx.h, h), ctx.h)
        # a one-rank communicator with an unsharded model: the sharded entry points reduce to the single-GPU recursion
        m = pkg.XXZChain(14, nup=7)
        psi = torch.randn(m.N, dtype=torch.complex128, device="cuda")
        out, out2 = torch.empty_like(psi), torch.empty_like(psi)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        pkg.check(pkg.lib().sd_apply_sharded(ctx.h, m.h, h, 2, out.data_ptr(), psi.data_ptr(), m.N, 1), ctx.h)
        pkg.check(pkg.lib().sd_apply_dev(ctx.h, m.h, 2, out2.data_ptr(), psi.data_ptr(), m.N), ctx.h)
        torch.cuda.synchronize()
        assert torch.equal(out, out2)
        # the routed exchange list is validated when it is installed: empty = back to the default; a peer outside the communicator,
        # a send into the halo, a receive into the vector, a relay range beyond the relay buffer, descending batches are refused
        from spindynamics_jl_amd import _lib
        X = _lib.sd_xop
        assert pkg.lib().sd_comm_set_exchange_ops(h, None, 0, 0) == _lib.SD_OK
        for bad in ([X(0, 0, 0, 0, 0, 8)], [X(0, 1, 0, 0, 0, 8)], [X(0, -1, 1, 1, 0, 8)]):
            arr = (X * len(bad))(*bad)
            assert pkg.lib().sd_comm_set_exchange_ops(h, arr, len(bad), 0) == _lib.SD_EARG        # one rank: every peer is out of range or itself
    finally:
        pkg.lib().sd_comm_destroy(h)


def test_bench_starts_its_own_ranks_two_processes_one_gpu():
    """`python bench.py --gpus 2` from a plain invocation (no launcher, WORLD_SIZE unset): the parent starts two ranks under
    torch.distributed.run, relays rank 0's single JSON line and exits 0.  gloo backend: both ranks share this GPU (the
    N > 1 control flow -- self-checks, overlapped exchange, per-rank breakdown -- is the one RCCL runs)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SD_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--L", "24", "--steps", "5",
                        "--warmup", "1"], cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["value"] > 0
    assert "bit for bit" in line["selfcheck"] and "FELL BACK" not in line["selfcheck"]
    per_rank = line["config"]["per_rank_ms"]
    assert [p["rank"] for p in per_rank] == [0, 1]
    for p in per_rank:
        assert set(("pack", "interior", "exchange_wait", "boundary", "bytes_sent_per_peer")) <= set(p)
    assert sum(p["rows_owned"] for p in per_rank) == 2704156          # C(24, 12)
    # the headline is the library's own sharded step (sd_apply_sharded); with gloo (two ranks on one GPU: RCCL cannot run) its
    # communicator is torch.distributed behind callbacks, checked bit for bit against the Python-issued step before timing
    lp = line["config"]["library_path"]
    assert lp["communicator"] == "torch" and lp["status"].startswith("ok") and lp["ms_per_step"] > 0
    assert line["config"]["path"].startswith("library") and lp["python_path_ms_per_step"] > 0


def test_bench_tries_the_relays_after_the_direct_exchange_four_processes_one_gpu():
    """`bench.py --gpus 4` measures the direct exchange first and then tries the two-hop relays (forced here: at four ranks no
    route pays by itself): routed halo and H psi must equal the direct ones to the bit before the routed step is timed, and it
    becomes the headline only when faster.  gloo backend, four ranks on this GPU; either outcome is a green run, the line must
    say which."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SD_RELAY")}
    env.update(SD_BENCH_BACKEND="gloo", SD_BENCH_RELAY_TRIAL="2", SD_RELAY_MIN="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--L", "24", "--steps", "4",
                        "--warmup", "1"], cwd=root, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["value"] > 0
    trial = line["config"]["relay_trial"]
    assert trial["status"].startswith("ok"), trial
    assert trial["ms_per_step"] > 0 and trial["direct_ms_per_step"] > 0
    routed_headline = "relays" in line["config"]["path"]
    assert routed_headline == (trial["ms_per_step"] < trial["direct_ms_per_step"])
    assert ("two-hop" in line["config"]["halo_routing"]) == routed_headline
    assert abs(line["ms_per_step"] - (trial["ms_per_step"] if routed_headline else trial["direct_ms_per_step"])) < 1e-9


def test_bench_relay_trial_that_hangs_leaves_the_direct_line_and_a_green_exit():
    """The safety net of the relay trial: its watchdog (fired at once here) prints the line of the direct exchange measured before
    it, marked as such, and every rank leaves with status 0 -- an optional leg cannot cost the measurement."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SD_RELAY")}
    env.update(SD_BENCH_BACKEND="gloo", SD_BENCH_RELAY_TRIAL="2", SD_RELAY_MIN="0", SD_BENCH_RELAY_TIMEOUT="0.000001")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--L", "22", "--steps", "3",
                        "--warmup", "1"], cwd=root, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 3 and line["value"] > 0
    assert line["config"]["relay_trial"]["status"].startswith("timeout")
    assert "relays" not in line["config"]["path"] and line["config"]["library_path"]["status"].startswith("ok")


def test_bench_rccl_leg_runs_with_one_rank():
    """The library's own RCCL communicator with a 1-rank NCCL process group on this GPU: creation from a broadcast id, ring
    self-test, sd_apply_sharded on it compared bit for bit with the Python-issued step, a timed loop -- everything of the path
    bench.py --gpus N makes its headline, except what needs a peer."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "profiles", "bench_rccl_leg_one_rank.py"), "20"], cwd=root,
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    assert "rccl_one_rank_ms:" in r.stdout
