"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/spindyn.h declares, refuses to
run without a GPU (no CPU fallback), and its host-only entry points (basis queries, small host numerics, shard
plans) agree with the oracle.  No device compute is called here."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "spindyn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sd_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    lib = C.CDLL(pkg.LIB_PATH)
    names = header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/spindyn.h but not exported"
    # the ctypes prototype table covers the header exactly
    assert set(pkg.PROTOTYPES) == set(names)


def header_prototypes():
    """name -> (return kind, [parameter kinds]) parsed from include/spindyn.h; kinds: int, i64, u64, double, ptr, void"""
    src = open(os.path.join(ROOT, "include", "spindyn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct sd_comm_callbacks \{.*?\} sd_comm_callbacks;", "", src, flags=re.S)   # function-pointer members

    def kind(t):
        t = t.strip()
        if "*" in t or "[" in t:
            return "ptr"
        t = re.sub(r"\b(const|unsigned)\b", "", t).split()
        base = t[0] if t else "void"
        if base == "sd_apply_fn":          # a function-pointer typedef
            return "ptr"
        return {"int": "int", "int64_t": "i64", "uint64_t": "u64", "double": "double", "void": "void"}[base]

    out = {}
    for ret, name, params in re.findall(r"([A-Za-z_][A-Za-z0-9_ ]*?[ \*]+)(sd_[A-Za-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ps = [] if params.strip() in ("", "void") else [kind(x) for x in params.split(",")]
        out[name] = (kind(ret), ps)
    return out


def test_ctypes_prototypes_match_the_header_argument_by_argument(pkg):
    """The Python mirror's ctypes table against the C declarations: same number of parameters, and each of the same kind
    (int / int64 / uint64 / double / pointer) -- a drifted signature must not pass (tests/cabi_consumer.c pins the
    types of the calls it makes with the C compiler; this covers every entry point)."""
    protos = header_prototypes()
    assert set(protos) == set(pkg.PROTOTYPES)

    def ckind(t):
        if t is None:
            return "void"
        if t is C.c_int:
            return "int"
        if t is C.c_int64:
            return "i64"
        if t is C.c_uint64:
            return "u64"
        if t is C.c_double:
            return "double"
        return "ptr"            # c_void_p, c_char_p, POINTER(...), function pointers

    for name, (res, args) in pkg.PROTOTYPES.items():
        hret, hargs = protos[name]
        assert ckind(res) == hret, (name, "return", ckind(res), hret)
        assert [ckind(a) for a in args] == hargs, (name, [ckind(a) for a in args], hargs)


def test_no_cpu_fallback(pkg):
    l = pkg.lib()
    if l.sd_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert l.sd_ctx_create(0, C.byref(h)) == 6  # SD_ENODEV
    with pytest.raises(pkg.SpinDynError):
        pkg.XXZChain(4, nup=2)


def test_product_does_not_import_oracle():
    pk = os.path.join(ROOT, "spindynamics.jl_amd")
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the CPU oracle", "").replace("bit-identical to the CPU", ""), f


@pytest.mark.parametrize("L,nup", [(4, 2), (8, 3), (12, 6), (15, 7), (16, 8), (18, 4), (20, 10), (7, 0), (7, 7), (6, None)])
def test_host_basis_matches_oracle(pkg, O, L, nup):
    m = pkg.XXZChain(L, nup=nup, ctx=None)
    r = O.XXZChain(L, nup=nup)
    st = r.states
    assert m.N == r.N
    assert np.array_equal(m.states, st)
    assert np.array_equal(m.rank(st), np.arange(m.N))
    bad = np.array([(1 << L) | 1, (1 << L) - 1 if nup not in (None, L) else 1 << L], dtype=np.uint64)
    assert (m.rank(bad) == -1).all()


def test_large_L_index_closed_form(pkg):
    # 64-bit indices: L=36 nup=18 has 9 075 135 300 rows (> 2^32); first/last states and a rank/unrank round trip
    m = pkg.XXZChain(36, nup=18, ctx=None)
    assert m.N == 9075135300
    assert int(m.states_range(0, 1)[0]) == (1 << 18) - 1
    assert int(m.states_range(m.N - 1, 1)[0]) == ((1 << 18) - 1) << 18
    rng = np.random.default_rng(0)
    idx = rng.integers(0, m.N, 2000)
    st = np.array([m.states_range(int(i), 1)[0] for i in idx], dtype=np.uint64)
    assert all(bin(int(s)).count("1") == 18 for s in st)
    assert np.array_equal(m.rank(st), idx)
    # neighbouring indices are ordered lexicographically over site lists (ascending bit-reversed value)
    s2 = m.states_range(123456789, 3)
    key = [tuple(i for i in range(36) if (int(s) >> i) & 1) for s in s2]
    assert key == sorted(key)


def test_argument_validation(pkg):
    for kw in [dict(L=0, nup=0), dict(L=64, nup=1), dict(L=4, nup=5), dict(L=4, nup=-3)]:
        with pytest.raises(pkg.ArgumentError):
            pkg.XXZChain(kw["L"], nup=kw["nup"], ctx=None)
    with pytest.raises(pkg.ArgumentError):
        pkg.XXZChain(4, nup=2, boundary="twisted", ctx=None)
    with pytest.raises(pkg.ArgumentError):
        pkg.build_model(4, nup=2, hopping=[(1, 5, 1.0)], ctx=None)
    with pytest.raises(pkg.ArgumentError):
        pkg.groundstate(pkg.XXZChain(4, nup=2, ctx=None), method="unknown")
    with pytest.raises(pkg.ArgumentError):
        pkg.time_evolve(pkg.XXZChain(4, nup=2, ctx=None), np.zeros(6, complex), 0.1, method="unknown")
    with pytest.raises(pkg.ArgumentError):
        pkg.dynamical_structure_factor(pkg.XXZChain(4, nup=2, ctx=None), np.zeros(6), [0.0], [0.0], method="unknown")


def test_host_numerics_match_oracle(pkg, O):
    rng = np.random.default_rng(1)
    for n in (1, 2, 7, 40, 100):
        d, e = rng.standard_normal(n), rng.standard_normal(max(n - 1, 0))
        w, z = pkg.symtridiag_eig(d, e)
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        assert np.abs(w - np.linalg.eigvalsh(T)).max() <= 1e-12
        assert np.abs(T @ z - z * w).max() <= 1e-12
    assert np.abs(pkg.chebyshev_coeffs(50, 2.7, -0.3, 0.8) - O.chebyshev_coeffs(50, 2.7, -0.3, 0.8)).max() <= 1e-15
    for kern in ("jackson", "lorentz", "none"):
        assert np.array_equal(pkg.get_kernel(33, kern), O.get_kernel(33, kern))
    assert pkg.rescaling_from_bounds(-3.0, 5.0) == O.rescaling_from_bounds(-3.0, 5.0)
    mu = rng.standard_normal(30) * 0.1
    om = np.linspace(-1, 4, 57)
    assert np.array_equal(pkg.kpm_reconstruct(mu, om, 3.1, 0.2, -2.0), O.kpm_reconstruct(mu, om, 3.1, 0.2, -2.0))
    al, be = rng.standard_normal(12), rng.standard_normal(11)
    for br in ("lorentz", "gauss"):
        got = pkg.spectral_from_tridiagonal(al, be, 1.3, -0.5, om, eta=0.07, broaden=br)
        assert np.abs(got - O.spectral_from_tridiagonal(al, be, 1.3, -0.5, om, eta=0.07, broaden=br)).max() <= 1e-12
    with pytest.raises(pkg.ArgumentError):
        pkg.spectral_from_tridiagonal(al, be, 1.0, 0.0, om, broaden="box")


def test_randn_host_stream(pkg):
    x = np.empty(100000)
    y = np.empty(1000)
    l = pkg.lib()
    dp = C.POINTER(C.c_double)
    assert l.sd_fill_randn_host(x.ctypes.data_as(dp), len(x), 20260821, 0) == 0
    assert l.sd_fill_randn_host(y.ctypes.data_as(dp), len(y), 20260821, 5000) == 0
    assert np.array_equal(x[5000:6000], y)            # keyed by the global element index
    assert abs(x.mean()) < 0.02 and abs(x.std() - 1) < 0.02


def imported_rows(models, slabs, r):
    """Global basis index of every element of rank r's halo buffer, in halo order (both ownership modes)."""
    info = models[r].shard_info()
    out = np.full(int(info.n_halo), -1, dtype=np.int64)
    for (peer, lo, cnt, g) in slabs[r][0]:
        lo -= int(info.n_local)
        if int(info.mode) == 0:
            out[lo:lo + cnt] = np.arange(g, g + cnt)
        elif not int(models[peer].shard_info().packed):
            # class mode, contiguous runs: the k-th slab r receives from the peer is the k-th slab the peer sends to r, a run of
            # the peer's own vector
            k = [x for x in slabs[r][0] if x[0] == peer].index((peer, lo + int(info.n_local), cnt, g))
            (_q, so, scnt, _g) = [x for x in slabs[peer][1] if x[0] == r][k]
            assert scnt == cnt
            out[lo:lo + cnt] = models[peer].local_rows()[so:so + cnt]
        else:   # class mode, packed: the peer packs its tiles in natural order; exactly one message per (peer, r) pair
            rows = models[peer].local_rows()
            src, dst, ln = models[peer].pack_list()
            (soff, scnt), = [(o, c) for (q, o, c, _g) in slabs[peer][1] if q == r]
            assert scnt == cnt
            sel = (dst >= soff) & (dst < soff + scnt)
            packed = np.concatenate([rows[a:a + c] for a, c in zip(src[sel], ln[sel])])
            out[lo:lo + cnt] = packed
    return out


@pytest.mark.parametrize("mode", ["range", "class", "class-direct"])
@pytest.mark.parametrize("L,nup,P", [(12, 6, 2), (14, 7, 3), (16, 8, 2), (16, 8, 8), (18, 9, 4), (17, 5, 5)])
def test_shard_plan_is_consistent(pkg, L, nup, P, mode, monkeypatch):
    monkeypatch.setenv("SD_SUFFIX_BITS", "6")      # many tiles even at small L
    if mode == "class-direct":                     # the form large plans take: contiguous runs sent straight from the vector
        monkeypatch.setenv("SD_SHARD_PACK", "0")
        mode = "class"
    check_shard_plan(pkg, L, nup, P, mode, "open")


@pytest.mark.parametrize("seed", range(int(os.environ.get("SD_PLAN_FUZZ_N", "30"))))
def test_shard_plan_random(pkg, seed, monkeypatch):
    """Seeded random sector, rank count (also more ranks than cells), tile size, ownership mode and boundary condition: the
    owned rows partition the basis, sends pair with receives, and every hop partner of every owned row -- the periodic bond
    included -- is owned or imported.  The multi-GPU path cannot be run on hardware here; this is its plan-level proof."""
    rng = np.random.default_rng(900 + seed)
    L = int(rng.integers(8, 19))
    nup = int(rng.integers(1, L))
    P = int(rng.integers(2, 9))
    monkeypatch.setenv("SD_SUFFIX_BITS", str(int(rng.integers(3, 11))))
    pack = str(rng.choice(["auto", "0", "1"]))          # cell mode: packed send buffer, or runs of the vector itself
    if pack != "auto":
        monkeypatch.setenv("SD_SHARD_PACK", pack)
    check_shard_plan(pkg, L, nup, P, str(rng.choice(["range", "class"])), str(rng.choice(["open", "periodic"])))


def check_shard_plan(pkg, L, nup, P, mode, boundary):
    models = []
    for r in range(P):
        m = pkg.XXZChain(L, nup=nup, ctx=None, boundary=boundary)
        m.set_shard(r, P, mode)
        models.append(m)
    infos = [m.shard_info() for m in models]
    N = models[0].N
    # the owned rows of all ranks partition [0, N)
    rows = [m.local_rows() for m in models]
    assert np.array_equal(np.sort(np.concatenate(rows)), np.arange(N))
    if all(int(i.mode) == 0 for i in infos):        # (class mode falls back to ranges for plans with few prefix sites)
        assert infos[0].row_lo == 0 and infos[-1].row_hi == N
        for a, b in zip(infos[:-1], infos[1:]):
            assert a.row_hi == b.row_lo
        for r in range(P):
            assert np.array_equal(rows[r], np.arange(infos[r].row_lo, infos[r].row_hi))
    slabs = [m.shard_slabs() for m in models]
    for r in range(P):
        recv, send = slabs[r]
        off = infos[r].n_local
        for (peer, lo, cnt, g) in recv:             # halo slabs are packed back to back after the owned rows
            assert lo == off and cnt > 0 and peer != r
            off += cnt
        assert off == infos[r].n_local + infos[r].n_halo
        for q in range(P):                          # what r sends to q is exactly what q receives from r, in order
            s_ = [c for (peer, _, c, _g) in send if peer == q]
            t_ = [c for (peer, _, c, _g) in slabs[q][0] if peer == r]
            assert s_ == t_
        if int(infos[r].mode) == 1 and int(infos[r].packed):
            assert sum(c for (_, _, c, _g) in send) == infos[r].n_send
            assert all(len([1 for (peer, _, _, _g) in send if peer == q]) <= 1 for q in range(P))
        if int(infos[r].mode) == 1 and not int(infos[r].packed):
            assert infos[r].n_send == 0 and all(0 <= o and o + c <= infos[r].n_local for (_, o, c, _g) in send)   # runs of the vector itself
        assert len({int(i.packed) for i in infos}) == 1                      # every rank made the same choice
    # every hop partner of every owned row is either owned or inside the halo
    full = pkg.XXZChain(L, nup=nup, ctx=None, boundary=boundary)
    st = full.states
    bonds = [(i, i + 1) for i in range(1, L)] + ([(L, 1)] if boundary == "periodic" and L > 2 else [])
    for r in range(P):
        have = np.zeros(N, dtype=bool)
        have[rows[r]] = True
        imp = imported_rows(models, slabs, r)
        assert (imp >= 0).all()
        have[imp] = True
        mine = st[rows[r]].astype(np.uint64)
        for (i, j) in bonds:
            bi = (mine >> np.uint64(i - 1)) & np.uint64(1)
            bj = (mine >> np.uint64(j - 1)) & np.uint64(1)
            fl = mine[bi != bj] ^ np.uint64((1 << (i - 1)) | (1 << (j - 1)))
            assert have[full.rank(fl)].all()


def test_class_mode_cuts_halo_volume(pkg, monkeypatch):
    """The popcount-cell ownership imports several times fewer rows than index ranges (here L=24, 4 ranks)."""
    monkeypatch.setenv("SD_SUFFIX_BITS", "8")
    tot = {}
    for mode in ("range", "class"):
        h = 0
        for r in range(4):
            m = pkg.XXZChain(24, nup=12, ctx=None)
            m.set_shard(r, 4, mode)
            i = m.shard_info()
            assert int(i.mode) == (1 if mode == "class" else 0)
            h = max(h, i.n_halo / i.n_local)
        tot[mode] = h
    assert tot["class"] < 0.5 * tot["range"]


def _julia_ccalls(text):
    """(name, return kind, [argument kinds], number of values passed) of every `ccall((:sd_x, libspindyn), Ret, (T...), args...)`"""
    def jkind(t):
        t = t.strip()
        if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
            return "ptr"
        return {"Cint": "int", "Int64": "i64", "UInt64": "u64", "Float64": "double", "Cvoid": "void"}[t]

    def split_top(s):      # split at commas outside (), {} and []
        parts, depth, cur = [], 0, ""
        for ch in s:
            if ch in "({[":
                depth += 1
            elif ch in ")}]":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur)
        return [p.strip() for p in parts]

    calls = []
    for mt in re.finditer(r"ccall\(\(:(sd_[A-Za-z0-9_]+),\s*libspindyn\)\s*,", text):
        i, depth = mt.end(), 1            # scan to the parenthesis that closes this ccall
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        items = split_top(text[mt.end():i - 1])
        ret, types = items[0], items[1]
        assert types.startswith("(") and types.endswith(")"), (mt.group(1), types)
        targs = [jkind(t) for t in split_top(types[1:-1]) if t]
        calls.append((mt.group(1), jkind(ret), targs, len(items) - 2))
    return calls


@pytest.mark.parametrize("rel", ["julia/SpinDynamicsMI.jl", "INTEGRATION.md"])
def test_julia_ccalls_match_the_header(rel):
    """The Julia shim has never been run (no Julia here): at least every ccall in it -- and in INTEGRATION.md's binding
    examples -- must name an exported function, declare the header's parameter kinds in the header's order, and pass as
    many values as it declares."""
    protos = header_prototypes()
    calls = _julia_ccalls(open(os.path.join(ROOT, rel)).read())
    assert len(calls) >= (20 if rel.endswith(".jl") else 3)
    for name, ret, targs, nvals in calls:
        assert name in protos, f"{rel}: ccall of {name}, which include/spindyn.h does not declare"
        hret, hargs = protos[name]
        if name in ("sd_last_error", "sd_status_string"):
            hret = "ptr"                       # const char * parses as a pointer return
        assert ret == hret, (rel, name, "return", ret, hret)
        assert targs == hargs, (rel, name, targs, hargs)
        assert nvals == len(targs), (rel, name, "values passed", nvals, "declared", len(targs))
    # a ccall whose name is not a literal cannot be compiled by Julia
    assert not re.search(r"ccall\(\((?!:)[A-Za-z_]", open(os.path.join(ROOT, rel)).read())


def test_bench_gpus_n_launches_ranks_itself():
    """bench.py --gpus 2 without a launcher must start its own ranks (the driver calls it exactly so for N = 1, and the
    scaling runs must not die on a usage message).  Without a GPU the ranks fail, which the parent reports as its status."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--L", "12", "--steps", "1", "--warmup", "0",
                        "--no-cpu"], cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert "starting 2 ranks" in r.stderr, r.stderr[-2000:]
    assert "launch with torch.distributed.run" not in r.stderr + r.stdout
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0          # the ranks could not create a context: the parent hands that on


def test_julia_shim_block_structure_is_balanced():
    """No Julia here to parse julia/SpinDynamicsMI.jl: at least every block opener (function, if, for, while, struct, module,
    begin, let, try, do, quote, macro) must have its `end`, and (), [], {} must balance -- strings and comments removed, and
    `for` / `if` / `end` inside brackets (comprehensions, a[end]) not counted."""
    text = open(os.path.join(ROOT, "julia", "SpinDynamicsMI.jl")).read()
    text = re.sub(r'"""(?:.|\n)*?"""', '""', text)
    text = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', text)
    text = re.sub(r"#=.*?=#", "", text, flags=re.S)
    text = re.sub(r"#[^\n]*", "", text)
    depth = {"(": 0, "[": 0, "{": 0}
    pairs = {")": "(", "]": "[", "}": "{"}
    opened, closed = 0, 0
    for tok in re.finditer(r"[()\[\]{}]|\b[A-Za-z_][A-Za-z_0-9!]*\b", text):
        t = tok.group(0)
        if t in depth:
            depth[t] += 1
        elif t in pairs:
            depth[pairs[t]] -= 1
            assert depth[pairs[t]] >= 0, "unbalanced %s near offset %d" % (t, tok.start())
        elif depth["("] == 0 and depth["["] == 0 and depth["{"] == 0:
            if t in ("function", "if", "for", "while", "struct", "module", "begin", "let", "try", "do", "quote", "macro"):
                # `mutable struct` is one opener; a one-line `f(x) = ...` definition has no keyword at all
                opened += 1
            elif t == "end":
                closed += 1
    assert depth == {"(": 0, "[": 0, "{": 0}, depth
    assert opened == closed, (opened, closed)


def test_bench_helpers_hash_and_cores(tmp_path, monkeypatch):
    """bench.py: the stamp of the PMC traffic figure follows the code of the apply kernel, not its comments; the CPU baseline's
    thread count is what the process is allowed (affinity mask, cgroup quota), at least one."""
    import bench
    h0 = bench.kernel_source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", h0)
    # the same sources with a comment and blank lines added hash alike; a code change does not
    src = os.path.join(ROOT, bench.TRAFFIC_SOURCES[0])
    text = open(src).read()
    fake_root = tmp_path / "r"
    for rel in bench.TRAFFIC_SOURCES:
        dst = fake_root / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        dst.write_text(open(os.path.join(ROOT, rel)).read())
    monkeypatch.setattr(bench, "ROOT", str(fake_root))
    assert bench.kernel_source_hash() == h0
    (fake_root / bench.TRAFFIC_SOURCES[0]).write_text("// a note\n\n" + text.replace("\n", "\n   \n", 3) + "\n/* trailing\n comment */\n")
    assert bench.kernel_source_hash() == h0
    (fake_root / bench.TRAFFIC_SOURCES[0]).write_text(text.replace("__syncthreads();", "__syncthreads(); __syncthreads();", 1))
    assert bench.kernel_source_hash() != h0
    use, aff, quota = bench.host_cores()
    assert 1 <= use <= aff and (quota is None or quota > 0)
