"""Two-pass probe (no product code).

  P1 = the library's own tiled kernel with the prefix bonds 1..m switched off (DIAG instantiation, SD_DEBUG_SKIP = 256*m), run
       with the recurrence epilogue so that it also reads a partial-result stream (what pass 1 of a real two-pass apply moves:
       psi + partial in, out written = 48 B/row + the far reads that still miss), orbit generators taken from the bonds > m;
  P2 = profiles/probe_p2.hip: the hops on bonds 1..m only, over column super-tiles (32 B/row written fresh, 48 accumulate).

  python profiles/probe_twopass.py L m G[,G..] [reps] [mode: time|counters] [variant: l2|lds]

Prints one JSON line per measurement; correctness: P1(prev = -P2) must equal the full apply to rounding."""
import ctypes as C
import json
import os
import subprocess
import sys
from math import comb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g

L = int(sys.argv[1])
m = int(sys.argv[2])
Gs = [int(x) for x in sys.argv[3].split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
mode = sys.argv[5] if len(sys.argv) > 5 else "time"
variant = sys.argv[6] if len(sys.argv) > 6 else "l2"
nup = L // 2
a = m + 1

so = os.path.join(ROOT, "profiles", "libprobe_p2.so")
src = os.path.join(ROOT, "profiles", "probe_p2.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", so])
probe = C.CDLL(so)
vp, i64 = C.c_void_p, C.c_int64
probe.probe_p2_launch.argtypes = [C.c_int, C.c_int, vp, vp, vp, i64, vp, C.c_int, vp, vp]
probe.probe_p2_lds_launch.argtypes = [C.c_int, C.c_int, vp, vp, vp, i64, vp, vp, vp, C.c_int, vp, C.c_int, vp]
dev = torch.device("cuda")


def block_tables():
    """blk_base[A] = first global row whose sites 1..a read A (site i = bit i-1, the reference's combination order),
    blk_len[k] = rows of a block with k up spins among the first a sites."""
    rest = L - a
    base = np.full(1 << a, 0, np.int64)
    valid = np.zeros(1 << a, bool)
    for A in range(1 << a):
        k = bin(A).count("1")
        r = nup - k
        if r < 0 or r > rest:
            continue
        valid[A] = True
        idx, rr = 0, nup
        for site in range(1, a + 1):
            if rr <= 0:
                break
            if (A >> (site - 1)) & 1:
                rr -= 1
            else:
                idx += comb(L - site, rr - 1)
        base[A] = idx
    blen = {k: comb(rest, nup - k) for k in range(a + 1) if 0 <= nup - k <= rest}
    return base, valid, blen


def build_items(G, base, valid, blen):
    """(A, chunk) work items: the items of one super-tile (filling k, chunk c) are queued back to back on ONE XCD (block b ->
    XCD b % 8), super-tiles dealt round-robin."""
    pop = np.array([bin(A).count("1") for A in range(1 << a)])
    queues = [[] for _ in range(8)]
    s = 0
    for k in sorted(blen):
        As = np.nonzero(valid & (pop == k))[0].astype(np.int64)
        if len(As) == 0:
            continue
        n = blen[k]
        nch = (n + G - 1) // G
        for c0 in range(0, nch, 8 * 64):                      # build in slabs to bound memory
            cs = np.arange(c0, min(nch, c0 + 8 * 64), dtype=np.int64)
            row0 = base[As][None, :] + cs[:, None] * G          # (chunks, nA)
            nn = np.minimum(G, n - cs * G)[:, None] + 0 * row0
            AA = As[None, :] + 0 * row0
            for j, c in enumerate(cs):
                queues[(s + j) % 8].append((row0[j], nn[j], AA[j]))
            s += len(cs)
    qa = []
    for q in queues:
        if q:
            qa.append((np.concatenate([t[0] for t in q]), np.concatenate([t[1] for t in q]), np.concatenate([t[2] for t in q])))
        else:
            qa.append((np.zeros(0, np.int64),) * 3)
    longest = max(len(t[0]) for t in qa)
    items = np.zeros((longest, 8), dtype=[("row0", "<i8"), ("n", "<i4"), ("A", "<u4")])
    for x, (r0, nn, AA) in enumerate(qa):
        items["row0"][: len(r0), x] = r0
        items["n"][: len(r0), x] = nn
        items["A"][: len(r0), x] = AA
    return items.reshape(-1)


def timed(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


pkg = g.load_package()
lib = pkg.lib()
os.environ.pop("SD_DEBUG_SKIP", None)
os.environ.pop("SD_XCD_ORBIT_FROM", None)
full = pkg.XXZChain(L, nup=nup)
N = full.N
op_full = pkg.ShardedOperator(full, 0, 1)
psi = op_full.fill_randn(op_full.empty(torch.complex128, dev), 7)
psi /= float(op_full.norm(psi))
ref = torch.empty_like(psi)
op_full.apply(ref, psi)
ms_full = timed(lambda: op_full.apply(ref, psi), reps)
print(json.dumps({"what": "full apply (product kernel)", "L": L, "N": N, "ms": ms_full}), flush=True)

os.environ["SD_DEBUG_SKIP"] = str(256 * m)
os.environ["SD_XCD_ORBIT_FROM"] = str(m + 1)
p1 = pkg.XXZChain(L, nup=nup)
op1 = pkg.ShardedOperator(p1, 0, 1)
os.environ.pop("SD_DEBUG_SKIP", None)
os.environ.pop("SD_XCD_ORBIT_FROM", None)

base, valid, blen = block_tables()
d_base = torch.from_numpy(base).to(dev)
Jarr = torch.full((12,), 0.5, dtype=torch.float64, device=dev)
S = torch.zeros_like(psi)
out1 = torch.empty_like(psi)
stream = torch.cuda.current_stream().cuda_stream


def run_p1(prev):
    p1.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    pkg.check(lib.sd_apply_sharded_dev(p1.ctx.h, p1.h, 2, out1.data_ptr(), psi.data_ptr(), None, N, 3, 2.0, 0.0, 0.0, 0.0,
                                       prev.data_ptr(), None, 0), p1.ctx.h)


def run_p1_plain():
    op1.apply(out1, psi)


ms_p1_plain = timed(run_p1_plain, reps)
print(json.dumps({"what": "P1 plain epilogue (32 B/row + far)", "m": m, "ms": ms_p1_plain}), flush=True)

for G in Gs:
    if variant == "l2":
        items = build_items(G, base, valid, blen)
        d_items = torch.from_numpy(items.view(np.uint8)).to(dev)
        n_items = len(items)
        assert int(items["n"].sum()) == N, (int(items["n"].sum()), N)

        def run_p2(md, dst=S):
            rc = probe.probe_p2_launch(G, md, psi.data_ptr(), dst.data_ptr(), d_items.data_ptr(), n_items, d_base.data_ptr(), m,
                                       Jarr.data_ptr(), stream)
            assert rc == 0, rc
    else:
        raise SystemExit("lds variant: see probe_twopass_lds.py")
    S.zero_()
    run_p2(0)
    torch.cuda.synchronize()
    negS = -S
    run_p1(negS)
    torch.cuda.synchronize()
    err = float((out1 - ref).abs().max())
    scale = float(ref.abs().max())
    ms_p2_w = timed(lambda: run_p2(0), reps)
    ms_p2_acc = timed(lambda: run_p2(1, out1), reps)
    ms_p1 = timed(lambda: run_p1(negS), reps)
    print(json.dumps({"what": "two-pass probe", "L": L, "N": N, "m": m, "G": G, "items": n_items,
                      "P2_write_only_ms": ms_p2_w, "P2_write_only_GBs_alg32": N * 32 / ms_p2_w / 1e6,
                      "P2_accumulate_ms": ms_p2_acc, "P1_with_partial_stream_ms": ms_p1, "P1_plain_ms": ms_p1_plain,
                      "sum_ms (P2 write-only + P1 with partial)": ms_p2_w + ms_p1, "full_ms": ms_full,
                      "max_abs_err_vs_full": err, "max_abs_ref": scale}), flush=True)
    del d_items
if mode == "counters":
    # a few extra launches of each kernel for the PMC passes (kernel names tell them apart)
    for _ in range(3):
        op_full.apply(ref, psi)
        run_p1(negS)
        run_p2(0)
    torch.cuda.synchronize()
