#!/usr/bin/env python3
"""Per-bond view of the L2 model of profiles/l2_model.py, and a bench for orders of the ORBITS inside an XCD queue.

Same model (byte-capacity LRU per XCD over half tiles, W tiles in flight, a tile's reads spread over its lifetime); prints,
per far bond, how many of its reads hit.  usage: python profiles/l2_order_lab.py L order [order ...]
orders: base (first-seen order, one orbit per XCD turn = what csrc/basis.cpp does), run<k> (k consecutive orbits per XCD),
        block<k> (XCD x gets the x-th contiguous eighth of every block of 8k orbits), hi<k> (orbits sorted by the top k prefix
        sites first -- Gray-reflected -- then first-seen)
"""
import os
import sys
from collections import OrderedDict
from math import comb

L = int(sys.argv[1])
orders = sys.argv[2:]
LS, nup = 12, L // 2
p = L - LS
CAP = int(float(os.environ.get("CAP_MIB", "4")) * (1 << 20))
W = int(os.environ.get("W", "160"))
LO, HI = int(os.environ.get("LO", "513")), int(os.environ.get("HI", "1024"))
FO = int(os.environ.get("FO", "6"))
pc = [bin(i).count("1") for i in range(1 << 16)]


def popc(x):
    return pc[x & 0xFFFF] + pc[x >> 16]


def tile_len(P):
    t = nup - popc(P)
    return comb(LS, t) if 0 <= t <= LS else 0


def n_up_first(P):
    t = nup - popc(P)
    return comb(LS - 1, t - 1) if t >= 1 else 0


tiles = [P for P in range(1 << p) if LO <= tile_len(P) <= HI]


PMAX = p - int(os.environ.get("EXCL", "0"))          # EXCL=2: the last odd bond (p-1, p) is never a generator


def canon(P):
    C0, member, ng, b = P, 0, 0, 1
    while b + 1 <= PMAX and ng < FO:
        if ((P >> (b - 1)) ^ (P >> b)) & 1:
            if not (P >> (b - 1)) & 1:
                C0 ^= 3 << (b - 1)
                member |= 1 << ng
            ng += 1
        b += 2
    return C0, member


def orbits():
    first, groups = {}, []
    for P in tiles:
        C0, mem = canon(P)
        if C0 not in first:
            first[C0] = len(groups)
            groups.append((C0, []))
        groups[first[C0]][1].append((mem, P))
    return [(C0, [P for _, P in sorted(g)]) for C0, g in groups]


def gray_rank(x, k):
    """position of the k-bit value x in the reflected Gray sequence (consecutive positions differ in one bit)"""
    r = 0
    while x:
        r ^= x
        x >>= 1
    return r


def deal(orbs, name):
    q = [[] for _ in range(8)]
    if name == "base" or name.startswith("run"):
        oc = int(name[3:] or 1) if name.startswith("run") else 1
        for o, (_, ps) in enumerate(orbs):
            q[(o // oc) % 8].extend(ps)
    elif name.startswith("block"):
        k = int(name[5:])
        for o, (_, ps) in enumerate(orbs):
            q[(o % (8 * k)) // k].extend(ps)
    elif name.startswith("pair"):
        # orbits that differ in the last prefix site only (straddle partners) back to back on one XCD, k such pairs per turn
        k = int(name[4:] or 1)
        key = {}
        srt = sorted(range(len(orbs)), key=lambda i: (orbs[i][0] & ~(1 << (p - 1)), (orbs[i][0] >> (p - 1)) & 1))
        first = {}
        order = []
        for i in srt:
            order.append(i)
        # keep the first-seen order of the pairs
        grp = {}
        for i in range(len(orbs)):
            grp.setdefault(orbs[i][0] & ~(1 << (p - 1)), []).append(i)
        o = 0
        for g, members in grp.items():
            for i in members:
                q[(o // k) % 8].extend(orbs[i][1])
            o += 1
    elif name.startswith("hi"):
        k = int(name[2:].split("x")[0])
        oc = int(name.split("x")[1]) if "x" in name else 1
        srt = sorted(range(len(orbs)), key=lambda i: (gray_rank(orbs[i][0] >> (p - k), k), i))
        for o, i in enumerate(srt):
            q[(o // oc) % 8].extend(orbs[i][1])
    elif name.startswith("col"):
        # column groups: all tiles that agree on the prefix sites a+1..p (and are in this length class) back to back on one XCD --
        # a group is closed under the prefix bonds 1..a-1 (the structure of the two-pass probe's P2, with whole tiles as segments)
        oc = int(name.split("x")[1]) if "x" in name else 1
        bypop = "p" in name.split("x")[0][3:] or name.startswith("colp")
        a = int("".join(ch for ch in name[3:].split("x")[0] if ch.isdigit()))
        grp = {}
        for P in tiles:
            key = (P >> a, popc(P & ((1 << a) - 1))) if bypop else (P >> a)
            grp.setdefault(key, []).append(P)
        for o, (_, ps) in enumerate(grp.items()):
            q[(o // oc) % 8].extend(ps)
    elif name.startswith("asc") or name.startswith("pasc"):
        # tiles in ascending order of the prefix taken as an integer (site 1 = bit 0 varies fastest: the transpose of the memory
        # order), optionally split by the filling of the whole prefix first (pasc), dealt to the XCDs in runs of K tiles
        K = int(name.lstrip("pasc"))
        srt = sorted(tiles, key=(lambda P: (popc(P), P)) if name.startswith("pasc") else (lambda P: P))
        for i, P in enumerate(srt):
            q[(i // K) % 8].append(P)
    else:
        raise SystemExit("unknown order " + name)
    return q


PAIR_E = int(os.environ.get("PAIR_E", "0"))   # > 0: the two tiles related by prefix bond PAIR_E share a workgroup, that bond is an LDS read


def events(queue):
    ev = []
    for j, P in enumerate(queue):
        t0 = j / W
        reads = [(-1, (P, 0), n_up_first(P) * 16), (-1, (P, 1), (tile_len(P) - n_up_first(P)) * 16)]
        for b in range(1, p):
            if b == PAIR_E:
                continue
            if ((P >> (b - 1)) ^ (P >> b)) & 1:
                Q = P ^ (3 << (b - 1))
                reads.append((b, (Q, 0), n_up_first(Q) * 16))
                reads.append((b, (Q, 1), (tile_len(Q) - n_up_first(Q)) * 16))
        Q = P ^ (1 << (p - 1))
        if tile_len(Q) > 0:
            half = 0 if (P >> (p - 1)) & 1 else 1
            sz = n_up_first(Q) * 16 if half == 0 else (tile_len(Q) - n_up_first(Q)) * 16
            if sz > 0:
                reads.append((p, (Q, half), sz))
        n = len(reads)
        for k, (b, obj, sz) in enumerate(reads):
            ev.append((t0 + (k // 2) / (n // 2 + 1), b, obj, sz))
    return ev


def simulate(queue):
    ev = events(queue)
    ev.sort(key=lambda e: e[0])
    d, used = OrderedDict(), 0
    req = {}
    miss = {}
    for _, b, obj, sz in ev:
        req[b] = req.get(b, 0) + sz
        if obj in d:
            d.move_to_end(obj)
            continue
        miss[b] = miss.get(b, 0) + sz
        d[obj] = sz
        used += sz
        while used > CAP:
            _, s2 = d.popitem(last=False)
            used -= s2
    return req, miss


orbs = orbits()
print(f"L={L} p={p} FO={FO} tiles {len(tiles)} orbits {len(orbs)} (mean {len(tiles) / len(orbs):.1f} tiles)", flush=True)
for name in orders:
    q = deal(orbs, name)
    R, M = {}, {}
    rows = 0
    for x in (0, 3):
        r, m = simulate(q[x])
        rows += sum(tile_len(P) for P in q[x])
        for b in r:
            R[b] = R.get(b, 0) + r[b]
            M[b] = M.get(b, 0) + m.get(b, 0)
    tr, tm = sum(R.values()), sum(M.values())
    print(f"{name:10s} requests {tr / rows:6.1f} B/row  misses {tm / rows:6.1f} B/row  hit {1 - tm / tr:.3f}", flush=True)
    if os.environ.get("STATS"):
        print("   bond: miss B/row (hit rate)  " + "  ".join(f"{b}:{M[b] / rows:.1f}({1 - M[b] / R[b]:.2f})" for b in sorted(R)), flush=True)
