#!/usr/bin/env python3
"""Offline model of one XCD's L2 for a given tile processing order (experiment tool, CPU only).

Tiles = prefix configurations P of sites 1..p (p = L - LS); tile P holds C(LS, nup - popcount(P)) rows of 16 B.  A tile reads
its own rows and, for every flippable prefix bond b (bits b-1, b differ), the whole tile P ^ (3 << (b-1)); for the straddling
bond half of tile P ^ (1 << (p-1)).  Blocks are dealt round-robin to 8 XCDs (block i -> XCD i % 8); an XCD keeps W tiles in
flight; a tile's reads are spread evenly over its lifetime.  The L2 is modelled as a byte-capacity LRU over half-tiles.

usage: python profiles/l2_model.py [L] [order ...]      orders: lex chunk orbit greedy ...
"""
import os
import sys
from collections import OrderedDict
from math import comb

import numpy as np

L = int(sys.argv[1]) if len(sys.argv) > 1 else 28
orders = sys.argv[2:] or ["lex", "chunk", "orbit"]
LS, nup = int(os.environ.get("LS", "12")), L // 2
p = L - LS
CAP = int(float(os.environ.get("CAP_MIB", "4")) * (1 << 20))          # bytes of L2 per XCD
W = int(os.environ.get("W", "160"))                # tiles in flight per XCD (32 CUs x 5 workgroups)
LO, HI = int(os.environ.get("LO", "513")), int(os.environ.get("HI", "1024"))     # length class simulated (the 256-thread launch)


def tile_len(P):
    return comb(LS, nup - bin(P).count("1")) if 0 <= nup - bin(P).count("1") <= LS else 0


def n_up_first(P):     # rows whose first suffix site is up
    t = nup - bin(P).count("1")
    return comb(LS - 1, t - 1) if t >= 1 else 0


all_tiles = [P for P in range(1 << p) if LO <= tile_len(P) <= HI]


def canon(P, FO, greedy):
    C0, member, ng, b = P, 0, 0, 1
    while b + 1 <= p and ng < FO:
        if ((P >> (b - 1)) ^ (P >> b)) & 1:
            if not (P >> (b - 1)) & 1:
                C0 ^= 3 << (b - 1)
                member |= 1 << ng
            ng += 1
            b += 2
        else:
            b += 1 if greedy else 2
    return C0, member


def order_orbit(tiles, FO=6, greedy=False, OC=1, gray=False):
    first, keys = {}, []
    for P in tiles:
        C0, mem = canon(P, FO, greedy)
        if C0 not in first:
            first[C0] = len(first)
        if gray:
            mem = mem ^ (mem >> 1)
        keys.append((first[C0], mem))
    idx = sorted(range(len(tiles)), key=lambda k: keys[k])
    q = [[] for _ in range(8)]
    o, k = 0, 0
    while k < len(idx):
        e = k
        while e < len(idx) and keys[idx[e]][0] == keys[idx[k]][0]:
            e += 1
        q[(o // OC) % 8].extend(tiles[i] for i in idx[k:e])
        o += 1
        k = e
    return q


def order_chunk(tiles, CH=32):
    q = [[] for _ in range(8)]
    for c in range(0, len(tiles), CH):
        q[(c // CH) % 8].extend(tiles[c:c + CH])
    return q


def order_lex(tiles):
    return [tiles[x::8] for x in range(8)]


def events(queue):
    ev = []
    life = 1.0
    for j, P in enumerate(queue):
        t0 = j * life / W
        reads = [((P, 0), n_up_first(P) * 16), ((P, 1), (tile_len(P) - n_up_first(P)) * 16)]
        for b in range(1, p):
            if ((P >> (b - 1)) ^ (P >> b)) & 1:
                Q = P ^ (3 << (b - 1))
                reads.append(((Q, 0), n_up_first(Q) * 16))
                reads.append(((Q, 1), (tile_len(Q) - n_up_first(Q)) * 16))
        Q = P ^ (1 << (p - 1))
        if tile_len(Q) > 0:
            # bit p of P up: our down-first rows read Q's up-first rows; bit p down: our up-first rows read Q's down-first rows
            half = 0 if (P >> (p - 1)) & 1 else 1
            sz = n_up_first(Q) * 16 if half == 0 else (tile_len(Q) - n_up_first(Q)) * 16
            if sz > 0:
                reads.append(((Q, half), sz))
        n = len(reads)
        for k, (obj, sz) in enumerate(reads):
            ev.append((t0 + life * (k // 2) / (n // 2 + 1), obj, sz))
    return ev


class LRU:
    def __init__(self, cap):
        self.cap, self.used, self.d = cap, 0, OrderedDict()

    def access(self, obj, sz):
        """True on a hit"""
        if obj in self.d:
            self.d.move_to_end(obj)
            return True
        self.d[obj] = sz
        self.used += sz
        while self.used > self.cap:
            _, s2 = self.d.popitem(last=False)
            self.used -= s2
        return False


def simulate(queue):
    """events of one XCD -> (bytes requested, bytes missed)"""
    ev = events(queue)
    ev.sort(key=lambda e: e[0])
    l2, req, miss = LRU(CAP), 0, 0
    for _, obj, sz in ev:
        req += sz
        if not l2.access(obj, sz):
            miss += sz
    return req, miss


def simulate_all(queues, mall_bytes):
    """all eight XCDs with private L2s in front of one shared memory-side cache -> (requested, L2 misses, MALL misses)"""
    ev = []
    for x, q in enumerate(queues):
        ev.extend((t, x, obj, sz) for (t, obj, sz) in events(q))
    ev.sort(key=lambda e: e[0])
    l2 = [LRU(CAP) for _ in queues]
    mall = LRU(mall_bytes)
    req = m2 = m3 = 0
    for _, x, obj, sz in ev:
        req += sz
        if l2[x].access(obj, sz):
            continue
        m2 += sz
        if not mall.access(obj, sz):
            m3 += sz
    return req, m2, m3


def parse(name):
    if name == "lex":
        return order_lex(all_tiles)
    if name.startswith("chunk"):
        return order_chunk(all_tiles, int(name[5:] or 32))
    if name.startswith("orbit") or name.startswith("greedy") or name.startswith("gray"):
        kind = "greedy" if name.startswith("greedy") else ("gray" if name.startswith("gray") else "orbit")
        rest = name[len(kind):]
        FO, OC = 6, 1
        if rest:
            parts = rest.split("x")
            FO = int(parts[0])
            if len(parts) > 1:
                OC = int(parts[1])
        return order_orbit(all_tiles, FO, kind == "greedy", OC, kind == "gray")
    raise SystemExit("unknown order " + name)


print(f"L={L} p={p} tiles in class: {len(all_tiles)}  rows: {sum(tile_len(P) for P in all_tiles)}")
MALL = float(os.environ.get("MALL_MIB", "0"))          # > 0: simulate all XCDs behind a shared memory-side cache of this size
for name in orders:
    q = parse(name)
    if MALL > 0:
        r, m2, m3 = simulate_all(q, int(MALL * (1 << 20)))
        rows = sum(tile_len(P) for x in q for P in x)
        print(f"{name:12s} read requests {r / rows:6.1f} B/row   L2 misses {m2 / rows:6.1f} B/row   memory-side misses {m3 / rows:6.1f} B/row", flush=True)
        continue
    tot_req = tot_miss = 0
    for x in (0, 3):                       # two of the eight XCDs are enough for a rate
        r, m_ = simulate(q[x])
        tot_req += r
        tot_miss += m_
    rows = sum(tile_len(P) for P in q[0]) + sum(tile_len(P) for P in q[3])
    print(f"{name:12s} read requests {tot_req / rows:6.1f} B/row   L2 misses {tot_miss / rows:6.1f} B/row   hit {1 - tot_miss / tot_req:.3f}", flush=True)
