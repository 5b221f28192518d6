#!/usr/bin/env python3
"""Offline model of one XCD's L2 for the register-group apply kernel (experiment tool, CPU only; companion of l2_model.py).

Work unit = GROUP of 2^f tiles (prefix configurations of sites 1..p, p = L - LS) related by the first f flippable odd
prefix bonds.  A group reads every member tile once (own rows, held in registers) and, per member and per flippable
NON-generator prefix bond, the whole partner tile; plus the straddling half tile.  Groups are dealt to the 8 XCDs by
super-orbits (the next FO2 flippable odd bonds), an XCD keeps W groups in flight, a group's reads are spread over its
lifetime in bond order.  The L2 is a byte-capacity LRU over half-tiles.

usage: LS=10 F=4 W=64 FO2=2 python profiles/l2_model_groups.py [L]
"""
import os
import sys
from collections import OrderedDict
from math import comb

L = int(sys.argv[1]) if len(sys.argv) > 1 else 28
LS, nup = int(os.environ.get("LS", "10")), L // 2
p = L - LS
F = int(os.environ.get("F", "4"))
FO2 = int(os.environ.get("FO2", "2"))
CAP = int(float(os.environ.get("CAP_MIB", "4")) * (1 << 20))
W = int(os.environ.get("W", "64"))
ES = 16


def tile_len(P):
    t = nup - bin(P).count("1")
    return comb(LS, t) if 0 <= t <= LS else 0


def n_up_first(P):
    t = nup - bin(P).count("1")
    return comb(LS - 1, t - 1) if t >= 1 else 0


def gens_of(P, nmax):
    g = []
    b = 1
    while b + 1 <= p and len(g) < nmax:
        if ((P >> (b - 1)) ^ (P >> b)) & 1:
            g.append(b)
        b += 2
    return g


def canon(P, gens):
    C0, mem = P, 0
    for k, b in enumerate(gens):
        if not (P >> (b - 1)) & 1:
            C0 ^= 3 << (b - 1)
            mem |= 1 << k
    return C0, mem


# groups: canonical representative with exactly F generators (fewer: smaller groups, same treatment)
groups = {}
for P in range(1 << p):
    if tile_len(P) == 0:
        continue
    g = gens_of(P, F)
    C0, mem = canon(P, g)
    groups.setdefault((C0, tuple(g)), []).append(P)
glist = sorted(groups.items(), key=lambda kv: kv[0][0])

# super-orbit key: canonical rep under the next FO2 flippable odd bonds after the group's generators
def super_key(C0, g):
    allg = gens_of(C0, F + FO2)
    extra = allg[len(g):]
    S0, mem = canon(C0, extra)
    return S0, mem

first = {}
keyed = []
for (C0, g), members in glist:
    S0, mem = super_key(C0, g)
    if S0 not in first:
        first[S0] = len(first)
    keyed.append((first[S0], mem, C0, g, members))
keyed.sort(key=lambda r: (r[0], r[1]))
queues = [[] for _ in range(8)]
o, k = 0, 0
CH = int(os.environ.get("CH", "0"))     # > 0: runs of CH lexicographically consecutive groups per XCD instead of super-orbits
if CH > 0:
    keyed.sort(key=lambda r: r[2])
    # bit-reversed representative: consecutive keys differ in the LOWEST prefix sites (neighbouring tiles in memory)
    def brev(P):
        return int(format(P, "0%db" % p)[::-1], 2)
    keyed.sort(key=lambda r: brev(r[2]) if os.environ.get("BREV") else r[2])
    for c in range(0, len(keyed), CH):
        queues[(c // CH) % 8].extend(keyed[c:c + CH])
    k = len(keyed)
while k < len(keyed):
    e = k
    while e < len(keyed) and keyed[e][0] == keyed[k][0]:
        e += 1
    queues[o % 8].extend(keyed[k:e])
    o += 1
    k = e


def events(queue):
    ev = []
    life = 1.0
    for j, (_, _, C0, g, members) in enumerate(queue):
        t0 = j * life / W
        gset = set(g)
        for P in members:
            ev.append((t0, (P, 0), n_up_first(P) * ES))
            ev.append((t0, (P, 1), (tile_len(P) - n_up_first(P)) * ES))
        # far reads in bond order, spread over (0.1 .. 0.7) of the lifetime
        for b in range(1, p + 1):
            tb = t0 + life * (0.1 + 0.6 * b / p)
            for P in members:
                if b <= p - 1:
                    if b in gset:
                        continue
                    if ((P >> (b - 1)) ^ (P >> b)) & 1:
                        Q = P ^ (3 << (b - 1))
                        ev.append((tb, (Q, 0), n_up_first(Q) * ES))
                        ev.append((tb, (Q, 1), (tile_len(Q) - n_up_first(Q)) * ES))
                else:
                    Q = P ^ (1 << (p - 1))
                    if tile_len(Q) > 0:
                        half = 0 if (P >> (p - 1)) & 1 else 1
                        sz = n_up_first(Q) * ES if half == 0 else (tile_len(Q) - n_up_first(Q)) * ES
                        if sz > 0:
                            ev.append((tb, (Q, half), sz))
    return ev


class LRU:
    def __init__(self, cap):
        self.cap, self.used, self.d = cap, 0, OrderedDict()

    def access(self, obj, sz):
        if obj in self.d:
            self.d.move_to_end(obj)
            return True
        self.d[obj] = sz
        self.used += sz
        while self.used > self.cap:
            _, s2 = self.d.popitem(last=False)
            self.used -= s2
        return False


tot_req = tot_miss = rows = 0
for x in (0, 3):
    ev = events(queues[x])
    ev.sort(key=lambda e: e[0])
    l2 = LRU(CAP)
    for _, obj, sz in ev:
        if sz == 0:
            continue
        tot_req += sz
        if not l2.access(obj, sz):
            tot_miss += sz
    rows += sum(tile_len(P) for r in queues[x] for P in r[4])
sizes = {}
for r in keyed:
    sizes[len(r[4])] = sizes.get(len(r[4]), 0) + len(r[4])
print(f"L={L} LS={LS} p={p} F={F} FO2={FO2} W={W} groups={len(keyed)} tiles by group size={sorted(sizes.items())}")
print(f"read requests {tot_req / rows:6.1f} B/row   L2 misses {tot_miss / rows:6.1f} B/row (+{ES} write)   hit {1 - tot_miss / tot_req:.3f}")
