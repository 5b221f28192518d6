"""Would LDS orbit groups cut the fabric bytes?  (CPU-only model, extends profiles/l2_model.py)

Idea: one workgroup holds the 2^g tiles of a sub-orbit (g disjoint flippable prefix bonds) in LDS, so that g of a row's ~10 far reads become
LDS reads -- every included bond is flippable for every tile of the group, unlike two more suffix sites (x4 LDS for one read).
Answer (python profiles/l2_model_groups.py 28): requests 148 -> 117 (g=2) -> 103 B/row (g=3), but L2 MISSES 94 -> 91.5 -> 86.5 B/row: the reads
that move into LDS are exactly the ones the orbit order already serves from L2.  3-8 % fewer fabric bytes for 4-8x the LDS per workgroup
(1-2 workgroups per CU).  Not built."""
import os, sys
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "28", "lex"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'l2_model.py')).read().split('print(f"L={L}')[0])

def group_queues(tiles, FO, g):
    """orbit order; consecutive 2^g members (the first g generators) form one workgroup's group"""
    first, keys = {}, []
    for P in tiles:
        C0, mem = canon(P, FO, False)
        if C0 not in first:
            first[C0] = len(first)
        keys.append((first[C0], mem >> g, mem & ((1 << g) - 1)))
    idx = sorted(range(len(tiles)), key=lambda k: keys[k])
    q = [[] for _ in range(8)]
    o, k = 0, 0
    while k < len(idx):
        e = k
        while e < len(idx) and keys[idx[e]][0] == keys[idx[k]][0]:
            e += 1
        # split the orbit into groups
        grp, cur = [], None
        for i in idx[k:e]:
            gk = keys[i][1]
            if gk != cur:
                grp.append([]); cur = gk
            grp[-1].append(tiles[i])
        q[o % 8].extend(grp)
        o += 1
        k = e
    return q

def events_group(queue, Wg):
    ev = []
    life = 1.0
    for j, G in enumerate(queue):
        t0 = j * life / Wg
        S = set(G)
        reads = []
        for P in G:
            reads.append(((P, 0), n_up_first(P) * 16)); reads.append(((P, 1), (tile_len(P) - n_up_first(P)) * 16))
        far = []
        for P in G:
            for b in range(1, p):
                if ((P >> (b - 1)) ^ (P >> b)) & 1:
                    Q = P ^ (3 << (b - 1))
                    if Q in S: continue
                    far.append(((Q, 0), n_up_first(Q) * 16)); far.append(((Q, 1), (tile_len(Q) - n_up_first(Q)) * 16))
            Q = P ^ (1 << (p - 1))
            if tile_len(Q) > 0:
                half = 0 if (P >> (p - 1)) & 1 else 1
                sz = n_up_first(Q) * 16 if half == 0 else (tile_len(Q) - n_up_first(Q)) * 16
                if sz > 0: far.append(((Q, half), sz))
        allr = reads + far
        n = len(allr)
        for k_, (obj, sz) in enumerate(allr):
            ev.append((t0 + life * k_ / (n + 1), obj, sz))
    return ev

def run(q, Wg, label):
    tot_req = tot_miss = rows = 0
    for x in (0, 3):
        ev = events_group(q[x], Wg); ev.sort(key=lambda e: e[0])
        l2 = LRU(CAP)
        for _, obj, sz in ev:
            tot_req += sz
            if not l2.access(obj, sz): tot_miss += sz
        rows += sum(tile_len(P) for G in q[x] for P in G)
    print(f"{label:28s} requests {tot_req/rows:6.1f} B/row  L2 misses {tot_miss/rows:6.1f} B/row  hit {1-tot_miss/tot_req:.3f}  (+16 B/row write)", flush=True)

print(f"L={L} p={p} tiles {len(all_tiles)}")
for FO in (6, 8):
    for g, Wg in ((0, 192), (0, 160), (1, 96), (2, 64), (2, 48), (3, 32), (3, 24)):
        run(group_queues(all_tiles, FO, g), Wg, f"orbit{FO} group 2^{g} W={Wg}")
