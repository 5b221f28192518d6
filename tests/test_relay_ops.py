"""The routed halo exchange as data: `dist.relay_ops` turns a routing plan (`dist.relay_routes`) into every rank's list of sends and
receives in batches -- the list the torch.distributed transport posts (`_RelayExchange`) AND the list the library's own RCCL
communicator executes (`sd_comm_set_exchange_ops`, csrc/comm.cpp).  The RCCL executor cannot meet a second rank on a one-GPU box,
so its input is pinned here on the host: the lists of all ranks are run by a tiny in-process "wire" (per batch, the k-th send from
a to b pairs with the k-th receive of b from a -- NCCL's matching rule inside a group) on vectors that hold global row numbers, and
the halo every rank ends up with must be exactly the rows the direct exchange delivers."""
import os

import numpy as np
import pytest


def _plans(pkg, L, nup, P, monkeypatch, pack):
    monkeypatch.setenv("SD_SUFFIX_BITS", "6")
    monkeypatch.setenv("SD_SHARD_PACK", pack)
    ops = []
    for r in range(P):
        m = pkg.XXZChain(L, nup=nup, ctx=None)
        op = pkg.ShardedOperator(m, r, P, mode="class", exchange_fn=lambda *a: None)
        ops.append(op)
    return ops


def _direct_halo(ops, vecs):
    """what one grouped send / receive of the slab lists delivers (k-th slab r <- q pairs with the k-th slab q -> r)"""
    halos = []
    for op in ops:
        h = np.full(op.n_halo, -1, dtype=np.int64)
        for q, oq in enumerate(ops):
            recvs = [s for s in op.recv_slabs if s[0] == q]
            sends = [s for s in oq.send_slabs if s[0] == op.rank]
            assert [s[2] for s in sends] == [s[2] for s in recvs]
            for (_p, so, cnt, _g), (_p2, ro, _c, _g2) in zip(sends, recvs):
                h[ro - op.n_local:ro - op.n_local + cnt] = vecs[q][so:so + cnt]
        halos.append(h)
    return halos


@pytest.mark.parametrize("L,nup,P,pack,nb", [(16, 8, 3, "1", 3), (16, 8, 4, "0", 4), (20, 10, 8, "0", 4), (20, 10, 8, "1", 2), (18, 8, 5, "0", 1)])
def test_routed_exchange_op_lists_deliver_the_direct_halo(pkg, L, nup, P, pack, nb, monkeypatch):
    from spindynamics_jl_amd import dist as D
    ops = _plans(pkg, L, nup, P, monkeypatch, pack)
    if ops[0].mode != "class":
        pytest.skip("this plan fell back to index ranges")
    # what the send slabs index: the packed send buffer (filled through the pack list) or the vector itself
    rows = [op.model.local_rows() for op in ops]
    vecs = []
    for op, rw in zip(ops, rows):
        if op.packed:
            buf = np.full(op.n_send, -1, dtype=np.int64)
            for a, b, c in zip(*op.model.pack_list()):
                buf[b:b + c] = rw[a:a + c]
            vecs.append(buf)
        else:
            vecs.append(rw.astype(np.int64))
    want = _direct_halo(ops, vecs)
    # the routing plan every rank would compute from the gathered receive lists
    runs = {(int(peer), op.rank): [] for op in ops for (peer, _o, _c, _g) in op.recv_slabs}
    for op in ops:
        for (peer, _o, cnt, _g) in op.recv_slabs:
            runs[(int(peer), op.rank)].append(int(cnt))
    M = {pr: sum(v) for pr, v in runs.items()}
    routes = D.relay_routes(M, 4, 0, force=True)
    assert any(k >= 0 for lst in routes.values() for (k, _u) in lst)
    load = D.relay_link_loads(M, routes)
    assert sum(load.values()) >= sum(M.values())                     # two hops put more bytes on the wire, never fewer
    lists = []
    for op in ops:
        op._relay_runs, op._relay_M = runs, M
        ol, n_relay = D.relay_ops(op, routes, nb)
        assert all(ol[i][0] <= ol[i + 1][0] for i in range(len(ol) - 1))      # batches ascending (what sd_comm_set_exchange_ops demands)
        lists.append((ol, n_relay))
    halo = [np.full(op.n_halo, -1, dtype=np.int64) for op in ops]
    relay = [np.full(max(n, 1), -1, dtype=np.int64) for (_ol, n) in lists]
    for b in range(nb + 1):
        sends, recvs = {}, {}
        for me, (ol, _n) in enumerate(lists):
            for (bb, peer, kind, buf, off, cnt) in ol:
                if bb != b:
                    continue
                assert 0 <= peer < P and peer != me and cnt > 0
                if kind == 0:
                    assert buf in (0, 2)
                    src = vecs[me] if buf == 0 else relay[me]
                    assert off + cnt <= len(src) and (src[off:off + cnt] >= 0).all()      # never forwards what has not arrived
                    sends.setdefault((me, peer), []).append(src[off:off + cnt].copy())
                else:
                    assert buf in (1, 2)
                    recvs.setdefault((peer, me), []).append((me, buf, off, cnt))
        assert set(sends) == set(recvs)
        for pr, lst in sends.items():
            assert [len(x) for x in lst] == [c for (_m, _b, _o, c) in recvs[pr]]        # the k-th send meets the k-th receive
            for data, (me, buf, off, cnt) in zip(lst, recvs[pr]):
                dst = halo[me] if buf == 1 else relay[me]
                assert (dst[off:off + cnt] == -1).all()                                   # every element is delivered once
                dst[off:off + cnt] = data
    for r in range(P):
        assert np.array_equal(halo[r], want[r])


def test_relay_routes_balance_the_links(pkg):
    """A 4-rank ring-like matrix with one heavy pair: the heavy message is spread over idle links, the busiest link drops."""
    from spindynamics_jl_amd import dist as D
    M = {(0, 1): 8000, (1, 0): 8000, (2, 3): 1000, (3, 2): 1000, (1, 2): 1000, (2, 1): 1000}
    routes = D.relay_routes(M, 8, 0)
    load = D.relay_link_loads(M, routes)
    assert max(load.values()) <= 0.6 * 8000
    # two ranks: there is no third party to relay through
    assert D.relay_routes({(0, 1): 100, (1, 0): 100}, 8, 0) == {(0, 1): [(-1, 1)], (1, 0): [(-1, 1)]}
