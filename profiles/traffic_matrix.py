#!/usr/bin/env python3
"""Per-link halo traffic of the sharded apply, straight from the plans the library builds (host only, no GPU needed):
for every rank of a P-way sharding the recv slabs of sd_model_shard_slabs give bytes[receiver][owner] per apply.  xGMI is
point to point (7 links x ~153 GB/s per GPU, MI355X_MICROARCH.md / prompt), so a pair's bytes ride ONE link; the step
time is bounded by the busiest link direction, not by the aggregate.

usage: python profiles/traffic_matrix.py [L ...]       (default 32 36; writes markdown to stdout)
Projected step time = max(exchange, interior) + boundary with the measured per-row kernel cost (19.4 ps/row at L=32) and a
link rate band of 60-77 GB/s per direction (RCCL point-to-point over one xGMI link; not measured on this pool: one GPU per box).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
Ls = [int(x) for x in sys.argv[1:]] or [32, 36]
ES = 16                      # ComplexF64
PS_PER_ROW = 19.4e-12        # measured: 11.68 ms / 601 M rows (profiles/bench_r02.json)
LINK = (60e9, 77e9)

print("# Halo traffic per apply from the shard plans (`profiles/traffic_matrix.py`)\n")
for L in Ls:
    for mode in ("class", "range"):
        for P in (2, 4, 8):
            M = np.zeros((P, P))
            nloc, nint_rows = np.zeros(P), np.zeros(P)
            for r in range(P):
                m = pkg.XXZChain(L, nup=L // 2, ctx=None)
                m.set_shard(r, P, mode)
                info = m.shard_info()
                recv, _send = m.shard_slabs()
                nloc[r] = int(info.n_local)
                for (peer, _off, cnt, _g) in recv:
                    M[r, peer] += cnt * ES
                lb, gb, ln = m.local_tiles()
                del m
            imp = M.sum(axis=1)
            busiest = M.max()
            pairs = int((M > 0).sum())
            t_link = [busiest / x * 1e3 for x in LINK[::-1]]
            t_rank_in = [imp.max() / (x * min(7, P - 1)) * 1e3 for x in LINK[::-1]]
            t_local = nloc.max() * PS_PER_ROW * 1e3
            print(f"## L={L} P={P} mode={mode}\n")
            print(f"rows per rank {nloc.min() / 1e6:.1f}-{nloc.max() / 1e6:.1f} M; imported rows per owned row (worst rank) "
                  f"{(imp / ES / np.maximum(nloc, 1)).max():.3f}; imports per rank {imp.min() / 1e9:.2f}-{imp.max() / 1e9:.2f} GB; "
                  f"{pairs} directed pairs of {P * (P - 1)} carry data; busiest link direction {busiest / 1e9:.3f} GB "
                  f"= {t_link[0]:.1f}-{t_link[1]:.1f} ms at 77-60 GB/s; if the worst rank's imports were spread evenly over "
                  f"its {min(7, P - 1)} links: {t_rank_in[0]:.1f}-{t_rank_in[1]:.1f} ms; local kernel work {t_local:.1f} ms "
                  f"(single GPU: {sum(nloc) * PS_PER_ROW * 1e3:.1f} ms)\n")
            print("GB received (row = receiver, column = owner):\n")
            print("| | " + " | ".join(str(q) for q in range(P)) + " |")
            print("|---|" + "---|" * P)
            for r in range(P):
                print(f"| {r} | " + " | ".join(f"{M[r, q] / 1e9:.3f}" if M[r, q] else "·" for q in range(P)) + " |")
            print("", flush=True)
            if mode == "class" and P >= 3:
                # the two-hop routing the exchange uses with SD_RELAY=1 (dist.relay_routes): pipelined, so the busiest LINK counts
                dmod = sys.modules[pkg.__name__ + ".dist"]
                Mel = {(q, r): int(M[r, q] // ES) for r in range(P) for q in range(P) if M[r, q] > 0}
                routes = dmod.relay_routes(Mel, 8, 65536)
                load = dmod.relay_link_loads(Mel, routes)
                b = max(load.values()) * ES / 1e9
                wire = sum(load.values()) / max(1, sum(Mel.values()))
                relayed = any(k >= 0 for lst in routes.values() for (k, _u) in lst)
                floor = max(max(M.sum(axis=1)), max(M.sum(axis=0))) / min(7, P - 1) / 1e9
                print(f"two-hop relays (`SD_RELAY=1`, 8 routing units per message, pipelined in 4 batches): busiest link {busiest / 1e9:.3f} GB -> "
                      f"{b:.3f} GB ({'kept direct: no gain' if not relayed else f'{(1 - b * 1e9 / busiest) * 100:.0f} % less'}; "
                      f"{b / 77e9 * 1e3 * 1e9:.1f}-{b / 60e9 * 1e3 * 1e9:.1f} ms at 77-60 GB/s); bytes on the wire x{wire:.2f}; "
                      f"floor if the busiest rank's traffic were spread evenly over its {min(7, P - 1)} links: {floor:.3f} GB\n", flush=True)
