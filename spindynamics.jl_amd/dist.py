"""Sharding of the Hilbert-space vector over GPUs (one process per GPU).

The reference has no distributed layer; this is the new capability BASELINE.json's north star asks for.  A rank owns a set
of whole tiles and its vectors hold exactly the n_local owned elements.  Before each apply the psi values of hop partners
that live on other ranks are imported into ONE halo buffer per operator (n_halo elements, shared by every vector of a
recursion).  A hop on a prefix bond maps whole tiles onto whole tiles, so whole tiles travel, never per-element index lists.

Two ownership modes (csrc/basis.cpp):
  "range": contiguous basis-index ranges; a few contiguous slabs of psi itself are sent per peer.
  "class" (default): ownership by the up-spin counts of nested site blocks (nested bisections for 2 / 4 / 8 ranks, runs of popcount
      cells otherwise); only the bonds at block ends can cross a cut, so 3-4x fewer rows are imported (L=32: 0.15 / 0.41 / 0.73 per
      owned row at 2 / 4 / 8 ranks instead of 0.52 / 1.50 / 2.50).  The tiles a peer needs form long contiguous runs of the owner's
      vector and travel straight from it, one send per run (`packed` False); plans whose runs are short pack them into a send
      buffer and ONE message travels per (owner, receiver) pair (`packed` True).
Transport: the library's own RCCL communicator (default under an NCCL process group, csrc/comm.cpp) or torch.distributed P2P ops
behind callbacks (gloo in the CPU tests and rehearsals); optional pipelined two-hop relays (relay_routes / relay_ops).
"""
import ctypes as C
import os

from . import _lib
from ._lib import check, lib


class _StagedRequests:
    """wait() for a host-staged exchange: complete the gloo requests, then copy the received halo to the device."""

    def __init__(self, reqs, dst_dev, dst_host):
        self.reqs, self.dst_dev, self.dst_host = reqs, dst_dev, dst_host

    def wait(self):
        for r in self.reqs:
            r.wait()
        self.dst_dev.copy_(self.dst_host)


_dev_tensor = _lib.dev_tensor


def relay_routes(M, chunks=8, min_elems=0, force=False):
    """Two-hop routing of halo messages through peers whose links would idle (opt-in, SD_RELAY=1).
    M[(owner, receiver)] = elements of the pair's message.  xGMI is point to point: a pair's bytes ride ONE link, and the
    busiest (owner, receiver) pair of a sharded apply carries two to three times the mean.  Every message of at least min_elems
    elements is cut into `chunks` routing units; each unit takes the path -- direct, or owner -> k -> receiver -- that keeps the
    BUSIEST LINK smallest, where a link's load is everything it carries on either hop: the exchange is pipelined (`_RelayExchange`:
    the second hop of one slice travels while the first hop of the next is under way), so what bounds it is the most loaded link,
    not the sum of two rounds.  Largest units first, ties by rank index: every rank computes the same plan.  The plan is kept only
    if its busiest link carries less than 0.9 of the busiest direct message (force=True, SD_RELAY=2, keeps it regardless: tests).
    Returns {(owner, receiver): [(via or -1, units), ...]} with the units of one path merged (they sum to the message's unit
    count: `chunks`, or 1 for a message too small to split)."""
    ranks = sorted({a for pr in M for a in pr})
    load = {}
    units = []
    for (o, r), n in M.items():
        c = chunks if n >= max(min_elems, chunks) else 1
        for k in range(c):
            units.append((n * (k + 1) // c - n * k // c, o, r, c))
    units.sort(key=lambda t: (-t[0], t[1], t[2]))
    count = {}
    for sz, o, r, c in units:
        if sz == 0:
            continue
        top = max(load.values(), default=0)
        d = load.get((o, r), 0) + sz
        best = (max(top, d), d, 0, -1)                      # (busiest link afterwards, this path's busiest link, hops - 1, via)
        if c > 1:
            for k in ranks:
                if k == o or k == r:
                    continue
                a, b = load.get((o, k), 0) + sz, load.get((k, r), 0) + sz
                cand = (max(top, a, b), max(a, b), 1, k)
                if cand[:3] < best[:3]:
                    best = cand
        k = best[3]
        if k < 0:
            load[(o, r)] = load.get((o, r), 0) + sz
        else:
            load[(o, k)] = load.get((o, k), 0) + sz
            load[(k, r)] = load.get((k, r), 0) + sz
        count.setdefault((o, r), {})
        count[(o, r)][k] = count[(o, r)].get(k, 0) + 1
    if not force and max(load.values(), default=0) >= 0.9 * max(M.values(), default=0):
        return {pr: [(-1, 1)] for pr, n in M.items() if n > 0}         # not worth a second hop: everything direct
    return {pr: [(k, per_via[k]) for k in sorted(per_via)] for pr, per_via in count.items()}


def relay_link_loads(M, routes):
    """{(a, b): elements link a -> b carries per exchange} under `routes` (both hops), for reports and tests."""
    load = {}
    for (o, r), lst in routes.items():
        tot = sum(u for _k, u in lst)
        lo = 0
        for (k, u) in lst:
            sz = M[(o, r)] * (lo + u) // tot - M[(o, r)] * lo // tot
            lo += u
            for link in (((o, r),) if k < 0 else ((o, k), (k, r))):
                load[link] = load.get(link, 0) + sz
    return load


def relay_ops(op, routes, nb=4):
    """The operations of ONE rank for a halo exchange over `routes`, pipelined in nb + 1 batches: a list of
    (batch, peer, kind, buf, offset, count) with kind 0 = send / 1 = receive, buf 0 = the vector (or packed send buffer) / 1 = the
    halo buffer / 2 = this rank's relay buffer, offsets and counts in elements -- and the relay buffer's size.  Every message is
    cut into nb slices, each slice into the sub-ranges its paths carry (in proportion to their routing units); batch b holds the
    direct pieces and the first hops of slice b and the second hops of slice b - 1.  All ranks walk the same canonical order
    (batch, sorted pairs, paths in route order, segments in order), which is what pairs a send with its receive.  A message is
    a list of segments at its owner -- one for a packed send buffer, one per contiguous run when runs travel straight from psi --
    and one contiguous range of the halo at its receiver; a relay receives the owner's segments back to back and forwards them as
    one piece.  Shared by the torch.distributed transport (_RelayExchange) and the library's RCCL communicator
    (sd_comm_set_exchange_ops)."""
    me, nl = op.rank, op.n_local
    runs = {}                                   # receiver -> [(offset in the vector, count)] of what I own and it needs, in order
    for (peer, off, cnt, _g) in op.send_slabs:
        runs.setdefault(peer, []).append((off, cnt))
    recv_at, recv_runs = {}, {}                 # owner -> first halo element of its message; its segments (halo offset, count)
    for (peer, off, cnt, _g) in op.recv_slabs:
        recv_at.setdefault(peer, off - nl)
        recv_runs.setdefault(peer, []).append((off - nl, cnt))

    def cut(lst, lo, hi):
        """[(offset, count)] of the elements [lo, hi) of a message laid out as the runs lst = [(offset, count), ...]"""
        out, pos = [], 0
        for (off, cnt) in lst:
            a, b = max(lo, pos), min(hi, pos + cnt)
            if a < b:
                out.append((off + a - pos, b - a))
            pos += cnt
        return out

    M = op._relay_M
    ops, slot = [], 0
    second = []                                 # second hops, posted one batch later
    for b in range(nb):
        for (o, r) in sorted(routes):
            n = M[(o, r)]
            units = sum(u for _k, u in routes[(o, r)])
            s_lo, s_hi = n * b // nb, n * (b + 1) // nb
            u0 = 0
            for (k, u) in routes[(o, r)]:
                lo = s_lo + (s_hi - s_lo) * u0 // units
                hi = s_lo + (s_hi - s_lo) * (u0 + u) // units
                u0 += u
                if hi <= lo:
                    continue
                if k < 0:
                    if me == o:
                        ops += [(b, r, 0, 0, off, cnt) for (off, cnt) in cut(runs[r], lo, hi)]
                    if me == r:                 # the same segments as the owner sends: its runs, cut at the same places
                        ops += [(b, o, 1, 1, off, cnt) for (off, cnt) in cut(recv_runs[o], lo, hi)]
                    continue
                if me == o:
                    ops += [(b, k, 0, 0, off, cnt) for (off, cnt) in cut(runs[r], lo, hi)]
                if me == k:
                    pos = slot
                    for cnt in op._relay_runs_of(o, r, lo, hi):
                        ops.append((b, o, 1, 2, pos, cnt))
                        pos += cnt
                    second.append((b + 1, r, 0, 2, slot, hi - lo))
                    slot = pos
                if me == r:
                    second.append((b + 1, k, 1, 1, recv_at[o] + lo, hi - lo))
    # merge: within a batch, first hops / direct pieces of slice b come before the second hops of slice b - 1 on EVERY rank
    allops = sorted(range(len(ops)), key=lambda i: ops[i][0])
    out = []
    for b in range(nb + 1):
        out += [ops[i] for i in allops if ops[i][0] == b]
        out += [x for x in second if x[0] == b]
    return out, slot


class _RelayExchange:
    """One halo exchange over the routes of relay_routes on torch.distributed: the batches of relay_ops as batch_isend_irecv
    calls.  On RCCL the batches are consecutive groups on one stream, so a link's second-hop traffic overlaps the first hops of
    the next slice and an exchange costs about what its busiest link carries, not the sum of two rounds; host-staged transports
    (gloo) complete a batch before the next one is posted."""

    def __init__(self, op, routes, src, dst, per, group, finish=None, nb=4):
        import torch
        import torch.distributed as dist
        ops, n_relay = relay_ops(op, routes, nb)
        relay = torch.empty(max(n_relay, 1) * per, dtype=src.dtype, device=src.device)
        bufs = (src, dst, relay)
        self._batches = [[] for _ in range(nb + 1)]
        for (b, peer, kind, buf, off, cnt) in ops:
            t = bufs[buf][off * per:(off + cnt) * per]
            self._batches[b].append(dist.P2POp(dist.isend if kind == 0 else dist.irecv, t, peer, group))
        self._relay, self._finish, self._dist = relay, finish, dist
        self._staged = dist.get_backend(group) != "nccl"
        self._reqs = []
        self._next = 0
        self._post()                                # NCCL: every batch is queued now (consecutive groups on one stream); gloo: batch 0

    def _post(self):
        while self._next < len(self._batches):
            ops = self._batches[self._next]
            self._next += 1
            if ops:
                self._reqs.extend(self._dist.batch_isend_irecv(ops))
            if self._staged:
                return                              # host-staged transports complete a batch before the next one is posted

    def wait(self):
        while True:
            for r in self._reqs:
                r.wait()
            self._reqs = []
            if self._next >= len(self._batches):
                break
            self._post()
        if self._finish is not None:
            self._finish()


class TorchComm:
    """sd_comm (include/spindyn.h) whose callbacks move the bytes with torch.distributed: what the C recursion-level entry
    points (sd_*_sharded) call for the halo exchange of each apply and for the sum of the scalars of each reduction.
    Backend "nccl" (= RCCL over xGMI): device to device; "gloo" (rehearsals with several ranks on one GPU): staged
    through the host."""

    def __init__(self, op, device, group=None):
        import torch.distributed as dist
        self.op, self.device, self.group = op, device, group
        self.backend = dist.get_backend(group)
        self._reqs = []
        self._err = None
        self._cbs = _lib.sd_comm_callbacks(None, _lib.EXCHANGE_START_FN(self._start), _lib.EXCHANGE_WAIT_FN(self._wait),
                                           _lib.ALLREDUCE_FN(self._allreduce))     # keeps the trampolines alive
        self.h = C.c_void_p()
        check(lib().sd_comm_from_callbacks(C.byref(self._cbs), op.rank, op.world, C.byref(self.h)))

    def _start(self, _user, dtype, src_ptr, halo_ptr):
        try:
            import torch
            import torch.distributed as dist
            op = self.op
            per = 2 if dtype == _lib.SD_C128 else 1
            n_src = op.n_send if op.packed else op.n_local
            src = _dev_tensor(src_ptr, max(n_src, 1) * per, self.device)
            dst = _dev_tensor(halo_ptr, max(op.n_halo, 1) * per, self.device)
            staged = self.backend == "gloo"
            if staged:
                torch.cuda.current_stream(self.device).synchronize()      # the pack kernel of the C side has finished
                dst_dev, src, dst = dst, src.cpu(), torch.empty(dst.shape, dtype=dst.dtype)
            nl = op.n_local
            routes = op.relay_plan(self.group)
            if routes is not None:
                fin = (lambda: dst_dev.copy_(dst)) if staged else None
                self._reqs = [_RelayExchange(op, routes, src, dst, per, self.group, fin, int(os.environ.get("SD_RELAY_BATCHES", "4")))]
                return 0
            ops = []
            for (peer, off, cnt, _g) in op.recv_slabs:
                ops.append(dist.P2POp(dist.irecv, dst[(off - nl) * per:(off - nl + cnt) * per], peer, self.group))
            for (peer, off, cnt, _g) in op.send_slabs:
                ops.append(dist.P2POp(dist.isend, src[off * per:(off + cnt) * per], peer, self.group))
            reqs = dist.batch_isend_irecv(ops) if ops else []
            self._reqs = [_StagedRequests(reqs, dst_dev, dst)] if (staged and reqs) else reqs
            return 0
        except Exception as e:      # an exception must not unwind through the C frames
            self._err = e
            return 1

    def _wait(self, _user):
        try:
            for r in self._reqs:
                r.wait()
            self._reqs = []
            return 0
        except Exception as e:
            self._err = e
            return 1

    def _allreduce(self, _user, vals, count):
        try:
            import numpy as np
            import torch
            import torch.distributed as dist
            a = np.ctypeslib.as_array(vals, shape=(count,))
            t = torch.from_numpy(a.copy())
            if self.backend == "nccl":
                t = t.to(self.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            a[:] = t.cpu().numpy()
            return 0
        except Exception as e:
            self._err = e
            return 1

    def close(self):
        if self.h:
            lib().sd_comm_destroy(self.h)
            self.h = None


class RcclComm:
    """sd_comm on RCCL itself (sd_comm_rccl_create): grouped ncclSend/ncclRecv beside the interior tiles and ncclAllReduce
    of the device scalars, all queued on HIP streams by the C side -- a recursion step never touches the host.  The
    128-byte id is made on rank 0 and broadcast with torch.distributed (any transport would do)."""

    def __init__(self, op, device, group=None):
        import torch
        import torch.distributed as dist
        idbuf = (C.c_ubyte * 128)()
        if op.rank == 0:
            check(lib().sd_comm_rccl_unique_id(idbuf))
        t = torch.tensor(list(idbuf), dtype=torch.uint8)
        if dist.get_backend(group) == "nccl":
            t = t.to(device)
        dist.broadcast(t, 0, group=group)
        idbuf = (C.c_ubyte * 128)(*[int(v) for v in t.cpu().tolist()])
        self.h = C.c_void_p()
        m = op.model
        check(lib().sd_comm_rccl_create(m.ctx.h, op.rank, op.world, idbuf, C.byref(self.h)), m.ctx.h)
        self._err = None
        self.routed = False

    def set_routes(self, op, routes, nb=4):
        """Install the pipelined two-hop exchange of `routes` (dist.relay_routes) in the library's communicator
        (sd_comm_set_exchange_ops); routes None: back to one grouped send / receive of the slab lists."""
        if routes is None:
            check(lib().sd_comm_set_exchange_ops(self.h, None, 0, 0))
            self.routed = False
            return
        ops, n_relay = relay_ops(op, routes, nb)
        arr = (_lib.sd_xop * max(len(ops), 1))()
        for i, (b, peer, kind, buf, off, cnt) in enumerate(ops):
            arr[i] = _lib.sd_xop(int(b), int(peer), int(kind), int(buf), int(off), int(cnt))
        check(lib().sd_comm_set_exchange_ops(self.h, arr, len(ops), int(n_relay)))
        self.routed = True

    def close(self):
        if self.h:
            lib().sd_comm_destroy(self.h)
            self.h = None


class ShardedOperator:
    def __init__(self, model, rank, world, exchange_fn=None, mode=None, pack_fn=None, reduce_fn=None):
        self.model = model
        self.rank, self.world = rank, world
        self._exchange_fn = exchange_fn   # tests inject an emulated exchange for virtual shards in one process
        self._pack_fn = pack_fn           # CPU tests inject a numpy pack (the product packs with a HIP kernel)
        self._reduce_fn = reduce_fn       # CPU tests inject numpy local reductions (the product reduces with HIP kernels)
        model.set_shard(rank, world, mode)
        info = model.shard_info()
        self.mode = "class" if int(info.mode) == 1 else "range"
        self.packed = bool(int(info.packed))   # the send slabs index the packed send buffer (short runs) instead of the vector itself
        self.n_send = int(info.n_send)
        self._send = {}
        self.overlap = True               # run interior tiles while the halo exchange is in flight
        self.n_interior_tiles = int(info.n_interior_tiles)
        self.n_local, self.n_halo = int(info.n_local), int(info.n_halo)
        self.row_lo, self.row_hi = int(info.row_lo), int(info.row_hi)
        self.recv_slabs, self.send_slabs = model.shard_slabs()
        self._halo = {}
        self._comm = None
        self.comm_kind, self.comm_note = None, ""     # which communicator comm() settled on ("rccl" / "torch"), and why not RCCL
        self._routes = None               # two-hop routes of the halo messages (SD_RELAY=1, popcount-cell ownership, >= 3 ranks)

    def relay_plan(self, group=None, relay=None):
        """Routes of relay_routes for this operator's exchange, or None: built once, collectively, from every rank's receive
        list -- per (owner, receiver) pair the lengths of the segments the message consists of (one for a packed send buffer, one
        per contiguous run when the runs travel straight from psi).  Cell ownership, at least three ranks.  `relay` stands in
        for the environment's SD_RELAY ("1": routes that pay, "2": forced) when given."""
        import os
        if relay is None:
            relay = os.environ.get("SD_RELAY", "0")
        if self._routes is None:
            self._routes = False
            if relay not in ("", "0") and self.mode == "class" and self.world >= 3 \
                    and self._exchange_fn is None:
                import torch.distributed as dist
                mine = {}
                for (peer, _off, cnt, _g) in self.recv_slabs:
                    mine.setdefault(int(peer), []).append(int(cnt))
                everyone = [None] * self.world
                dist.all_gather_object(everyone, mine, group=group)
                self._relay_runs = {(o, r): lst for r, d in enumerate(everyone) for o, lst in d.items()}
                self._relay_M = {pr: sum(lst) for pr, lst in self._relay_runs.items()}
                routes = relay_routes(self._relay_M, int(os.environ.get("SD_RELAY_CHUNKS", "8")),
                                      int(os.environ.get("SD_RELAY_MIN", "65536")), force=relay == "2")
                if any(k >= 0 for lst in routes.values() for (k, _u) in lst):
                    self._routes = routes
        return self._routes or None

    def set_relay(self, routes):
        """Switch the exchange of BOTH transports to `routes` (a relay_plan result) or, with None, back to the direct exchange:
        the torch.distributed transport reads self._routes, the library's RCCL communicator gets the op list installed or removed.
        Collective in the sense that every rank must make the same call."""
        import os
        self._routes = routes if routes is not None else False
        if isinstance(self._comm, RcclComm):
            self._comm.set_routes(self, routes, int(os.environ.get("SD_RELAY_BATCHES", "4")))

    def _relay_runs_of(self, o, r, lo, hi):
        """Lengths of the segments in which the elements [lo, hi) of message o -> r travel from their owner."""
        out, pos = [], 0
        for cnt in self._relay_runs[(o, r)]:
            a, b = max(lo, pos), min(hi, pos + cnt)
            if a < b:
                out.append(b - a)
            pos += cnt
        return out

    def comm(self, device, group=None):
        """The communicator handed to the C recursion-level entry points (None for a single rank).

        Default (SD_COMM unset or "auto"): with an NCCL (= RCCL) process group the library's OWN RCCL communicator -- grouped
        ncclSend/ncclRecv beside the interior tiles and ncclAllReduce of the device scalars, queued on HIP streams by the C
        side, a recursion step never touches the host -- once every rank has created it and passed its self-test (an all-reduce
        and a ring send/receive, csrc/comm.cpp); otherwise, and for any other backend (gloo rehearsals), torch.distributed
        behind the callback communicator, which costs one host round trip per reduction.  The ranks agree on the outcome, so
        all of them take the same path; `self.comm_kind` says which ("rccl" / "torch"), `self.comm_note` why a fallback happened.
        SD_COMM=torch forces the callbacks, SD_COMM=rccl makes a failure of the RCCL path an error instead of a fallback."""
        import os
        if self.world == 1:
            return None
        if self._exchange_fn is not None:
            raise _lib.ArgumentError("virtual shards (one process) cover single applies only: the recursions need real ranks")
        if self._comm is None:
            import torch
            import torch.distributed as dist
            kind = os.environ.get("SD_COMM", "auto")
            backend = dist.get_backend(group)
            self.comm_kind, self.comm_note = None, ""
            if kind == "rccl" or (kind == "auto" and backend == "nccl"):
                cm, ok = None, False
                try:
                    cm = RcclComm(self, device, group)
                    m = self.model
                    m.ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)
                    check(lib().sd_comm_selftest(m.ctx.h, cm.h), m.ctx.h)
                    ok = True
                except Exception as e:          # noqa: BLE001 -- agreed on below: every rank falls back together
                    self.comm_note = "rank %d: the library's RCCL communicator failed (%r)" % (self.rank, e)
                t = torch.tensor([1.0 if ok else 0.0], device=device if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
                if float(t.item()) == 1.0:
                    self._comm, self.comm_kind = cm, "rccl"
                    routes = self.relay_plan(group)          # SD_RELAY: the same routing the torch.distributed transport uses
                    if routes is not None:
                        cm.set_routes(self, routes, int(os.environ.get("SD_RELAY_BATCHES", "4")))
                else:
                    if cm is not None:
                        cm.close()
                    if not self.comm_note:
                        self.comm_note = "another rank's RCCL communicator failed its self-test"
                    if kind == "rccl":
                        raise _lib.SpinDynError(_lib.SD_ECOMM, self.comm_note)
            if self._comm is None:
                self._comm, self.comm_kind = TorchComm(self, device, group), "torch"
        return self._comm

    def apply_lib(self, out, psi, group=None, overlap=True):
        """out = H psi on the owned rows through the library's own sharded entry point sd_apply_sharded (pack, halo exchange on the
        communicator of comm(), interior tiles beside it, boundary tiles after it) -- the step every sharded recursion runs."""
        code = _lib.SD_C128 if psi.is_complex() else _lib.SD_F64
        self._call(lib().sd_apply_sharded, psi, code, out.data_ptr(), psi.data_ptr(), self.n_local, 1 if overlap else 0, group=group)
        return out

    def _call(self, fn, x, *args, group=None):
        """Run a sd_*_sharded entry point on the stream torch is using for x's device."""
        import torch
        m = self.model
        m.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
        cm = self.comm(x.device, group)
        rc = fn(m.ctx.h, m.h, cm.h if cm is not None else None, *args)
        if rc != _lib.SD_OK and cm is not None and getattr(cm, "_err", None) is not None:
            err, cm._err = cm._err, None
            raise err
        check(rc, m.ctx.h)

    # ---- buffers ----
    def empty(self, dtype, device):
        """A vector of this shard: the n_local owned elements."""
        import torch
        return torch.empty(self.n_local, dtype=dtype, device=device)

    def halo(self, like):
        """The halo buffer matching `like`'s dtype/device (allocated once, reused by every exchange)."""
        import torch
        key = (like.dtype, str(like.device))
        if key not in self._halo:
            self._halo[key] = torch.empty(max(self.n_halo, 1), dtype=like.dtype, device=like.device)
        return self._halo[key]

    def sendbuf(self, like):
        import torch
        key = (like.dtype, str(like.device))
        if key not in self._send:
            self._send[key] = torch.empty(max(self.n_send, 1), dtype=like.dtype, device=like.device)
        return self._send[key]

    def pack(self, psi):
        """class mode: gather the tiles the peers asked for into the contiguous send buffer."""
        buf = self.sendbuf(psi)
        if self.n_send == 0:
            return buf
        if self._pack_fn is not None:
            self._pack_fn(self, psi, buf)
            return buf
        import torch
        m = self.model
        m.ctx.set_stream(torch.cuda.current_stream(psi.device).cuda_stream)
        code = _lib.SD_C128 if psi.is_complex() else _lib.SD_F64
        check(lib().sd_shard_pack_dev(m.ctx.h, m.h, code, psi.data_ptr(), buf.data_ptr()), m.ctx.h)
        return buf

    def halo_bytes(self, itemsize=16):
        return self.n_halo * itemsize

    def exchange_start(self, psi, group=None):
        """Post the halo exchange for psi (pack + grouped isend/irecv) and return the pending requests."""
        halo = self.halo(psi)
        if self._exchange_fn is not None:
            self._exchange_fn(self, psi, halo)
            return halo, []
        if self.world == 1:
            return halo, []
        import torch
        import torch.distributed as dist
        out = self.pack(psi) if self.packed else psi     # what the send slabs index
        src = torch.view_as_real(out) if out.is_complex() else out
        dst = torch.view_as_real(halo) if halo.is_complex() else halo
        nl = self.n_local
        staged = src.is_cuda and dist.get_backend(group) == "gloo"
        if staged:
            # rehearsal only (several ranks sharing one GPU, where RCCL refuses to run): gloo has no device send/recv,
            # so the messages go through host copies.  The production backend is "nccl" (= RCCL), device to device.
            src_dev, dst_dev = src, dst
            src, dst = src.cpu(), torch.empty(dst.shape, dtype=dst.dtype)
        routes = self.relay_plan(group)
        if routes is not None:
            per = 2 if out.is_complex() else 1
            fin = (lambda: dst_dev.copy_(dst)) if staged else None
            return halo, [_RelayExchange(self, routes, src.reshape(-1), dst.reshape(-1), per, group, fin, int(os.environ.get("SD_RELAY_BATCHES", "4")))]
        ops = []
        for (peer, off, cnt, _g) in self.recv_slabs:          # recv offsets are counted from the start of [owned | halo]
            ops.append(dist.P2POp(dist.irecv, dst[off - nl:off - nl + cnt], peer, group))
        for (peer, off, cnt, _g) in self.send_slabs:
            ops.append(dist.P2POp(dist.isend, src[off:off + cnt], peer, group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        if staged and reqs:
            reqs = [_StagedRequests(reqs, dst_dev, dst)]
        return halo, reqs

    def exchange(self, psi, group=None):
        """Fill the halo buffer from the owning ranks' copies of psi (collective over all ranks)."""
        halo, reqs = self.exchange_start(psi, group)
        for req in reqs:
            req.wait()
        return halo

    def _launch(self, out, psi, halo, epilogue, a=1.0, b=0.0, c=0j, prev=None, acc=None, part=0, c0=0j):
        import torch
        if self.n_local == 0:          # a rank may own no tile when there are fewer tiles than ranks
            return out
        m = self.model
        m.ctx.set_stream(torch.cuda.current_stream(psi.device).cuda_stream)
        code = _lib.SD_C128 if psi.is_complex() else _lib.SD_F64
        c = complex(c)
        if epilogue == 4:              # second term of a Chebyshev pair: psi_t += c0*psi; psi_t += c*out
            c0 = complex(c0)
            check(lib().sd_apply_sharded_cheb2_dev(m.ctx.h, m.h, out.data_ptr(), psi.data_ptr(),
                                                   halo.data_ptr() if self.n_halo else None, self.n_local, float(a), float(b),
                                                   c0.real, c0.imag, c.real, c.imag, prev.data_ptr(), acc.data_ptr(), part),
                  m.ctx.h)
            return out
        check(lib().sd_apply_sharded_dev(m.ctx.h, m.h, code, out.data_ptr(), psi.data_ptr(),
                                         halo.data_ptr() if self.n_halo else None, self.n_local, epilogue,
                                         float(a), float(b), c.real, c.imag,
                                         prev.data_ptr() if prev is not None else None,
                                         acc.data_ptr() if acc is not None else None, part), m.ctx.h)
        return out

    def _apply(self, out, psi, group, epilogue, exchange=True, **kw):
        """Exchange overlapped with compute: the interior tiles (all partners owned) run while the halo is in flight,
        the boundary tiles after it has landed."""
        if not exchange:
            return self._launch(out, psi, self.halo(psi), epilogue, **kw)
        halo, reqs = self.exchange_start(psi, group)
        if not reqs or not self.overlap:
            for req in reqs:
                req.wait()
            return self._launch(out, psi, halo, epilogue, **kw)
        self._launch(out, psi, halo, epilogue, part=1, **kw)
        for req in reqs:
            req.wait()                 # the compute stream now waits for the received halo
        return self._launch(out, psi, halo, epilogue, part=2, **kw)

    def apply(self, out, psi, group=None, exchange=True):
        """out = H psi on the owned rows; the halo is refreshed first (pass exchange=False to reuse it)."""
        return self._apply(out, psi, group, 0, exchange)

    def apply_profiled(self, out, psi, group=None):
        """One overlapped apply with HIP events between its phases; returns ms of {pack, post (host time to post the
        sends / receives), interior, exchange_wait (what the compute stream still waits for the halo after the interior
        tiles), boundary}.  Diagnostic twin of apply(): same launches in the same order (bench.py, N > 1)."""
        import time
        import torch
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        ev[5].record()
        if self.packed and self.world > 1 and self._exchange_fn is None:
            self.pack(psi)                       # timed alone; exchange_start packs the same bytes again
        ev[6].record()
        ev[0].record()
        t0 = time.perf_counter()
        halo, reqs = self.exchange_start(psi, group)
        post_ms = (time.perf_counter() - t0) * 1e3
        ev[1].record()
        self._launch(out, psi, halo, 0, part=1)
        ev[2].record()
        for req in reqs:
            req.wait()
        ev[3].record()
        self._launch(out, psi, halo, 0, part=2)
        ev[4].record()
        torch.cuda.synchronize(psi.device)
        return {"pack": ev[5].elapsed_time(ev[6]), "post_host": post_ms, "pack_and_post_on_stream": ev[0].elapsed_time(ev[1]),
                "interior": ev[1].elapsed_time(ev[2]), "exchange_wait": ev[2].elapsed_time(ev[3]),
                "boundary": ev[3].elapsed_time(ev[4])}

    def apply_rescaled(self, out, psi, a, b, group=None):
        return self._apply(out, psi, group, 1, a=a, b=b)

    def cheb_step(self, phi_next, phi_curr, phi_prev, psi_t, a, b, c, group=None):
        """One fused Chebyshev term (src/TimeEvolution/Chebyshev.jl:110-121) on this shard."""
        return self._apply(phi_next, phi_curr, group, 2, a=a, b=b, c=c, prev=phi_prev, acc=psi_t)

    def chebyshev_time_evolve(self, psi0, dt, cheb_n=100, Ebounds=(-1.0, 1.0), group=None):
        """chebyshev_time_evolve (src/TimeEvolution/Chebyshev.jl:61-124) on a sharded ComplexF64 state (psi0 = this
        rank's owned elements): sd_chebyshev_evolve_sharded -- every term is one halo exchange + one fused device pass,
        terms in pairs, nothing but the recursion's three vectors and psi(t) allocated.  Returns psi(t) (owned part)."""
        import torch
        if int(cheb_n) < 1:
            raise AssertionError("cheb_n must be >= 1")
        psi0 = psi0.to(torch.complex128).contiguous()
        out = torch.empty_like(psi0)
        self._call(lib().sd_chebyshev_evolve_sharded, psi0, psi0.data_ptr(), self.n_local, float(dt), int(cheb_n),
                   float(Ebounds[0]), float(Ebounds[1]), out.data_ptr(), group=group)
        return out

    def krylov_time_evolve(self, psi0, dt, kry_m=30, group=None):
        """krylov_time_evolve (src/TimeEvolution/Krylov.jl:136-192) on a sharded state: sd_krylov_evolve_sharded."""
        import torch
        psi0 = psi0.contiguous()
        code = _lib.SD_C128 if psi0.is_complex() else _lib.SD_F64
        out = torch.empty(self.n_local, dtype=torch.complex128, device=psi0.device)
        self._call(lib().sd_krylov_evolve_sharded, psi0, code, psi0.data_ptr(), self.n_local, float(dt), int(kry_m),
                   out.data_ptr(), group=group)
        return out

    # ---- KPM on a sharded state (BASELINE config 5: L=36 over 8 GPUs) ----
    def _allreduce(self, vals, device, group=None):
        import torch
        import torch.distributed as dist
        if self.world == 1 or self._exchange_fn is not None:
            return [float(v) for v in vals]
        t = torch.tensor(list(vals), dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return [float(v) for v in t.tolist()]

    def dot(self, x, y, group=None):
        """<x|y> over all ranks (conjugate-linear in x)."""
        import torch
        loc = (C.c_double * 2)(0.0, 0.0)
        if self._reduce_fn is not None:
            v = complex(self._reduce_fn("dot", x, y))
            loc[0], loc[1] = v.real, v.imag
        elif self.n_local:
            m = self.model
            m.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
            code = _lib.SD_C128 if x.is_complex() else _lib.SD_F64
            check(lib().sd_dot_dev(m.ctx.h, code, x.data_ptr(), y.data_ptr(), self.n_local, loc), m.ctx.h)
        re, im = self._allreduce([loc[0], loc[1]], x.device, group)
        return complex(re, im)

    def Sz_q_vector(self, psi0, q):
        """Sz_q_vector on the owned rows (purely local: a diagonal operator)."""
        import torch
        phi = torch.empty(self.n_local, dtype=torch.complex128, device=psi0.device)
        if self.n_local:
            m = self.model
            m.ctx.set_stream(torch.cuda.current_stream(psi0.device).cuda_stream)
            code = _lib.SD_C128 if psi0.is_complex() else _lib.SD_F64
            check(lib().sd_szq_dev(m.ctx.h, m.h, code, psi0.data_ptr(), self.n_local, float(q), phi.data_ptr()), m.ctx.h)
        return phi

    def kpm_moments(self, phi, M, a, b, group=None, doubling=True):
        """compute_chebyshev_moments (src/KPM_Sqw.jl:95-128) for a normalised sharded phi; returns mu[0..M):
        sd_kpm_moments_sharded.  doubling (default): two moments per apply, mu_2n = 2<v_n|v_n> - mu_0,
        mu_2n+1 = 2Re<v_n|v_n+1> - mu_1; doubling=False: the reference's loop, one moment <phi|v_k> per apply."""
        import numpy as np
        if M < 2:
            raise _lib.ArgumentError("kpm_m must be >= 2")
        mu = np.zeros(int(M))
        before = self.model.ctx.kpm_doubling          # the caller's own choice for unsharded calls is put back afterwards
        self.model.ctx.set_kpm_doubling(bool(doubling))
        try:
            phi = phi.contiguous()
            self._call(lib().sd_kpm_moments_sharded, phi, phi.data_ptr(), self.n_local, int(M), float(a), float(b),
                       mu.ctypes.data_as(C.POINTER(C.c_double)), group=group)
        finally:
            self.model.ctx.set_kpm_doubling(before)
        return mu

    # ---- Lanczos on a sharded state (energy bounds for the sharded KPM / Chebyshev drivers) ----
    def lanczos_extremal(self, lanc_m=100, tol=1e-12, psi0=None, seed=0, group=None, device="cuda", negate=False):
        """lanczos_extremal (src/Lanczos.jl:27-84) with the vectors sharded: sd_lanczos_extremal_sharded, returns
        (Emin, Emax) of the Lanczos matrix.  psi0: this rank's owned part of the start vector (ComplexF64); default: the
        counter-based N(0,1) vector of `seed` (identical for every sharding).  alpha_j and beta_j are summed over the ranks
        on the device (RCCL) or through one host round trip each (torch.distributed callbacks)."""
        import torch
        if int(lanc_m) < 1:
            raise _lib.ArgumentError("lanc_m must be >= 1")
        lo, hi = C.c_double(), C.c_double()
        anchor = psi0 if psi0 is not None else torch.empty(0, device=device)
        if psi0 is not None:
            psi0 = psi0.to(torch.complex128).contiguous()
            if float(self.norm(psi0, group)) == 0.0:
                raise _lib.ZeroNormError("starting vector has zero norm")
        self._call(lib().sd_lanczos_extremal_sharded, anchor, int(lanc_m), float(tol),
                   psi0.data_ptr() if psi0 is not None else None, int(seed), 1 if negate else 0, C.byref(lo), C.byref(hi),
                   group=group)
        return lo.value, hi.value

    def estimate_energy_bounds(self, lanc_m=80, seed=0, group=None, device="cuda"):
        """estimate_energy_bounds (src/Lanczos.jl:255-271): Emax from a Lanczos run on H, Emin = -Emax of a run on -H."""
        import torch
        lo, hi = C.c_double(), C.c_double()
        self._call(lib().sd_energy_bounds_sharded, torch.empty(0, device=device), int(lanc_m), int(seed), C.byref(lo),
                   C.byref(hi), group=group)
        return lo.value, hi.value

    def kpm_sqw(self, psi0, q_list, omega, a=None, b=None, kpm_m=200, kernel="jackson", group=None, seed=0):
        """kpm_sqw (src/KPM_Sqw.jl:191-256) on a sharded ComplexF64/Float64 psi0: sd_kpm_sqw_sharded; without (a, b) the
        rescaling is estimated as the reference does (:212-214: energy bounds from two Lanczos runs, lanc_m = 80)."""
        import numpy as np
        psi0 = psi0.contiguous()
        code = _lib.SD_C128 if psi0.is_complex() else _lib.SD_F64
        q = np.ascontiguousarray(q_list, dtype=np.float64)
        om = np.ascontiguousarray(omega, dtype=np.float64)
        S = np.zeros((len(q), len(om)))
        dp = C.POINTER(C.c_double)
        have = a is not None and b is not None
        self._call(lib().sd_kpm_sqw_sharded, psi0, code, psi0.data_ptr(), self.n_local, q.ctypes.data_as(dp), len(q),
                   om.ctypes.data_as(dp), len(om), 1 if have else 0, float(a) if have else 0.0, float(b) if have else 0.0,
                   int(kpm_m), _lib.KERNELS.get(kernel, 2), int(seed), S.ctypes.data_as(dp), group=group)
        return S

    def fill_randn(self, x, seed):
        """Counter-based N(0,1) keyed by the GLOBAL element index: identical for every sharding."""
        import torch
        if self.n_local == 0:
            return x
        m = self.model
        m.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
        code = _lib.SD_C128 if x.is_complex() else _lib.SD_F64
        check(lib().sd_fill_randn_local_dev(m.ctx.h, m.h, code, x.data_ptr(), int(seed)), m.ctx.h)
        return x

    def norm(self, x, group=None):
        import torch
        loc = C.c_double(0.0)
        if self._reduce_fn is not None:
            loc.value = float(self._reduce_fn("nrm2sq", x, None))
        elif self.n_local:
            m = self.model
            m.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
            code = _lib.SD_C128 if x.is_complex() else _lib.SD_F64
            check(lib().sd_nrm2sq_dev(m.ctx.h, code, x.data_ptr(), self.n_local, C.byref(loc)), m.ctx.h)
        return float(self._allreduce([loc.value], x.device, group)[0]) ** 0.5


def kpm_sqw_replicas(psi0, model, q_list, omega, a, b, kpm_m=200, kernel="jackson", group=None):
    """kpm_sqw (src/KPM_Sqw.jl:191-256) with the momenta dealt over the ranks instead of the vector: the q values are
    independent (the reference threads over them, :218), so when the recursion's vectors fit one GPU (L <= 32) rank r runs
    q_list[r::world] with the unsharded single-GPU recursion on its own replica of psi0 and the rows are combined with one
    all-reduce of the Q x W matrix -- no data-path communication at all.  `model` is an UNSHARDED model on this rank's
    GPU; every rank passes the same psi0 / q_list / omega and gets the full matrix back.  Without an initialised process
    group it is the plain single-GPU kpm_sqw."""
    import numpy as np
    from .solvers import kpm_sqw
    q_all = np.ascontiguousarray(q_list, dtype=np.float64)
    world, rank, dist = 1, 0, None
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            dist = None
    except ImportError:
        dist = None
    S = np.zeros((len(q_all), len(omega)))
    mine = np.arange(rank, len(q_all), world)
    if len(mine):
        S[mine] = kpm_sqw(psi0, model, q_all[mine], omega, a=a, b=b, kpm_m=kpm_m, kernel=kernel)
    if dist is not None and world > 1:
        import torch
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.from_numpy(S).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)     # every row is non-zero on exactly one rank
        S = t.cpu().numpy()
    return S
