"""Test helper (not a test module): P ranks of a sharded recursion as P THREADS of one process on one GPU.

The recursion-level sharded entry points (sd_*_sharded, include/spindyn.h) are collective: every rank runs the same
loop and meets its peers in the halo exchange of each apply and in the sum of each reduction.  `tests/test_gpu_sharded.py`
checks the sharded APPLY with virtual shards stepped one after the other; a whole recursion needs all ranks alive at
once.  Here each rank is a thread with its own sd_ctx and sharded model, and the communicator is `sd_comm_from_callbacks`
with callbacks that move the bytes between the ranks' device buffers (device-to-device copies following the very slab
lists a real exchange sends) and sum the scalars in rank order -- everything of the multi-rank path except the wire.
ctypes releases the GIL for the duration of a library call, so the threads really run side by side.
"""
import ctypes as C
import threading

import numpy as np


class _Shared:
    def __init__(self, P):
        self.P = P
        self.barrier = threading.Barrier(P, timeout=600)
        self.src = [None] * P          # (device pointer of what the send slabs index, dtype code) per rank
        self.red = np.zeros((P, 1024))
        self.ops = [None] * P
        self.n_exchange = 0
        self.n_reduce = 0


class ThreadComm:
    """sd_comm whose callbacks exchange with the sibling threads of this process."""

    def __init__(self, pkg, shared, op, device):
        from spindynamics_jl_amd import _lib
        self._lib, self.sh, self.op, self.device = _lib, shared, op, device
        self._err = None
        self._halo = 0
        self._dtype = _lib.SD_C128
        self._cbs = _lib.sd_comm_callbacks(None, _lib.EXCHANGE_START_FN(self._start), _lib.EXCHANGE_WAIT_FN(self._wait),
                                           _lib.ALLREDUCE_FN(self._allreduce))
        self.h = C.c_void_p()
        _lib.check(_lib.lib().sd_comm_from_callbacks(C.byref(self._cbs), op.rank, op.world, C.byref(self.h)))

    def _start(self, _user, dtype, src_ptr, halo_ptr):
        try:
            import torch
            torch.cuda.current_stream(self.device).synchronize()       # the pack kernel of the C side has finished
            self.sh.src[self.op.rank] = (int(src_ptr or 0), int(dtype))
            self._halo, self._dtype = int(halo_ptr or 0), int(dtype)
            return 0
        except Exception as e:          # must not unwind through the C frames
            self._err = e
            self.sh.barrier.abort()
            return 1

    def _wait(self, _user):
        try:
            import torch
            sh, op, _lib = self.sh, self.op, self._lib
            sh.barrier.wait()                                           # every rank has posted its source buffer
            per = 2 if self._dtype == _lib.SD_C128 else 1
            nl = op.n_local
            if op.n_halo:
                dst = _lib.dev_tensor(self._halo, op.n_halo * per, self.device)
                for q in range(sh.P):
                    recvs = [s for s in op.recv_slabs if s[0] == q]
                    if not recvs:
                        continue
                    oq = sh.ops[q]
                    sends = [s for s in oq.send_slabs if s[0] == op.rank]
                    assert len(sends) == len(recvs)
                    n_src = oq.n_send if oq.packed else oq.n_local
                    src = _lib.dev_tensor(sh.src[q][0], n_src * per, self.device)
                    for (_p, so, cnt, _g), (_p2, ro, cnt2, _g2) in zip(sends, recvs):
                        assert cnt == cnt2
                        dst[(ro - nl) * per:(ro - nl + cnt) * per] = src[so * per:(so + cnt) * per]
                torch.cuda.current_stream(self.device).synchronize()
            if op.rank == 0:
                sh.n_exchange += 1
            sh.barrier.wait()                                           # nobody reuses a source buffer before all have read it
            return 0
        except Exception as e:
            self._err = e
            self.sh.barrier.abort()
            return 1

    def _allreduce(self, _user, vals, count):
        try:
            sh = self.sh
            a = np.ctypeslib.as_array(vals, shape=(count,))
            sh.red[self.op.rank, :count] = a
            sh.barrier.wait()
            tot = np.zeros(count)
            for r in range(sh.P):                                       # rank order: the same sum on every rank
                tot += sh.red[r, :count]
            if self.op.rank == 0:
                sh.n_reduce += 1
            sh.barrier.wait()
            a[:] = tot
            return 0
        except Exception as e:
            self._err = e
            self.sh.barrier.abort()
            return 1

    def close(self):
        if self.h:
            self._lib.lib().sd_comm_destroy(self.h)
            self.h = None


class VirtualRanks:
    """P sharded operators of one model (one context each) wired to each other through ThreadComm."""

    def __init__(self, pkg, make_model, P, mode, device=0):
        import torch
        from spindynamics_jl_amd import _lib
        self.pkg, self.P = pkg, P
        self.device = torch.device("cuda", device)
        self.sh = _Shared(P)
        self.ctxs = [_lib.Context(device) for _ in range(P)]
        self.models = [make_model(ctx) for ctx in self.ctxs]
        self.ops = []
        for r in range(P):
            op = pkg.ShardedOperator(self.models[r], r, P, mode=mode)
            self.ops.append(op)
            self.sh.ops[r] = op
        for op in self.ops:
            op._comm = ThreadComm(pkg, self.sh, op, self.device)
        self.rows = [m.local_rows() for m in self.models]

    def scatter(self, vec):
        """This process' copy of a global host vector -> the ranks' owned parts on the device."""
        import torch
        return [torch.from_numpy(np.ascontiguousarray(vec[rows])).to(self.device) for rows in self.rows]

    def gather(self, parts, dtype=complex):
        n = sum(len(r) for r in self.rows)
        out = np.empty(n, dtype=dtype)
        for rows, p in zip(self.rows, parts):
            out[rows] = p.cpu().numpy() if hasattr(p, "cpu") else p
        return out

    def run(self, fn):
        """fn(rank, op) on every rank at once; returns the list of results (re-raises the first failure)."""
        res, errs = [None] * self.P, [None] * self.P

        def work(r):
            try:
                res[r] = fn(r, self.ops[r])
            except BaseException as e:      # noqa: BLE001 -- handed to the caller below
                errs[r] = e
                self.sh.barrier.abort()

        th = [threading.Thread(target=work, args=(r,)) for r in range(self.P)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in errs:
            if e is not None and not isinstance(e, threading.BrokenBarrierError):
                raise e
        for e in errs:
            if e is not None:
                raise e
        self.sh.barrier.reset()
        return res

    def close(self):
        for op in self.ops:
            if op._comm is not None:
                op._comm.close()
                op._comm = None
