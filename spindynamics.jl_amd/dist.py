"""Index-range sharding of the Hilbert-space vector over GPUs (one process per GPU).

The reference has no distributed layer; this is the new capability BASELINE.json's
north star asks for.  Rank r owns the contiguous basis-index range
[row_lo, row_hi) (tile aligned).  Before each apply the psi values of hop
partners that live on other ranks are imported into the halo tail of the local
vector: a grouped send/recv of a few CONTIGUOUS slabs per peer (a hop on a
prefix bond maps whole tiles onto whole tiles, so no per-element index lists
travel) -- torch.distributed P2P ops, i.e. RCCL over xGMI on GPUs (backend
"nccl") and gloo in the CPU tests.
"""
import ctypes as C

from . import _lib
from ._lib import check, lib


class ShardedOperator:
    def __init__(self, model, rank, world, exchange_fn=None):
        self.model = model
        self._exchange_fn = exchange_fn   # tests inject an emulated exchange for virtual shards in one process
        self.rank, self.world = rank, world
        model.set_shard(rank, world)
        info = model.shard_info()
        self.n_local, self.n_halo = int(info.n_local), int(info.n_halo)
        self.row_lo, self.row_hi = int(info.row_lo), int(info.row_hi)
        self.recv_slabs, self.send_slabs = model.shard_slabs()

    # ---- vectors: n_local owned elements followed by the halo tail ----
    def empty(self, dtype, device):
        import torch
        return torch.empty(self.n_local + self.n_halo, dtype=dtype, device=device)

    def halo_bytes(self, itemsize=16):
        return self.n_halo * itemsize

    def exchange(self, psi, group=None):
        """Fill the halo tail of psi from the owning ranks (collective over all ranks)."""
        if self._exchange_fn is not None:
            return self._exchange_fn(self, psi)
        if self.world == 1:
            return
        import torch
        import torch.distributed as dist
        flat = torch.view_as_real(psi) if psi.is_complex() else psi
        ops = []
        for (peer, off, cnt, _g) in self.recv_slabs:
            ops.append(dist.P2POp(dist.irecv, flat[off:off + cnt], peer, group))
        for (peer, off, cnt, _g) in self.send_slabs:
            ops.append(dist.P2POp(dist.isend, flat[off:off + cnt], peer, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def apply(self, out, psi, group=None, exchange=True):
        """out[:n_local] = (H psi)[row_lo:row_hi]; psi's halo is refreshed first."""
        import torch
        if exchange:
            self.exchange(psi, group)
        if self.n_local == 0:      # a rank may own no tile when there are fewer tiles than ranks
            return out
        m = self.model
        m.ctx.set_stream(torch.cuda.current_stream(psi.device).cuda_stream)
        code = _lib.SD_C128 if psi.is_complex() else _lib.SD_F64
        check(lib().sd_apply_dev(m.ctx.h, m.h, code, out.data_ptr(), psi.data_ptr(), self.n_local), m.ctx.h)
        return out

    def cheb_step(self, phi_next, phi_curr, phi_prev, psi_t, a, b, c, group=None):
        import torch
        self.exchange(phi_curr, group)
        if self.n_local == 0:
            return phi_next
        m = self.model
        m.ctx.set_stream(torch.cuda.current_stream(phi_curr.device).cuda_stream)
        c = complex(c)
        check(lib().sd_cheb_step_dev(m.ctx.h, m.h, phi_next.data_ptr(), phi_curr.data_ptr(), phi_prev.data_ptr(),
                                     psi_t.data_ptr(), self.n_local, float(a), float(b), c.real, c.imag), m.ctx.h)
        return phi_next

    def chebyshev_time_evolve(self, psi0, dt, cheb_n=100, Ebounds=(-1.0, 1.0), group=None):
        """chebyshev_time_evolve (src/TimeEvolution/Chebyshev.jl:61-124) on a sharded ComplexF64 state: psi0 is this
        rank's [owned | halo] tensor (owned part filled).  Every term is one halo exchange + one fused device pass.
        Returns a new tensor whose owned part holds psi(t)."""
        import torch
        from .solvers import chebyshev_coeffs
        if int(cheb_n) < 1:
            raise AssertionError("cheb_n must be >= 1")
        Emin, Emax = Ebounds
        a = (Emax - Emin) / (2 * 0.9999)
        b = (Emax + Emin) / 2
        c = chebyshev_coeffs(cheb_n, a, b, dt)
        m = self.model
        prev, cur, nxt, acc = psi0.clone(), torch.zeros_like(psi0), torch.zeros_like(psi0), torch.zeros_like(psi0)
        nl = self.n_local
        # phi_curr = H~ phi_prev  (Chebyshev.jl:93)
        self.exchange(prev, group)
        if nl:
            m.ctx.set_stream(torch.cuda.current_stream(prev.device).cuda_stream)
            check(lib().sd_apply_rescaled_dev(m.ctx.h, m.h, _lib.SD_C128, cur.data_ptr(), prev.data_ptr(), nl, float(a), float(b)),
                  m.ctx.h)
        # psi_t = c0*T0 + c1*T1  (Chebyshev.jl:96-102; same rounding sequence as the single-GPU kernel)
        acc[:nl] = 0
        acc[:nl] += complex(c[0]) * prev[:nl]
        if cheb_n >= 2:
            acc[:nl] += complex(c[1]) * cur[:nl]
        for k in range(2, int(cheb_n)):
            self.cheb_step(nxt, cur, prev, acc, a, b, complex(c[k]), group)
            prev, cur, nxt = cur, nxt, prev
        return acc

    def fill_randn(self, x, seed):
        """Counter-based N(0,1) keyed by the GLOBAL element index: identical for every sharding."""
        import torch
        if self.n_local == 0:
            return x
        m = self.model
        m.ctx.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
        per = 2 if x.is_complex() else 1
        check(lib().sd_fill_randn_dev(m.ctx.h, x.data_ptr(), self.n_local * per, int(seed), self.row_lo * per), m.ctx.h)
        return x

    def norm(self, x, group=None):
        import torch
        import torch.distributed as dist
        s = torch.linalg.vector_norm(x[: self.n_local]) ** 2
        if self.world > 1:
            dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
        return float(s.sqrt())
