"""Mirror of the reference's SpinModel module (src/SpinModel.jl) on top of sd_model.

`Model` carries the same descriptor fields as SpinModel.Model (:6-15) except
`states` / `idxmap`, which are never materialised for the full dimension: the
library reproduces the basis order of build_sector_basis (src/Basis.jl:37-53)
from closed-form ranking.  `model.states` is computed on demand for parity
checks.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArgumentError, check, lib


def _arr(vals, dtype):
    return np.ascontiguousarray(np.array(list(vals), dtype=dtype))


class Model:
    def __init__(self, L, nup=None, hopping=(), onsite_field=None, zz=(), ctx="default"):
        if not isinstance(L, (int, np.integer)):
            raise ArgumentError("L must be an integer")
        self.L = int(L)
        self.nup = None if nup is None else int(nup)
        self.mode = "full" if nup is None else "sector"
        self.hopping_list = [(int(i), int(j), float(J)) for (i, j, J) in hopping]
        self.zz_list = [(int(i), int(j), float(J)) for (i, j, J) in zz]
        if onsite_field is None:
            onsite_field = np.zeros(max(self.L, 0))
        self.onsite_field = np.ascontiguousarray(onsite_field, dtype=np.float64)
        if self.L >= 1 and len(self.onsite_field) != self.L:
            raise ArgumentError("onsite_field must have L entries")
        self.ctx = _lib.default_context() if ctx == "default" else ctx
        hi = _arr((h[0] for h in self.hopping_list), np.int32)
        hj = _arr((h[1] for h in self.hopping_list), np.int32)
        hJ = _arr((h[2] for h in self.hopping_list), np.float64)
        zi = _arr((h[0] for h in self.zz_list), np.int32)
        zj = _arr((h[1] for h in self.zz_list), np.int32)
        zJ = _arr((h[2] for h in self.zz_list), np.float64)
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        self.h = C.c_void_p()
        ch = self.ctx.h if self.ctx is not None else None
        check(lib().sd_model_create(ch, self.L, -1 if self.nup is None else self.nup,
                                    len(hi), hi.ctypes.data_as(ip), hj.ctypes.data_as(ip), hJ.ctypes.data_as(dp),
                                    len(zi), zi.ctypes.data_as(ip), zj.ctypes.data_as(ip), zJ.ctypes.data_as(dp),
                                    self.onsite_field.ctypes.data_as(dp), C.byref(self.h)), ch)
        self.N = int(lib().sd_model_dim(self.h))

    def __del__(self):
        try:
            # an operator this model installed must not outlive it: the trampoline closes over the model
            ctx = getattr(self, "ctx", None)
            if ctx is not None and getattr(ctx, "_apply_owner", None) is not None and ctx._apply_owner() is self:
                ctx.clear_apply()
        except Exception:
            pass
        try:
            if getattr(self, "h", None):
                lib().sd_model_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def __len__(self):
        return self.N

    def set_apply(self, fn):
        """Install `fn(out, psi, model)` as the operator of every recursion run on this model's CONTEXT (the reference's
        applyH! argument; sd_ctx_set_apply_callback), or restore the built-in operator with fn=None.  out and psi are torch tensors on the
        model's device (Float64 or ComplexF64, this rank's rows), valid during the call only; fn runs with torch's current
        stream set to the library's stream and must write H psi into out.

        The operator belongs to the context, as in the C ABI: it replaces H for every model whose recursions run on that
        context, so installing one on a context that other live models share (the default context is shared by all models
        built without `ctx=`) is refused while another model's operator is installed, and the solvers' `applyH=` argument --
        installed for the duration of one call -- is the safe route.  The trampoline is kept alive by the context and removed
        when the installing model is garbage-collected; the cached energy bounds of time_evolve(:chebyshev) are dropped."""
        if self.ctx is None:
            raise ArgumentError("this model has no context: build it with ctx= to run recursions with a caller's operator")
        self._apply_err = None
        self.__dict__.pop("_energy_bounds", None)       # bounds estimated for another operator must not be reused (api.py)
        ctx = self.ctx
        owner = ctx._apply_owner() if getattr(ctx, "_apply_owner", None) is not None else None
        if fn is None:
            if owner is None or owner is self:
                ctx.clear_apply()
            return
        if owner is not None and owner is not self:
            raise ArgumentError("another model has installed an operator on this context; give each model with its own "
                                "operator a context of its own (Model(..., ctx=Context(device))) or pass applyH= to the solver")
        import weakref
        me = weakref.ref(self)

        def tramp(_user, dtype, out_ptr, psi_ptr, n, stream):
            model = me()
            try:                        # an exception must not unwind through the C frames
                if model is None:
                    return 1
                import torch
                dev = torch.device("cuda", ctx.device)
                per = 2 if dtype == _lib.SD_C128 else 1
                out = _lib.dev_tensor(out_ptr, max(int(n), 1) * per, dev)[: int(n) * per]
                psi = _lib.dev_tensor(psi_ptr, max(int(n), 1) * per, dev)[: int(n) * per]
                if per == 2:
                    out, psi = torch.view_as_complex(out.view(-1, 2)), torch.view_as_complex(psi.view(-1, 2))
                st = torch.cuda.ExternalStream(int(stream), device=dev) if stream else torch.cuda.default_stream(dev)
                with torch.cuda.stream(st):
                    fn(out, psi, model)
                return 0
            except Exception as e:
                if model is not None:
                    model._apply_err = e
                return 1

        ctx.install_apply(_lib.APPLY_FN(tramp), me)

    # -- basis queries (host) --
    def states_range(self, start, count):
        out = np.empty(count, dtype=np.uint64)
        check(lib().sd_model_states(self.h, start, count, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    @property
    def states(self):
        """model.states (0-based position k holds the reference's states[k+1])."""
        return self.states_range(0, self.N)

    def rank(self, states):
        """0-based index of each state (reference idxmap value - 1), -1 when absent."""
        st = np.ascontiguousarray(states, dtype=np.uint64).ravel()
        out = np.empty(len(st), dtype=np.int64)
        check(lib().sd_model_rank(self.h, st.ctypes.data_as(C.POINTER(C.c_uint64)), len(st),
                                  out.ctypes.data_as(C.POINTER(C.c_int64))))
        return out

    @property
    def device_path(self):
        return {1: "tiled", 2: "full-tiled"}.get(lib().sd_model_path(self.h), "generic")

    # -- sharding --
    def set_shard(self, rank, nranks, mode=None):
        """mode: None/"auto" (env SD_SHARD_MODE, default "class"), "range" (contiguous index ranges) or "class"
        (popcount cells: 3-4x less halo traffic, owned rows are a union of tiles)."""
        code = {None: -1, "auto": -1, "range": 0, "class": 1}[mode]
        check(lib().sd_model_set_shard_mode(self.h, rank, nranks, code), self.ctx.h if self.ctx else None)

    def local_tiles(self):
        """(local_base, global_base, len) of this shard's tiles, natural order: local row lb+i <-> global row gb+i."""
        n = int(self.shard_info().n_local_tiles)
        lb, gb, ln = np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.int32)
        check(lib().sd_model_local_tiles(self.h, lb.ctypes.data_as(C.POINTER(C.c_int64)), gb.ctypes.data_as(C.POINTER(C.c_int64)),
                                         ln.ctypes.data_as(C.POINTER(C.c_int))))
        return lb, gb, ln

    def pack_list(self):
        """(src, dst, len) of the cell-mode pack list: psi[src:src+len] -> sendbuf[dst:dst+len]."""
        n = int(self.shard_info().n_pack)
        a, b, c = np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.int32)
        check(lib().sd_model_shard_pack_list(self.h, a.ctypes.data_as(C.POINTER(C.c_int64)), b.ctypes.data_as(C.POINTER(C.c_int64)),
                                             c.ctypes.data_as(C.POINTER(C.c_int))))
        return a, b, c

    def local_rows(self):
        """Global basis index of every local row (int64 array of n_local entries) -- for tests / import-export."""
        info = self.shard_info()
        if self.nup is None:          # full basis: a rank owns the contiguous rows [row_lo, row_hi) (top index bits)
            return np.arange(int(info.row_lo), int(info.row_hi), dtype=np.int64)
        lb, gb, ln = self.local_tiles()
        out = np.empty(int(self.shard_info().n_local), np.int64)
        for a, b, c in zip(lb, gb, ln):
            out[a:a + c] = np.arange(b, b + c)
        return out

    def shard_info(self):
        info = _lib.sd_shard_info()
        check(lib().sd_model_shard_info(self.h, C.byref(info)))
        return info

    def shard_slabs(self):
        info = self.shard_info()
        recv = (_lib.sd_slab * max(int(info.n_recv_slabs), 1))()
        send = (_lib.sd_slab * max(int(info.n_send_slabs), 1))()
        check(lib().sd_model_shard_slabs(self.h, recv, send))
        r = [(s.peer, int(s.local_offset), int(s.count), int(s.global_row)) for s in recv[: int(info.n_recv_slabs)]]
        s = [(s.peer, int(s.local_offset), int(s.count), int(s.global_row)) for s in send[: int(info.n_send_slabs)]]
        return r, s


def build_model(L, nup=None, hopping=(), onsite_field=None, zz=(), ctx="default"):
    """build_model(L; nup, hopping, onsite_field, zz) -- src/SpinModel.jl:23-38"""
    return Model(L, nup=nup, hopping=hopping, onsite_field=onsite_field, zz=zz, ctx=ctx)


def nn_hopping(L, J):
    """src/SpinModel.jl:40-42"""
    return [(i, i + 1, float(J)) for i in range(1, L)]


def long_range_hopping(L, J):
    """src/SpinModel.jl:44-46 (J is a callable J(i, j))"""
    return [(i, j, float(J(i, j))) for i in range(1, L + 1) for j in range(i + 1, L + 1)]


def XXZChain(L, Jxy=1.0, Jz=1.0, hz=0.0, nup=None, boundary="open", ctx="default"):
    """XXZChain(L; Jxy, Jz, hz, nup, boundary) -- src/SpinModel.jl:63-90"""
    hopping = [(i, i + 1, float(Jxy) / 2) for i in range(1, L)]
    zz = [(i, i + 1, float(Jz)) for i in range(1, L)]
    if boundary == "periodic":
        if L > 2:
            hopping.append((L, 1, float(Jxy) / 2))
            zz.append((L, 1, float(Jz)))
    elif boundary != "open":
        raise ArgumentError("boundary must be :open or :periodic")
    return Model(L, nup=nup, hopping=hopping, onsite_field=np.full(max(L, 0), float(hz)), zz=zz, ctx=ctx)


def momenta(model):
    """q = 2*pi*n/L, n = 0..L-1 -- src/SpinModel.jl:97-99"""
    return 2 * np.pi * np.arange(model.L) / model.L
