"""ctypes binding of libspindyn.so (include/spindyn.h).

The library is the product: there is no Python / CPU fallback.  Importing this
module without a built library raises, and creating a context without a GPU
raises (SD_ENODEV).
"""
import ctypes as C
import os
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SD_LIB_PATH") or os.path.join(_PKG, "libspindyn.so")   # SD_LIB_PATH: A/B builds of the same ABI

SD_OK, SD_EARG, SD_EDIM, SD_EZERO, SD_ENOMEM, SD_EHIP, SD_ENODEV, SD_EINTERNAL, SD_ECOMM = range(9)
SD_F64, SD_C128 = 1, 2
KERNELS = {"jackson": 0, "lorentz": 1}
BROADEN = {"lorentz": 0, "gauss": 1}


class SpinDynError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"libspindyn status {code}: {msg}")
        self.code = code


class ArgumentError(ValueError):
    """Julia ArgumentError (src/Basis.jl:10-16, src/SpinModel.jl:80, src/PublicAPI.jl:34,87,152)."""


class DimensionMismatch(ValueError):
    """Julia DimensionMismatch / length AssertionError (src/Hamiltonian.jl:63-66,220,289)."""


class ZeroNormError(RuntimeError):
    """error("starting vector has zero norm") (src/Lanczos.jl:210-212)."""


class sd_shard_info(C.Structure):
    _fields_ = [("rank", C.c_int), ("nranks", C.c_int), ("row_lo", C.c_int64), ("row_hi", C.c_int64),
                ("n_local", C.c_int64), ("n_halo", C.c_int64), ("n_recv_slabs", C.c_int64),
                ("n_send_slabs", C.c_int64), ("mode", C.c_int), ("n_send", C.c_int64), ("n_local_tiles", C.c_int64), ("n_pack", C.c_int64),
                ("n_interior_tiles", C.c_int64), ("n_interior_rows", C.c_int64), ("packed", C.c_int64)]


class sd_slab(C.Structure):
    _fields_ = [("peer", C.c_int), ("local_offset", C.c_int64), ("count", C.c_int64), ("global_row", C.c_int64)]


# sd_apply_fn (include/spindyn.h): the caller's operator at the recursion level (the reference's applyH! argument)
APPLY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class DevMem:
    """A raw device pointer as a __cuda_array_interface__ object (Float64 view), so that torch can wrap it without a copy."""

    def __init__(self, ptr, n_doubles):
        self.__cuda_array_interface__ = {"shape": (int(n_doubles),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


def dev_tensor(ptr, n_doubles, device):
    import torch
    return torch.as_tensor(DevMem(ptr, n_doubles), device=device)


# sd_comm_callbacks (include/spindyn.h): how a sharded recursion exchanges halos and sums scalars
EXCHANGE_START_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)
EXCHANGE_WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


class sd_xop(C.Structure):
    _fields_ = [("batch", C.c_int), ("peer", C.c_int), ("kind", C.c_int), ("buf", C.c_int), ("offset", C.c_int64), ("count", C.c_int64)]


class sd_comm_callbacks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("exchange_start", EXCHANGE_START_FN), ("exchange_wait", EXCHANGE_WAIT_FN),
                ("allreduce_sum", ALLREDUCE_FN)]


_vp, _i, _i64, _u64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
_ip, _dp, _u64p, _i64p, _fp = C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_float)

# name -> (restype, argtypes); this table is also what tests check against include/spindyn.h
PROTOTYPES = {
    "sd_version": (C.c_char_p, []),
    "sd_device_count": (_i, []),
    "sd_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "sd_ctx_destroy": (None, [_vp]),
    "sd_ctx_set_stream": (_i, [_vp, _vp]),
    "sd_ctx_set_kpm_doubling": (_i, [_vp, _i]),
    "sd_ctx_set_kpm_pair_q": (_i, [_vp, _i]),
    "sd_ctx_set_q_batch": (_i, [_vp, _i]),
    "sd_ctx_set_gs_blocked": (_i, [_vp, _i]),
    "sd_ctx_apply_count": (_i64, [_vp]),
    "sd_ctx_release_scratch": (_i, [_vp]),
    "sd_ctx_synchronize": (_i, [_vp]),
    "sd_last_error": (C.c_char_p, [_vp]),
    "sd_status_string": (C.c_char_p, [_i]),
    "sd_model_create": (_i, [_vp, _i, _i, _i, _ip, _ip, _dp, _i, _ip, _ip, _dp, _dp, C.POINTER(_vp)]),
    "sd_xxz_chain": (_i, [_vp, _i, _d, _d, _d, _i, _i, C.POINTER(_vp)]),
    "sd_model_destroy": (None, [_vp]),
    "sd_model_dim": (_i64, [_vp]),
    "sd_model_L": (_i, [_vp]),
    "sd_model_nup": (_i, [_vp]),
    "sd_model_path": (_i, [_vp]),
    "sd_model_states": (_i, [_vp, _i64, _i64, _u64p]),
    "sd_model_rank": (_i, [_vp, _u64p, _i64, _i64p]),
    "sd_apply": (_i, [_vp, _vp, _i, _vp, _vp, _i64]),
    "sd_apply_dev": (_i, [_vp, _vp, _i, _vp, _vp, _i64]),
    "sd_apply_rescaled": (_i, [_vp, _vp, _i, _vp, _vp, _i64, _d, _d]),
    "sd_apply_rescaled_dev": (_i, [_vp, _vp, _i, _vp, _vp, _i64, _d, _d]),
    "sd_szq": (_i, [_vp, _vp, _i, _vp, _i64, _d, _vp]),
    "sd_szq_dev": (_i, [_vp, _vp, _i, _vp, _i64, _d, _vp]),
    "sd_cheb_step_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _d, _d, _d, _d]),
    "sd_bench_apply_dev": (_i, [_vp, _vp, _i, _vp, _vp, _i64, _i, _fp]),
    "sd_lanczos_extremal": (_i, [_vp, _vp, _i, _d, _vp, _u64, _i, _dp, _dp]),
    "sd_energy_bounds": (_i, [_vp, _vp, _i, _vp, _vp, _u64, _dp, _dp]),
    "sd_lanczos_groundstate": (_i, [_vp, _vp, _i, _d, _d, _dp, _u64, _dp, _dp, _ip]),
    "sd_lanczos_tridiag": (_i, [_vp, _vp, _vp, _i64, _i, _d, _dp, _dp, _ip, _dp]),
    "sd_krylov_evolve": (_i, [_vp, _vp, _i, _vp, _i64, _d, _i, _vp]),
    "sd_krylov_evolve_dev": (_i, [_vp, _vp, _i, _vp, _i64, _d, _i, _vp]),
    "sd_chebyshev_evolve": (_i, [_vp, _vp, _vp, _i64, _d, _i, _d, _d, _vp]),
    "sd_chebyshev_evolve_dev": (_i, [_vp, _vp, _vp, _i64, _d, _i, _d, _d, _vp]),
    "sd_kpm_moments": (_i, [_vp, _vp, _vp, _i64, _i, _d, _d, _dp]),
    "sd_kpm_kernel": (_i, [_i, _i, _dp]),
    "sd_kpm_rescaling_from_bounds": (_i, [_d, _d, _dp, _dp]),
    "sd_kpm_reconstruct": (_i, [_dp, _i, _dp, _i, _d, _d, _d, _dp]),
    "sd_kpm_sqw": (_i, [_vp, _vp, _i, _vp, _i64, _dp, _i, _dp, _i, _i, _d, _d, _i, _i, _u64, _dp]),
    "sd_spectral_from_tridiagonal": (_i, [_dp, _dp, _i, _d, _d, _dp, _i, _d, _i, _dp]),
    "sd_lanczos_sqw": (_i, [_vp, _vp, _i, _vp, _i64, _dp, _i, _dp, _i, _i, _d, _i, _dp]),
    "sd_magnetization": (_i, [_vp, _vp, _i, _vp, _i64, _dp]),
    "sd_magnetization_dev": (_i, [_vp, _vp, _i, _vp, _i64, _dp]),
    "sd_connected_correlations": (_i, [_vp, _vp, _i, _vp, _i64, _dp]),
    "sd_connected_correlations_dev": (_i, [_vp, _vp, _i, _vp, _i64, _dp]),
    "sd_structure_factor": (_i, [_vp, _vp, _i, _vp, _i64, _dp, _dp]),
    "sd_structure_factor_dev": (_i, [_vp, _vp, _i, _vp, _i64, _dp, _dp]),
    "sd_initial_state_index": (_i, [_vp, _i, _ip, _i, _i64p]),
    "sd_dot_dev": (_i, [_vp, _i, _vp, _vp, _i64, _dp]),
    "sd_nrm2sq_dev": (_i, [_vp, _i, _vp, _i64, _dp]),
    "sd_spin_operator": (_i, [_vp, _vp, _i, _i, _i, _vp, _i64, _vp]),
    "sd_symtridiag_eig": (_i, [_i, _dp, _dp, _dp, _dp]),
    "sd_chebyshev_coeffs": (_i, [_i, _d, _d, _d, _dp]),
    "sd_fill_randn_dev": (_i, [_vp, _vp, _i64, _u64, _u64]),
    "sd_fill_randn_host": (_i, [_dp, _i64, _u64, _u64]),
    "sd_model_set_shard": (_i, [_vp, _i, _i]),
    "sd_apply_sharded_dev": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i64, _i, _d, _d, _d, _d, _vp, _vp, _i]),
    "sd_apply_sharded_cheb2_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _d, _d, _d, _d, _d, _d, _vp, _vp, _i]),
    "sd_kpm_step_sharded_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _d, _d, _i, _dp]),
    "sd_comm_from_callbacks": (_i, [C.POINTER(sd_comm_callbacks), _i, _i, C.POINTER(_vp)]),
    "sd_ctx_set_apply_callback": (_i, [_vp, APPLY_FN, _vp]),
    "sd_comm_rccl_unique_id": (_i, [_vp]),
    "sd_comm_rccl_create": (_i, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    "sd_comm_destroy": (None, [_vp]),
    "sd_comm_set_exchange_ops": (_i, [_vp, C.POINTER(sd_xop), _i64, _i64]),
    "sd_comm_selftest": (_i, [_vp, _vp]),
    "sd_apply_sharded": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i64, _i]),
    "sd_lanczos_extremal_sharded": (_i, [_vp, _vp, _vp, _i, _d, _vp, _u64, _i, _dp, _dp]),
    "sd_energy_bounds_sharded": (_i, [_vp, _vp, _vp, _i, _u64, _dp, _dp]),
    "sd_chebyshev_evolve_sharded": (_i, [_vp, _vp, _vp, _vp, _i64, _d, _i, _d, _d, _vp]),
    "sd_krylov_evolve_sharded": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _d, _i, _vp]),
    "sd_kpm_moments_sharded": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _d, _d, _dp]),
    "sd_kpm_sqw_sharded": (_i, [_vp, _vp, _vp, _i, _vp, _i64, _dp, _i, _dp, _i, _i, _d, _d, _i, _i, _u64, _dp]),
    "sd_dot_sharded": (_i, [_vp, _vp, _i, _vp, _vp, _i64, _dp]),
    "sd_model_set_shard_mode": (_i, [_vp, _i, _i, _i]),
    "sd_model_local_tiles": (_i, [_vp, _i64p, _i64p, _ip]),
    "sd_model_shard_pack_list": (_i, [_vp, _i64p, _i64p, _ip]),
    "sd_shard_pack_dev": (_i, [_vp, _vp, _i, _vp, _vp]),
    "sd_fill_randn_local_dev": (_i, [_vp, _vp, _i, _vp, _u64]),
    "sd_model_shard_info": (_i, [_vp, C.POINTER(sd_shard_info)]),
    "sd_model_shard_slabs": (_i, [_vp, C.POINTER(sd_slab), C.POINTER(sd_slab)]),
}

_lib = None


def lib():
    """The loaded library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950).  There is no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own HIP/HSA runtime under the same sonames as /opt/rocm.  Two HIP
        # runtimes cannot share a process, so when torch is installed let it load first: libspindyn then binds
        # to the runtime torch brought (device tensors from torch are usable either way).
        if "torch" not in sys.modules and not os.environ.get("SD_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, ctx=None):
    if rc == SD_OK:
        return
    msg = lib().sd_last_error(ctx)
    msg = msg.decode() if msg else lib().sd_status_string(rc).decode()
    if rc == SD_EARG:
        raise ArgumentError(msg)
    if rc == SD_EDIM:
        raise DimensionMismatch(msg)
    if rc == SD_EZERO:
        raise ZeroNormError(msg)
    raise SpinDynError(rc, msg)


class Context:
    """One device + stream + scratch (sd_ctx)."""

    def __init__(self, device=0):
        self.h = _vp()
        check(lib().sd_ctx_create(device, C.byref(self.h)))
        self.device = device
        self.kpm_doubling = True      # mirror of the library's per-context flag (sd_ctx_set_kpm_doubling)
        self._apply_cb, self._apply_owner = None, None   # a caller's operator (trampoline, weakref to the installing model)

    def set_stream(self, stream_ptr):
        check(lib().sd_ctx_set_stream(self.h, _vp(stream_ptr)), self.h)

    def set_kpm_doubling(self, on):
        """True (default): two Chebyshev moments per apply; False: the reference's one-moment-per-apply loop."""
        check(lib().sd_ctx_set_kpm_doubling(self.h, 1 if on else 0), self.h)
        self.kpm_doubling = bool(on)

    def set_kpm_pair_q(self, on):
        """True (default): kpm_sqw of a real psi0 computes each pair of momenta (q, 2 pi - q) once; False: every q on its own,
        as the reference does."""
        check(lib().sd_ctx_set_kpm_pair_q(self.h, 1 if on else 0), self.h)

    def set_q_batch(self, on):
        """True (default): the momenta of kpm_sqw / lanczos_sqw share the launches of their recursions at launch-bound sizes
        (bit-identical S); False: one momentum at a time."""
        check(lib().sd_ctx_set_q_batch(self.h, 1 if on else 0), self.h)

    def set_gs_blocked(self, on):
        """True (default): lanczos_groundstate re-orthogonalises in blocks of 8 columns; False: column by column (reference order)."""
        check(lib().sd_ctx_set_gs_blocked(self.h, 1 if on else 0), self.h)

    def install_apply(self, cb, owner_ref):
        """Install a ctypes APPLY_FN trampoline as this context's recursion-level operator (sd_ctx_set_apply_callback) and keep
        it alive here, with a weak reference to the model that installed it (model.Model.set_apply)."""
        check(lib().sd_ctx_set_apply_callback(self.h, cb, None), self.h)
        self._apply_cb, self._apply_owner = cb, owner_ref

    def clear_apply(self):
        """Back to the built-in operator; the trampoline is released only after the library has dropped its pointer."""
        if self.h:
            check(lib().sd_ctx_set_apply_callback(self.h, APPLY_FN(), None), self.h)
        self._apply_cb, self._apply_owner = None, None

    def apply_count(self):
        """Operator applications the recursion-level calls have queued on this context so far (one per recursion step)."""
        return int(lib().sd_ctx_apply_count(self.h))

    def release_scratch(self):
        """Free the staging buffers kept between host-pointer apply calls (re-created on demand)."""
        check(lib().sd_ctx_release_scratch(self.h), self.h)

    def synchronize(self):
        check(lib().sd_ctx_synchronize(self.h), self.h)

    def close(self):
        if self.h:
            lib().sd_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    """Process-wide context on LOCAL_RANK's device (one process per GPU)."""
    global _default_ctx
    if _default_ctx is None:
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        if dev >= lib().sd_device_count():
            dev = 0
        _default_ctx = Context(dev)
    return _default_ctx
