"""Mirrors of the reference's solver functions; each runs its whole recursion on
the device through one C-ABI call (include/spindyn.h, "recursion level").

The `applyH` argument of the reference functions is honoured: hamiltonian.apply_H selects the built-in fused
operator; any other callable `applyH(out, psi, model)` on torch device tensors is installed as the operator of the
recursion for the duration of the call (sd_ctx_set_apply_callback: the library calls it for out <- H psi and
applies its fused step as a second pass).
Where the reference draws a start vector from Julia's RNG (which cannot be
reproduced outside Julia) a `psi0=` / `seed=` keyword is offered instead of
`rng=`.
"""
import ctypes as C
import functools
import inspect

import numpy as np

from . import _lib
from ._lib import ArgumentError, DimensionMismatch, SD_C128, SD_F64, check, lib
from .hamiltonian import _bind_torch_stream, _is_torch, apply_H

_dp = C.POINTER(C.c_double)


def _with_operator(f):
    """Runs f with its `applyH` argument as the operator of `model`: nothing to do for the built-in apply_H, any other
    callable is installed as the recursion-level operator for the duration of the call; its own exception, if any, is
    what the caller sees."""
    sig = inspect.signature(f)

    @functools.wraps(f)
    def g(*args, **kw):
        ba = sig.bind(*args, **kw)
        applyH, model = ba.arguments["applyH"], ba.arguments["model"]
        if applyH is apply_H:
            return f(*args, **kw)
        if not callable(applyH):
            raise ArgumentError("applyH must be callable: applyH(out, psi, model)")
        ctx = model.ctx
        before = (ctx._apply_cb, ctx._apply_owner) if ctx is not None else (None, None)   # an operator installed with model.set_apply
        model.set_apply(applyH)
        try:
            return f(*args, **kw)
        except Exception:
            if getattr(model, "_apply_err", None) is not None:
                raise model._apply_err
            raise
        finally:
            model.set_apply(None)
            if before[0] is not None:
                ctx.install_apply(*before)
    return g


def _c128(x, n=None, name="vector"):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    if n is not None and len(x) != n:
        raise DimensionMismatch(f"{name} has length {len(x)}, expected {n}")
    return x


def _vec(x):
    x = np.ascontiguousarray(x)
    if np.iscomplexobj(x):
        return x.astype(np.complex128, copy=False), SD_C128
    return x.astype(np.float64, copy=False), SD_F64


def _ptr(x):
    return None if x is None else x.ctypes.data


@_with_operator
def lanczos_extremal(applyH, model, lanc_m=100, tol=1e-12, psi0=None, seed=0, negate=False):
    """lanczos_extremal(applyH!, model; lanc_m, tol, rng) -> (Emin, Emax) -- src/Lanczos.jl:27-84"""
    p0 = None if psi0 is None else _c128(psi0, model.N, "psi0")
    lo, hi = C.c_double(), C.c_double()
    check(lib().sd_lanczos_extremal(model.ctx.h, model.h, int(lanc_m), float(tol), _ptr(p0), int(seed), int(bool(negate)),
                                    C.byref(lo), C.byref(hi)), model.ctx.h)
    return lo.value, hi.value


@_with_operator
def estimate_energy_bounds(applyH, model, lanc_m=80, psi0_a=None, psi0_b=None, seed=0):
    """estimate_energy_bounds(applyH!, model; lanc_m=80) -> (Emin, Emax) -- src/Lanczos.jl:255-271"""
    a = None if psi0_a is None else _c128(psi0_a, model.N, "psi0_a")
    b = None if psi0_b is None else _c128(psi0_b, model.N, "psi0_b")
    lo, hi = C.c_double(), C.c_double()
    check(lib().sd_energy_bounds(model.ctx.h, model.h, int(lanc_m), _ptr(a), _ptr(b), int(seed), C.byref(lo), C.byref(hi)),
          model.ctx.h)
    return lo.value, hi.value


@_with_operator
def lanczos_groundstate(applyH, model, lanc_m=100, tol=1e-12, orthogonalize_tol=1e-10, psi0=None, seed=0):
    """lanczos_groundstate(applyH!, model; lanc_m, tol, orthogonalize_tol, rng) -> (E0, psi_gs) -- src/Lanczos.jl:87-181"""
    p0 = None
    if psi0 is not None:
        p0 = np.ascontiguousarray(psi0, dtype=np.float64)
        if len(p0) != model.N:
            raise DimensionMismatch("psi0 length")
    E0, ma = C.c_double(), C.c_int()
    gs = np.empty(model.N, dtype=np.float64)
    check(lib().sd_lanczos_groundstate(model.ctx.h, model.h, int(lanc_m), float(tol), float(orthogonalize_tol),
                                       None if p0 is None else p0.ctypes.data_as(_dp), int(seed), C.byref(E0),
                                       gs.ctypes.data_as(_dp), C.byref(ma)), model.ctx.h)
    return E0.value, gs


@_with_operator
def lanczos_tridiag(applyH, model, v, lanc_m=100, tol=1e-12):
    """lanczos_tridiag(applyH!, model, v; lanc_m, tol) -> (alpha, beta, norm_v) -- src/Lanczos.jl:196-246"""
    v = _c128(v)
    n = len(v)
    m = max(min(int(lanc_m), n), 1)
    alpha, beta = np.zeros(m), np.zeros(max(m - 1, 1))
    me, nv = C.c_int(), C.c_double()
    check(lib().sd_lanczos_tridiag(model.ctx.h, model.h, v.ctypes.data, n, int(lanc_m), float(tol), alpha.ctypes.data_as(_dp),
                                   beta.ctypes.data_as(_dp), C.byref(me), C.byref(nv)), model.ctx.h)
    return alpha[: me.value].copy(), beta[: max(me.value - 1, 0)].copy(), nv.value


@_with_operator
def krylov_time_evolve(psi0, dt, applyH, model, kry_m=30):
    """krylov_time_evolve(psi0, dt, applyH!, model; kry_m) -> psi(t) ComplexF64, normalised -- src/TimeEvolution/Krylov.jl:136-192"""
    if _is_torch(psi0):          # device-resident state
        import torch
        if psi0.dtype not in (torch.float64, torch.complex128):
            raise ArgumentError("vectors must be float64 or complex128")
        _bind_torch_stream(model, psi0)
        out = torch.empty(len(psi0), dtype=torch.complex128, device=psi0.device)
        check(lib().sd_krylov_evolve_dev(model.ctx.h, model.h, SD_C128 if psi0.is_complex() else SD_F64, psi0.data_ptr(),
                                         len(psi0), float(dt), int(kry_m), out.data_ptr()), model.ctx.h)
        return out
    x, code = _vec(psi0)
    out = np.empty(len(x), dtype=np.complex128)
    check(lib().sd_krylov_evolve(model.ctx.h, model.h, code, x.ctypes.data, len(x), float(dt), int(kry_m), out.ctypes.data),
          model.ctx.h)
    return out


@_with_operator
def chebyshev_time_evolve(psi0, dt, applyH, model, cheb_n=100, Ebounds=(-1.0, 1.0), workspace=None):
    """chebyshev_time_evolve(psi0, dt, applyH!, model; cheb_n, Ebounds) -- src/TimeEvolution/Chebyshev.jl:61-124.
    psi0 must be complex (the reference's workspace is typed by psi0 and receives complex coefficients)."""
    if _is_torch(psi0):          # device-resident state: no PCIe per step of a time evolution
        import torch
        if psi0.dtype != torch.complex128:
            raise ArgumentError("chebyshev_time_evolve needs a ComplexF64 psi0 (as the reference does)")
        if int(cheb_n) < 1:
            raise AssertionError("cheb_n must be >= 1")
        _bind_torch_stream(model, psi0)
        out = torch.empty_like(psi0)
        check(lib().sd_chebyshev_evolve_dev(model.ctx.h, model.h, psi0.data_ptr(), len(psi0), float(dt), int(cheb_n),
                                            float(Ebounds[0]), float(Ebounds[1]), out.data_ptr()), model.ctx.h)
        return out
    if not np.iscomplexobj(psi0):
        raise ArgumentError("chebyshev_time_evolve needs a ComplexF64 psi0 (as the reference does)")
    if int(cheb_n) < 1:
        raise AssertionError("cheb_n must be >= 1")
    x = _c128(psi0)
    out = np.empty(len(x), dtype=np.complex128)
    check(lib().sd_chebyshev_evolve(model.ctx.h, model.h, x.ctypes.data, len(x), float(dt), int(cheb_n), float(Ebounds[0]),
                                    float(Ebounds[1]), out.ctypes.data), model.ctx.h)
    return out


def chebyshev_coeffs(cheb_n, a, b, dt):
    c = np.empty(int(cheb_n), dtype=np.complex128)
    check(lib().sd_chebyshev_coeffs(int(cheb_n), float(a), float(b), float(dt), c.ctypes.data_as(_dp)))
    return c


def rescaling_from_bounds(Emin, Emax):
    """_rescaling_from_bounds -- src/KPM_Sqw.jl:13-17"""
    a, b = C.c_double(), C.c_double()
    check(lib().sd_kpm_rescaling_from_bounds(float(Emin), float(Emax), C.byref(a), C.byref(b)))
    return a.value, b.value


def get_rescaling_params(applyH, model, lanc_m=80, seed=0):
    """get_rescaling_params -- src/KPM_Sqw.jl:25-28"""
    return rescaling_from_bounds(*estimate_energy_bounds(applyH, model, lanc_m=lanc_m, seed=seed))


def get_kernel(M, kernel="jackson"):
    """get_kernel(M, kernel) -- src/KPM_Sqw.jl:131-145"""
    g = np.empty(int(M))
    check(lib().sd_kpm_kernel(int(M), _lib.KERNELS.get(kernel, 2), g.ctypes.data_as(_dp)))
    return g


@_with_operator
def compute_chebyshev_moments(applyH, phi, M, a, b, model):
    """compute_chebyshev_moments(apply_H!, phi, M, a, b, model) -- src/KPM_Sqw.jl:95-128"""
    phi = _c128(phi)
    mu = np.empty(int(M))
    check(lib().sd_kpm_moments(model.ctx.h, model.h, phi.ctypes.data, len(phi), int(M), float(a), float(b),
                               mu.ctypes.data_as(_dp)), model.ctx.h)
    return mu


def kpm_reconstruct(mu_damped, omega, a, b, E0):
    mu = np.ascontiguousarray(mu_damped, dtype=np.float64)
    om = np.ascontiguousarray(omega, dtype=np.float64)
    S = np.empty(len(om))
    check(lib().sd_kpm_reconstruct(mu.ctypes.data_as(_dp), len(mu), om.ctypes.data_as(_dp), len(om), float(a), float(b),
                                   float(E0), S.ctypes.data_as(_dp)))
    return S


def kpm_sw(phi, applyH, model, omega, a, b, E0, kpm_m=200, kernel="jackson"):
    """kpm_sw -- src/KPM_Sqw.jl:34-93"""
    mu = compute_chebyshev_moments(applyH, phi, kpm_m, a, b, model)
    mu *= get_kernel(kpm_m, kernel)
    return kpm_reconstruct(mu, omega, a, b, E0)


def kpm_sqw(psi0, model, q_list, omega, a=None, b=None, kpm_m=200, kernel="jackson", seed=0):
    """kpm_sqw(psi0, model, q_list, omega; a, b, kpm_m, kernel) -> Smat[Qn, W] -- src/KPM_Sqw.jl:191-256"""
    x, code = _vec(psi0)
    q = np.ascontiguousarray(q_list, dtype=np.float64)
    om = np.ascontiguousarray(omega, dtype=np.float64)
    S = np.empty((len(q), len(om)))
    have = a is not None and b is not None
    check(lib().sd_kpm_sqw(model.ctx.h, model.h, code, x.ctypes.data, len(x), q.ctypes.data_as(_dp), len(q),
                           om.ctypes.data_as(_dp), len(om), int(have), float(a) if have else 0.0, float(b) if have else 0.0,
                           int(kpm_m), _lib.KERNELS.get(kernel, 2), int(seed), S.ctypes.data_as(_dp)), model.ctx.h)
    return S


def spectral_from_tridiagonal(alpha, beta, norm_phi, E0, omega, eta=0.05, broaden="lorentz"):
    """spectral_from_tridiagonal -- src/LanczosSqw.jl:18-43"""
    if broaden not in _lib.BROADEN:
        raise ArgumentError(f"unknown broadening: {broaden}")
    al = np.ascontiguousarray(alpha, dtype=np.float64)
    be = np.ascontiguousarray(beta, dtype=np.float64)
    if len(be) == 0:
        be = np.zeros(1)
    om = np.ascontiguousarray(omega, dtype=np.float64)
    S = np.empty(len(om))
    check(lib().sd_spectral_from_tridiagonal(al.ctypes.data_as(_dp), be.ctypes.data_as(_dp), len(al), float(norm_phi), float(E0),
                                             om.ctypes.data_as(_dp), len(om), float(eta), _lib.BROADEN[broaden],
                                             S.ctypes.data_as(_dp)))
    return S


def lanczos_sqw(psi0, model, q_list, omega, lanc_m=200, eta=0.05, broaden="lorentz"):
    """lanczos_sqw -- src/LanczosSqw.jl:49-80"""
    if broaden not in _lib.BROADEN:
        raise ArgumentError(f"unknown broadening: {broaden}")
    x, code = _vec(psi0)
    q = np.ascontiguousarray(q_list, dtype=np.float64)
    om = np.ascontiguousarray(omega, dtype=np.float64)
    S = np.empty((len(q), len(om)))
    check(lib().sd_lanczos_sqw(model.ctx.h, model.h, code, x.ctypes.data, len(x), q.ctypes.data_as(_dp), len(q),
                               om.ctypes.data_as(_dp), len(om), int(lanc_m), float(eta), _lib.BROADEN[broaden],
                               S.ctypes.data_as(_dp)), model.ctx.h)
    return S


def symtridiag_eig(d, e, vectors=True):
    d = np.ascontiguousarray(d, dtype=np.float64)
    e = np.ascontiguousarray(e, dtype=np.float64)
    n = len(d)
    if len(e) == 0:
        e = np.zeros(1)
    w = np.empty(n)
    z = np.empty((n, n), order="F") if vectors else None
    check(lib().sd_symtridiag_eig(n, d.ctypes.data_as(_dp), e.ctypes.data_as(_dp), w.ctypes.data_as(_dp),
                                  z.ctypes.data_as(_dp) if vectors else None))
    return (w, z) if vectors else w
