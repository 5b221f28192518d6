"""Mirror of the reference's Hamiltonian module (src/Hamiltonian.jl): the
operator-level seam  apply_H!(out, psi, model)  and friends, routed to the HIP
kernels through the C ABI.  Vectors may be numpy arrays (host: copied through
PCIe for the call) or torch CUDA tensors (device: zero copy, launched on
torch's current stream).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ArgumentError, DimensionMismatch, SD_C128, SD_F64, check, lib


def bit_at(state, i):
    """src/Hamiltonian.jl:19-21"""
    return (int(state) >> i) & 1


def sz_value(bit):
    """src/Hamiltonian.jl:23-25"""
    return 0.5 if bit == 1 else -0.5


def flip_bits(state, i, j):
    """src/Hamiltonian.jl:27-29"""
    return int(state) ^ (1 << i) ^ (1 << j)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _dtype_code(x):
    if _is_torch(x):
        import torch
        if x.dtype == torch.float64:
            return SD_F64
        if x.dtype == torch.complex128:
            return SD_C128
        raise ArgumentError("vectors must be float64 or complex128")
    if x.dtype == np.float64:
        return SD_F64
    if x.dtype == np.complex128:
        return SD_C128
    raise ArgumentError("vectors must be float64 or complex128")


def _check_host(x, name):
    if not isinstance(x, np.ndarray) or x.ndim != 1 or not x.flags["C_CONTIGUOUS"]:
        raise ArgumentError(f"{name} must be a contiguous 1-D numpy array")


def _bind_torch_stream(model, t):
    import torch
    if not t.is_cuda or not t.is_contiguous() or t.dim() != 1:
        raise ArgumentError("torch vectors must be contiguous 1-D CUDA tensors")
    model.ctx.set_stream(torch.cuda.current_stream(t.device).cuda_stream)


def apply_H(out, psi, model):
    """apply_H!(out, psi, model) -- src/Hamiltonian.jl:211-273.  Overwrites and returns `out`."""
    if len(out) != len(psi):
        raise DimensionMismatch("length(out) != length(psi)")
    code = _dtype_code(psi)
    if _dtype_code(out) != code:
        raise ArgumentError("out and psi must have the same element type")
    ctx = model.ctx
    if _is_torch(psi):
        _bind_torch_stream(model, psi)
        _bind_torch_stream(model, out)
        check(lib().sd_apply_dev(ctx.h, model.h, code, out.data_ptr(), psi.data_ptr(), len(psi)), ctx.h)
    else:
        _check_host(out, "out"); _check_host(psi, "psi")
        check(lib().sd_apply(ctx.h, model.h, code, out.ctypes.data, psi.ctypes.data, len(psi)), ctx.h)
    return out


def apply_rescaled_H(out, psi, applyH, model, a, b):
    """apply_rescaled_H!(out, psi, applyH!, model, a, b) -- src/Hamiltonian.jl:286-301.
    `applyH` must be this module's apply_H (the fused kernel computes (H psi - b psi)/a in one pass)."""
    if applyH is not apply_H:
        raise ArgumentError("apply_rescaled_H is fused with the device apply: pass apply_H")
    if len(out) != len(psi):
        raise DimensionMismatch("length(out) != length(psi)")
    code = _dtype_code(psi)
    if _dtype_code(out) != code:
        raise ArgumentError("out and psi must have the same element type")
    ctx = model.ctx
    if _is_torch(psi):
        _bind_torch_stream(model, psi)
        check(lib().sd_apply_rescaled_dev(ctx.h, model.h, code, out.data_ptr(), psi.data_ptr(), len(psi), float(a), float(b)), ctx.h)
    else:
        _check_host(out, "out"); _check_host(psi, "psi")
        check(lib().sd_apply_rescaled(ctx.h, model.h, code, out.ctypes.data, psi.ctypes.data, len(psi), float(a), float(b)), ctx.h)
    return out


def Sz_q_vector(model, psi0, q):
    """Sz_q_vector(model, psi0, q) -- src/Hamiltonian.jl:307-337.  Returns a new ComplexF64 vector."""
    ctx = model.ctx
    if _is_torch(psi0):
        import torch
        code = _dtype_code(psi0)
        _bind_torch_stream(model, psi0)
        phi = torch.empty(len(psi0), dtype=torch.complex128, device=psi0.device)
        check(lib().sd_szq_dev(ctx.h, model.h, code, psi0.data_ptr(), len(psi0), float(q), phi.data_ptr()), ctx.h)
        return phi
    psi0 = np.ascontiguousarray(psi0)
    if psi0.dtype not in (np.float64, np.complex128):
        psi0 = psi0.astype(np.complex128 if np.iscomplexobj(psi0) else np.float64)
    phi = np.empty(len(psi0), dtype=np.complex128)
    check(lib().sd_szq(ctx.h, model.h, _dtype_code(psi0), psi0.ctypes.data, len(psi0), float(q), phi.ctypes.data), ctx.h)
    return phi


_SPIN_OPS = {"z": 0, "plus": 1, "minus": 2, "x": 3, "y": 4}


def create_spin_operator(site, op_type):
    """create_spin_operator(site, op_type) -- src/Hamiltonian.jl:49-136.  Returns operator(psi, model) -> new vector.
    op_type is "z", "plus", "minus", "x" or "y" (a leading ':' as in Julia symbols is accepted)."""
    if int(site) < 1:
        raise ArgumentError("site must be at least 1")
    name = str(op_type).lstrip(":")
    if name not in _SPIN_OPS:
        raise ArgumentError(f"unsupported spin operator: {op_type}; expected :z, :plus, :minus, :x, or :y")
    code = _SPIN_OPS[name]

    def operator(psi, model):
        psi = np.ascontiguousarray(psi)
        if psi.dtype not in (np.float64, np.complex128):
            psi = psi.astype(np.complex128 if np.iscomplexobj(psi) else np.float64)
        out = np.empty_like(psi)
        check(lib().sd_spin_operator(model.ctx.h, model.h, _dtype_code(psi), int(site), code, psi.ctypes.data, len(psi),
                                     out.ctypes.data), model.ctx.h)
        return out

    return operator


def cheb_step(phi_next, phi_curr, phi_prev, psi_t, model, a, b, c):
    """One fused Chebyshev term on torch CUDA complex128 tensors
    (src/TimeEvolution/Chebyshev.jl:110-121): phi_next = 2 H~ phi_curr - phi_prev; psi_t += c phi_next."""
    _bind_torch_stream(model, phi_curr)
    c = complex(c)
    check(lib().sd_cheb_step_dev(model.ctx.h, model.h, phi_next.data_ptr(), phi_curr.data_ptr(), phi_prev.data_ptr(),
                                 psi_t.data_ptr(), len(phi_curr), float(a), float(b), c.real, c.imag), model.ctx.h)
    return phi_next
