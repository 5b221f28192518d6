"""Mirror of the reference's Observables module (src/Observables.jl): device reductions over |psi|^2.
psi may be a numpy array (host) or a torch CUDA tensor (stays on the device)."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .hamiltonian import _bind_torch_stream, _dtype_code, _is_torch

_dp = C.POINTER(C.c_double)


def _call(name, psi, model, nout):
    outs = [np.empty(model.L) for _ in range(nout)]
    ptrs = [o.ctypes.data_as(_dp) for o in outs]
    if _is_torch(psi):
        _bind_torch_stream(model, psi)
        check(getattr(lib(), name + "_dev")(model.ctx.h, model.h, _dtype_code(psi), psi.data_ptr(), len(psi), *ptrs), model.ctx.h)
    else:
        x = np.ascontiguousarray(psi)
        if x.dtype not in (np.float64, np.complex128):
            x = x.astype(np.complex128 if np.iscomplexobj(x) else np.float64)
        check(getattr(lib(), name)(model.ctx.h, model.h, _dtype_code(x), x.ctypes.data, len(x), *ptrs), model.ctx.h)
    return outs


def magnetization_per_site(psi, model):
    """magnetization_per_site(psi, model) -> <S^z_i>, i = 1..L -- src/Observables.jl:14-36"""
    return _call("sd_magnetization", psi, model, 1)[0]


def connected_correlations(psi, model):
    """connected_correlations(psi, model) -> C_r, r = 0..L-1 -- src/Observables.jl:44-94"""
    return _call("sd_connected_correlations", psi, model, 1)[0]


def structure_factor_Sq(psi, model):
    """structure_factor_Sq(psi, model) -> Dict{q => S(q)}, q = 2 pi (n-1)/L -- src/Observables.jl:100-109"""
    q, S = _call("sd_structure_factor", psi, model, 2)
    return {float(a): float(b) for a, b in zip(q, S)}
