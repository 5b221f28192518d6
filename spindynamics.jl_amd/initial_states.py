"""Mirror of the reference's InitialStates module (src/InitialStates.jl): one-hot Float64 vectors whose position is
found with the closed-form rank (no states[] / idxmap).  `device=` returns a torch CUDA tensor instead of numpy (at
L=36 a host one-hot vector would be 72 GB)."""
import ctypes as C

import numpy as np

from ._lib import check, lib

DOMAIN_WALL, NEEL, POLARIZED_UP, POLARIZED_DOWN, POLARIZED_FLIPS = range(5)


def state_index(model, kind, flips=()):
    f = np.array(list(flips), dtype=np.int32)
    idx = C.c_int64()
    check(lib().sd_initial_state_index(model.h, kind, f.ctypes.data_as(C.POINTER(C.c_int)), len(f), C.byref(idx)),
          model.ctx.h if model.ctx else None)
    return idx.value


def _one_hot(model, idx, device):
    if device is None:
        psi0 = np.zeros(model.N)
        psi0[idx] = 1.0
        return psi0
    import torch
    psi0 = torch.zeros(model.N, dtype=torch.float64, device=device)
    psi0[idx] = 1.0
    return psi0


def domain_wall_state(model, device=None):
    """src/InitialStates.jl:9-34"""
    return _one_hot(model, state_index(model, DOMAIN_WALL), device)


def neel_state(model, device=None):
    """src/InitialStates.jl:40-63"""
    return _one_hot(model, state_index(model, NEEL), device)


def polarized_state(model, up=True, device=None):
    """src/InitialStates.jl:70-89"""
    return _one_hot(model, state_index(model, POLARIZED_UP if up else POLARIZED_DOWN), device)


def polarized_state_with_flips(model, flips, device=None):
    """src/InitialStates.jl:97-130 (flips: 1-based sites)"""
    return _one_hot(model, state_index(model, POLARIZED_FLIPS, flips), device)
