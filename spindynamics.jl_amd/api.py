"""Mirror of the reference's PublicAPI module (src/PublicAPI.jl:25-155): same
function names, keyword arguments and dispatch on `method`; Julia Symbols are
Python strings ("lanczos", "krylov", "chebyshev", "kpm")."""
import numpy as np

from ._lib import ArgumentError
from .hamiltonian import apply_H
from . import solvers


def groundstate(model, method="lanczos", **kwargs):
    """groundstate(model; method=:lanczos, kwargs...) -> (E0, psi) -- src/PublicAPI.jl:25-35"""
    if method == "lanczos":
        return solvers.lanczos_groundstate(apply_H, model, **kwargs)
    raise ArgumentError(f"unsupported ground-state method: {method}")


def time_evolve(model, psi0, t, method="krylov", Ebounds=None, **kwargs):
    """time_evolve(model, psi0, t; method=:krylov, Ebounds=nothing, kwargs...) -- src/PublicAPI.jl:50-88

    method="chebyshev" without Ebounds: the bounds of the built-in H are estimated once per (model, seed) and kept with the
    model (the reference re-estimates them on every call); a caller's operator on the model's context bypasses the cache."""
    if method == "krylov":
        return solvers.krylov_time_evolve(psi0, float(t), apply_H, model, **kwargs)
    if method == "chebyshev":
        seed = kwargs.pop("seed", 0)
        if Ebounds is None:
            # the reference estimates the bounds anew on every call (two 80-step Lanczos runs, src/PublicAPI.jl:68-75).  Here the
            # start vectors come from the counter-based generator of `seed`, so the estimate for a (model, seed) is the same
            # every time: it is computed once and kept with the model -- a loop of time steps pays for it once (at L=32: 3 s).
            # The cache describes the BUILT-IN operator only: with a caller's operator installed on the model's context
            # (model.set_apply, which also drops the cache) the bounds are estimated anew, as the reference does.
            if getattr(model.ctx, "_apply_cb", None) is not None:
                bounds = solvers.estimate_energy_bounds(apply_H, model, seed=seed)
            else:
                cache = model.__dict__.setdefault("_energy_bounds", {})
                if seed not in cache:
                    cache[seed] = solvers.estimate_energy_bounds(apply_H, model, seed=seed)
                bounds = cache[seed]
        else:
            bounds = Ebounds
        return solvers.chebyshev_time_evolve(psi0, float(t), apply_H, model, Ebounds=bounds, **kwargs)
    raise ArgumentError(f"unsupported time-evolution method: {method}")


def structure_factor(model, psi):
    """structure_factor(model, psi) -> same mapping as structure_factor_Sq -- src/PublicAPI.jl:101-106"""
    from .observables import structure_factor_Sq
    return structure_factor_Sq(psi, model)


def dynamical_structure_factor(model, psi0, q, omega, method="lanczos", **kwargs):
    """dynamical_structure_factor(model, psi0, q, omega; method=:lanczos, kwargs...) -> S[len(q), len(omega)]
    -- src/PublicAPI.jl:122-155"""
    q_list = np.asarray(q, dtype=np.float64)
    w = np.asarray(omega, dtype=np.float64)
    if method == "lanczos":
        return solvers.lanczos_sqw(psi0, model, q_list, w, **kwargs)
    if method == "kpm":
        return solvers.kpm_sqw(psi0, model, q_list, w, **kwargs)
    raise ArgumentError(f"unsupported dynamical structure-factor method: {method}")
