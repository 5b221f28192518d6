"""spindynamics.jl_amd -- host-side mirror of the SpinDynamics.jl interface for the
H|psi> hot path, backed by libspindyn.so (hand-written HIP for gfx950 / MI355X).

The directory name contains a dot, so load the package through
`__graft_entry__.load_package()` (registers it as `spindynamics_jl_amd`).
Importing does not need a GPU; creating a context / model does, and raises when
the library or a device is missing -- there is no CPU fallback.
"""
from ._lib import (ArgumentError, Context, check, DimensionMismatch, SpinDynError, ZeroNormError, default_context, lib,
                   LIB_PATH, PROTOTYPES)
from .model import Model, XXZChain, build_model, long_range_hopping, momenta, nn_hopping
from .hamiltonian import (Sz_q_vector, apply_H, apply_rescaled_H, bit_at, cheb_step, create_spin_operator, flip_bits,
                          sz_value)
from .solvers import (chebyshev_coeffs, chebyshev_time_evolve, compute_chebyshev_moments, estimate_energy_bounds,
                      get_kernel, get_rescaling_params, kpm_reconstruct, kpm_sqw, kpm_sw, krylov_time_evolve,
                      lanczos_extremal, lanczos_groundstate, lanczos_sqw, lanczos_tridiag, rescaling_from_bounds,
                      spectral_from_tridiagonal, symtridiag_eig)
from .observables import connected_correlations, magnetization_per_site, structure_factor_Sq
from . import initial_states
from .initial_states import domain_wall_state, neel_state, polarized_state, polarized_state_with_flips
from .api import dynamical_structure_factor, groundstate, structure_factor, time_evolve
from .dist import ShardedOperator, kpm_sqw_replicas

__all__ = [n for n in dir() if not n.startswith("_")]
