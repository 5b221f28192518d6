"""ctypes front-end of oracle/spin_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see the header of spin_oracle.c).  The product package
(spindynamics.jl_amd) never does.

Function names mirror the reference's Julia names (src/Hamiltonian.jl,
src/Lanczos.jl, src/TimeEvolution/*.jl, src/KPM_Sqw.jl, src/LanczosSqw.jl).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libspin_oracle.so")

_i32p = C.POINTER(C.c_int)
_f64p = C.POINTER(C.c_double)
_u64p = C.POINTER(C.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, "spin_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.so_binomial.restype = C.c_int64
        _lib.so_dim.restype = C.c_int64
        _lib.so_lookup.restype = C.c_int64
        _lib.so_lookup.argtypes = [C.c_void_p, C.c_uint64]
        _lib.so_bit_at.restype = C.c_uint64
        _lib.so_bit_at.argtypes = [C.c_uint64, C.c_int]
        _lib.so_flip_bits.restype = C.c_uint64
        _lib.so_flip_bits.argtypes = [C.c_uint64, C.c_int, C.c_int]
        _lib.so_sz_value.restype = C.c_double
        _lib.so_sz_value.argtypes = [C.c_uint64]
    return _lib


class OracleError(Exception):
    def __init__(self, code):
        super().__init__({1: "ArgumentError", 2: "DimensionMismatch", 3: "zero norm", 4: "out of memory"}.get(code, str(code)))
        self.code = code


def _chk(rc):
    if rc != 0:
        raise OracleError(rc)


def _dp(a):
    return a.ctypes.data_as(_f64p)


def _as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _vec(psi):
    """-> (contiguous array viewed as float64, nc)"""
    psi = np.ascontiguousarray(psi)
    if np.iscomplexobj(psi):
        psi = psi.astype(np.complex128, copy=False)
        return psi, 2
    psi = psi.astype(np.float64, copy=False)
    return psi, 1


class Model:
    """Mirror of SpinModel.Model (src/SpinModel.jl:6-15) backed by so_model."""

    def __init__(self, L, nup=None, hopping=(), onsite_field=None, zz=()):
        l = lib()
        self.L = L
        self.nup = nup
        self.mode = "full" if nup is None else "sector"
        self.hopping_list = [(int(i), int(j), float(J)) for (i, j, J) in hopping]
        self.zz_list = [(int(i), int(j), float(J)) for (i, j, J) in zz]
        self.onsite_field = np.zeros(L) if onsite_field is None else _as_f64(onsite_field)
        hi = np.array([h[0] for h in self.hopping_list], dtype=np.int32)
        hj = np.array([h[1] for h in self.hopping_list], dtype=np.int32)
        hJ = np.array([h[2] for h in self.hopping_list], dtype=np.float64)
        zi = np.array([h[0] for h in self.zz_list], dtype=np.int32)
        zj = np.array([h[1] for h in self.zz_list], dtype=np.int32)
        zJ = np.array([h[2] for h in self.zz_list], dtype=np.float64)
        self._h = C.c_void_p()
        _chk(l.so_model_create(C.c_int(L), C.c_int(-1 if nup is None else nup),
                               C.c_int(len(hi)), hi.ctypes.data_as(_i32p), hj.ctypes.data_as(_i32p), _dp(hJ),
                               C.c_int(len(zi)), zi.ctypes.data_as(_i32p), zj.ctypes.data_as(_i32p), _dp(zJ),
                               _dp(self.onsite_field), C.byref(self._h)))
        self.N = int(l.so_dim(self._h))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().so_model_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def states(self):
        out = np.empty(self.N, dtype=np.uint64)
        lib().so_states(self._h, out.ctypes.data_as(_u64p))
        return out

    def lookup(self, state):
        """get(model.idxmap, state, 0) -- 1-based, 0 when absent."""
        return int(lib().so_lookup(self._h, C.c_uint64(int(state))))


def build_model(L, nup=None, hopping=(), onsite_field=None, zz=()):
    return Model(L, nup, hopping, onsite_field, zz)


def XXZChain(L, Jxy=1.0, Jz=1.0, hz=0.0, nup=None, boundary="open"):
    """src/SpinModel.jl:63-90"""
    hopping = [(i, i + 1, float(Jxy) / 2) for i in range(1, L)]
    zz = [(i, i + 1, float(Jz)) for i in range(1, L)]
    if boundary == "periodic":
        if L > 2:
            hopping.append((L, 1, float(Jxy) / 2))
            zz.append((L, 1, float(Jz)))
    elif boundary != "open":
        raise OracleError(1)
    return Model(L, nup, hopping, np.full(L, float(hz)), zz)


def momenta(model):
    """src/SpinModel.jl:97-99"""
    return 2 * np.pi * np.arange(model.L) / model.L


def apply_H(model, psi):
    psi, nc = _vec(psi)
    out = np.empty_like(psi)
    _chk(lib().so_apply_H(model._h, C.c_int(nc), _dp(out.view(np.float64)), _dp(psi.view(np.float64)), C.c_int64(psi.shape[0])))
    return out


def apply_rescaled_H(model, psi, a, b):
    psi, nc = _vec(psi)
    out = np.empty_like(psi)
    _chk(lib().so_apply_rescaled_H(model._h, C.c_int(nc), _dp(out.view(np.float64)), _dp(psi.view(np.float64)),
                                   C.c_int64(psi.shape[0]), C.c_double(a), C.c_double(b)))
    return out


def Sz_q_vector(model, psi0, q):
    psi0, nc = _vec(psi0)
    phi = np.empty(psi0.shape[0], dtype=np.complex128)
    _chk(lib().so_szq(model._h, C.c_int(nc), _dp(psi0.view(np.float64)), C.c_int64(psi0.shape[0]), C.c_double(q),
                      _dp(phi.view(np.float64))))
    return phi


def spin_operator(model, site, op_type, psi):
    """create_spin_operator(site, op_type)(psi, model) -- src/Hamiltonian.jl:49-136, in the reference's scatter form
    (numpy loops; small models only).  op_type in {"z","plus","minus","x","y"}."""
    if site < 1 or op_type not in ("z", "plus", "minus", "x", "y"):
        raise OracleError(1)
    if site > model.L:
        raise OracleError(1)
    psi = np.asarray(psi)
    states = model.states
    if len(psi) != len(states):
        raise OracleError(2)
    if model.nup is not None and op_type != "z":
        raise OracleError(1)
    bit = site - 1
    res = np.zeros(len(psi), dtype=psi.dtype)   # zeros(T, length(psi)): a complex update of a real result is an error upstream
    for idx, st in enumerate(states):
        st = int(st)
        cur = (st >> bit) & 1
        if op_type == "z":
            res[idx] = (0.5 if cur else -0.5) * psi[idx]
            continue
        if op_type == "plus" and cur != 0:
            continue
        if op_type == "minus" and cur != 1:
            continue
        j = model.lookup(st ^ (1 << bit))
        if j == 0:
            continue
        if op_type in ("plus", "minus"):
            res[j - 1] += psi[idx]
        elif op_type == "x":
            res[j - 1] += 0.5 * psi[idx]
        else:
            if not np.iscomplexobj(res):
                raise TypeError("InexactError: S^y of a Float64 vector")
            res[j - 1] += (-0.5j if cur == 0 else 0.5j) * psi[idx]
    return res


def symtridiag_eig(d, e, vectors=True):
    d = _as_f64(d)
    e = _as_f64(e)
    n = len(d)
    w = np.empty(n)
    z = np.empty((n, n), order="F") if vectors else None
    _chk(lib().so_symtridiag_eig(C.c_int(n), _dp(d), _dp(e) if n > 1 else None, _dp(w), _dp(z) if vectors else None))
    return (w, z) if vectors else w


def lanczos_extremal(model, psi0, lanc_m=100, tol=1e-12, negate=False):
    psi0 = np.ascontiguousarray(psi0, dtype=np.complex128)
    lo, hi, mu = C.c_double(), C.c_double(), C.c_int()
    _chk(lib().so_lanczos_extremal(model._h, C.c_int(lanc_m), C.c_double(tol), _dp(psi0.view(np.float64)),
                                   C.c_int(int(negate)), C.byref(lo), C.byref(hi), C.byref(mu)))
    return lo.value, hi.value


def estimate_energy_bounds(model, psi0_a, psi0_b, lanc_m=80):
    a = np.ascontiguousarray(psi0_a, dtype=np.complex128)
    b = np.ascontiguousarray(psi0_b, dtype=np.complex128)
    lo, hi = C.c_double(), C.c_double()
    _chk(lib().so_estimate_energy_bounds(model._h, C.c_int(lanc_m), _dp(a.view(np.float64)), _dp(b.view(np.float64)),
                                         C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def lanczos_groundstate(model, psi0, lanc_m=100, tol=1e-12, orthogonalize_tol=1e-10):
    psi0 = _as_f64(psi0)
    E0, ma = C.c_double(), C.c_int()
    gs = np.empty(model.N)
    _chk(lib().so_lanczos_groundstate(model._h, C.c_int(lanc_m), C.c_double(tol), C.c_double(orthogonalize_tol),
                                      _dp(psi0), C.byref(E0), _dp(gs), C.byref(ma)))
    return E0.value, gs


def lanczos_tridiag(model, v, lanc_m=100, tol=1e-12):
    v = np.ascontiguousarray(v, dtype=np.complex128)
    n = v.shape[0]
    m = min(lanc_m, n)
    alpha = np.zeros(m)
    beta = np.zeros(max(m - 1, 1))
    me, nv = C.c_int(), C.c_double()
    _chk(lib().so_lanczos_tridiag(model._h, _dp(v.view(np.float64)), C.c_int64(n), C.c_int(lanc_m), C.c_double(tol),
                                  _dp(alpha), _dp(beta), C.byref(me), C.byref(nv)))
    return alpha[:me.value].copy(), beta[:max(me.value - 1, 0)].copy(), nv.value


def krylov_time_evolve(model, psi0, dt, kry_m=30):
    psi0, nc = _vec(psi0)
    out = np.empty(psi0.shape[0], dtype=np.complex128)
    _chk(lib().so_krylov_time_evolve(model._h, C.c_int(nc), _dp(psi0.view(np.float64)), C.c_double(dt), C.c_int(kry_m),
                                     _dp(out.view(np.float64))))
    return out


def chebyshev_coeffs(cheb_n, a, b, dt):
    c = np.empty(cheb_n, dtype=np.complex128)
    lib().so_chebyshev_coeffs(C.c_int(cheb_n), C.c_double(a), C.c_double(b), C.c_double(dt), _dp(c.view(np.float64)))
    return c


def chebyshev_time_evolve(model, psi0, dt, cheb_n=100, Ebounds=(-1.0, 1.0)):
    psi0 = np.ascontiguousarray(psi0, dtype=np.complex128)
    out = np.empty_like(psi0)
    _chk(lib().so_chebyshev_time_evolve(model._h, _dp(psi0.view(np.float64)), C.c_double(dt), C.c_int(cheb_n),
                                        C.c_double(Ebounds[0]), C.c_double(Ebounds[1]), _dp(out.view(np.float64))))
    return out


def rescaling_from_bounds(Emin, Emax):
    a, b = C.c_double(), C.c_double()
    lib().so_rescaling_from_bounds(C.c_double(Emin), C.c_double(Emax), C.byref(a), C.byref(b))
    return a.value, b.value


_KERNELS = {"jackson": 0, "lorentz": 1}


def get_kernel(M, kernel="jackson"):
    g = np.empty(M)
    lib().so_get_kernel(C.c_int(M), C.c_int(_KERNELS.get(kernel, 2)), _dp(g))
    return g


def compute_chebyshev_moments(model, phi, M, a, b):
    phi = np.ascontiguousarray(phi, dtype=np.complex128)
    mu = np.empty(M)
    _chk(lib().so_chebyshev_moments(model._h, _dp(phi.view(np.float64)), C.c_int(M), C.c_double(a), C.c_double(b), _dp(mu)))
    return mu


def kpm_reconstruct(mu_damped, omega, a, b, E0):
    mu = _as_f64(mu_damped)
    om = _as_f64(omega)
    S = np.empty(len(om))
    lib().so_kpm_reconstruct(_dp(mu), C.c_int(len(mu)), _dp(om), C.c_int(len(om)), C.c_double(a), C.c_double(b),
                             C.c_double(E0), _dp(S))
    return S


def kpm_sqw(model, psi0, q_list, omega, a, b, kpm_m=200, kernel="jackson"):
    psi0, nc = _vec(psi0)
    q = _as_f64(q_list)
    om = _as_f64(omega)
    S = np.empty((len(q), len(om)))
    _chk(lib().so_kpm_sqw(model._h, C.c_int(nc), _dp(psi0.view(np.float64)), _dp(q), C.c_int(len(q)), _dp(om),
                          C.c_int(len(om)), C.c_double(a), C.c_double(b), C.c_int(kpm_m),
                          C.c_int(_KERNELS.get(kernel, 2)), _dp(S)))
    return S


def spectral_from_tridiagonal(alpha, beta, norm_phi, E0, omega, eta=0.05, broaden="lorentz"):
    al = _as_f64(alpha)
    be = _as_f64(beta)
    om = _as_f64(omega)
    S = np.empty(len(om))
    _chk(lib().so_spectral_from_tridiagonal(_dp(al), _dp(be) if len(be) else None, C.c_int(len(al)), C.c_double(norm_phi),
                                            C.c_double(E0), _dp(om), C.c_int(len(om)), C.c_double(eta),
                                            C.c_int({"lorentz": 0, "gauss": 1}.get(broaden, 2)), _dp(S)))
    return S


def lanczos_sqw(model, psi0, q_list, omega, lanc_m=200, eta=0.05, broaden="lorentz"):
    psi0, nc = _vec(psi0)
    q = _as_f64(q_list)
    om = _as_f64(omega)
    S = np.empty((len(q), len(om)))
    _chk(lib().so_lanczos_sqw(model._h, C.c_int(nc), _dp(psi0.view(np.float64)), _dp(q), C.c_int(len(q)), _dp(om),
                              C.c_int(len(om)), C.c_int(lanc_m), C.c_double(eta),
                              C.c_int({"lorentz": 0, "gauss": 1}.get(broaden, 2)), _dp(S)))
    return S


def magnetization_per_site(psi, model):
    psi, nc = _vec(psi)
    out = np.empty(model.L)
    _chk(lib().so_magnetization_per_site(model._h, C.c_int(nc), _dp(psi.view(np.float64)), C.c_int64(psi.shape[0]), _dp(out)))
    return out


def connected_correlations(psi, model):
    psi, nc = _vec(psi)
    out = np.empty(model.L)
    _chk(lib().so_connected_correlations(model._h, C.c_int(nc), _dp(psi.view(np.float64)), C.c_int64(psi.shape[0]), _dp(out)))
    return out


def structure_factor_Sq(psi, model):
    psi, nc = _vec(psi)
    q, S = np.empty(model.L), np.empty(model.L)
    _chk(lib().so_structure_factor_Sq(model._h, C.c_int(nc), _dp(psi.view(np.float64)), C.c_int64(psi.shape[0]), _dp(q), _dp(S)))
    return {float(a): float(b) for a, b in zip(q, S)}


def _initial(model, kind, flips=()):
    f = np.array(list(flips), dtype=np.int32)
    out = np.empty(model.N)
    _chk(lib().so_initial_state(model._h, C.c_int(kind), f.ctypes.data_as(_i32p), C.c_int(len(f)), _dp(out)))
    return out


def domain_wall_state(model):
    return _initial(model, 0)


def neel_state(model):
    return _initial(model, 1)


def polarized_state(model, up=True):
    return _initial(model, 2 if up else 3)


def polarized_state_with_flips(model, flips):
    return _initial(model, 4, flips)


def num_threads():
    return int(lib().so_num_threads())


def set_num_threads(n):
    lib().so_set_num_threads(C.c_int(int(n)))
