// Device helpers shared by the apply / auxiliary kernels of libspindyn (gfx950).  Header-only, included by *.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "sd_internal.hpp"

#define SD_BIN_STRIDE 17

namespace sd_dev {

template <int NC> struct VT;
template <> struct VT<1> { using type = double; };
template <> struct VT<2> { using type = double2; };

__device__ __forceinline__ double vadd_mul(double acc, double J, double v) { return acc + J * v; }
__device__ __forceinline__ double2 vadd_mul(double2 acc, double J, double2 v) {
  return make_double2(acc.x + J * v.x, acc.y + J * v.y);
}
__device__ __forceinline__ double vscale(double d, double v) { return d * v; }
__device__ __forceinline__ double2 vscale(double d, double2 v) { return make_double2(d * v.x, d * v.y); }

__device__ __forceinline__ double sz_of(uint64_t bit) { return bit ? 0.5 : -0.5; }

// diagonal matrix element for configuration s -- src/Hamiltonian.jl:226-241
__device__ __forceinline__ double diag_of(const sd_dev_model &dm, uint64_t s) {
  if (dm.diag_mode == 1) {
    // uniform zz couplings whose partial sums are exact: sum of +-q == q*(n_par - n_anti)
    int anti = 0, k0 = 0;
    if (dm.n_zz_nn > 0) {
      uint64_t x = (s ^ (s >> 1)) & (((uint64_t)1 << (dm.L - 1)) - 1);
      anti = __popcll(x);
      k0 = dm.n_zz_nn;
    }
    for (int k = k0; k < dm.n_zz; ++k)
      anti += (int)(((s >> (dm.zz_i[k] - 1)) ^ (s >> (dm.zz_j[k] - 1))) & 1);
    return dm.diag_q * (double)(dm.n_zz - 2 * anti);
  }
  double d = 0.0;
  if (dm.diag_mode == 2) {   // the reference's loop, literally (SD_DIAG_LITERAL; what diag_mode 0 is checked against)
    for (int i = 1; i <= dm.L; ++i) d += dm.field[i - 1] * sz_of((s >> (i - 1)) & 1);
    for (int k = 0; k < dm.n_zz; ++k)
      d += (dm.zz_J[k] * sz_of((s >> (dm.zz_i[k] - 1)) & 1)) * sz_of((s >> (dm.zz_j[k] - 1)) & 1);
    return d;
  }
  // Same sum, same order, same bits: every term is +-(h_i/2) or +-(J_k/4) exactly, so it is selected, not multiplied.
  // All-zero fields add +-0 to 0 and are skipped.
  if (!dm.field_zero)
    for (int i = 0; i < dm.L; ++i) { const double h = dm.field_h[i]; d += ((s >> i) & 1) ? h : -h; }
  const uint64_t x = s ^ (s >> 1);                               // bit k: sites k+1, k+2 anti-parallel
  for (int k = 0; k < dm.n_zz_nn; ++k) { const double q = dm.zz_q[k]; d += ((x >> k) & 1) ? -q : q; }
  for (int k = dm.n_zz_nn; k < dm.n_zz; ++k) {
    const double q = dm.zz_q[k];
    d += (((s >> (dm.zz_i[k] - 1)) ^ (s >> (dm.zz_j[k] - 1))) & 1) ? -q : q;
  }
  return d;
}

// The list-order diagonal (diag_mode 0) split for a tile: the leading terms of the reference's sum depend on the prefix
// configuration P only -- the field terms of sites 1..p, or (all fields zero) the zz terms of the chain bonds inside the
// prefix -- so a thread forms them once (diag_head) and continues the same sequential sum per row (diag_tail).  Same
// order, same bits as diag_of.
struct DiagHead { double d0; int zz_from; };
__device__ __forceinline__ DiagHead diag_head(const sd_dev_model &dm, uint32_t P, int p) {
  DiagHead h{0.0, 0};
  if (!dm.field_zero) {
    for (int i = 0; i < p; ++i) { const double f = dm.field_h[i]; h.d0 += ((P >> i) & 1u) ? f : -f; }
  } else if (dm.n_zz_nn > 0 && p >= 2) {
    const uint32_t x = P ^ (P >> 1);
    for (int k = 0; k < p - 1; ++k) { const double q = dm.zz_q[k]; h.d0 += ((x >> k) & 1u) ? -q : q; }
    h.zz_from = p - 1;
  }
  return h;
}
__device__ __forceinline__ double diag_tail(const sd_dev_model &dm, const DiagHead &h, uint64_t s, int p) {
  double d = h.d0;
  if (!dm.field_zero)
    for (int i = p; i < dm.L; ++i) { const double f = dm.field_h[i]; d += ((s >> i) & 1) ? f : -f; }
  const uint64_t x = s ^ (s >> 1);
  for (int k = h.zz_from; k < dm.n_zz_nn; ++k) { const double q = dm.zz_q[k]; d += ((x >> k) & 1) ? -q : q; }
  for (int k = dm.n_zz_nn; k < dm.n_zz; ++k) {
    const double q = dm.zz_q[k];
    d += (((s >> (dm.zz_i[k] - 1)) ^ (s >> (dm.zz_j[k] - 1))) & 1) ? -q : q;
  }
  return d;
}

__device__ __forceinline__ int64_t binom_g(const sd_dev_model &dm, int n, int k) {
  return (k < 0 || k > n) ? 0 : dm.binom[n * (SD_MAX_L + 1) + k];
}

// combinadic unrank / rank in the reference order (generic path)
__device__ __forceinline__ uint64_t unrank_g(const sd_dev_model &dm, int64_t idx) {
  uint64_t s = 0;
  int r = dm.nup;
  for (int k = 1; k <= dm.L && r > 0; ++k) {
    int64_t c = binom_g(dm, dm.L - k, r - 1);
    if (idx < c) { s |= (uint64_t)1 << (k - 1); --r; }
    else idx -= c;
  }
  return s;
}
__device__ __forceinline__ int64_t rank_g(const sd_dev_model &dm, uint64_t s) {
  int64_t idx = 0;
  int r = dm.nup;
  for (int k = 1; k <= dm.L && r > 0; ++k) {
    if ((s >> (k - 1)) & 1) --r;
    else idx += binom_g(dm, dm.L - k, r - 1);
  }
  return idx;
}

// ---- epilogue: what is stored for row `row` given acc = (H psi)[row] ----
struct EpiSums { double s0, s1; };

template <int NC>
__device__ __forceinline__ void epilogue(int epi, const sd_epi_args &ea, int64_t row, typename VT<NC>::type acc,
                                         typename VT<NC>::type own, double *out, EpiSums &sums);

// Streams of the epilogues.  `out` is written once and not read again before the whole vector has gone by, and the side
// vectors of a fused recursion step (prev, phi, psi_t) are read / updated once per pass: non-temporal accesses keep them from
// displacing the psi rows that neighbouring tiles are about to re-read from L2 (measured on the plain apply at L=32:
// 11.43 -> 11.24 ms; `sc1` / `sc0 sc1` stores instead: 11.52 ms).  ea.stream_hint: bit 0 stores of out, bit 1 side streams.
typedef double sd_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_out(double *p, double v, int hint) {
  if (hint & 1) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ void st_out(double2 *p, double2 v, int hint) {
  if (hint & 1) { sd_d2v w; w.x = v.x; w.y = v.y; __builtin_nontemporal_store(w, reinterpret_cast<sd_d2v *>(p)); }
  else *p = v;
}
__device__ __forceinline__ double ld_side(const double *p, int hint) { return (hint & 2) ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ double2 ld_side(const double2 *p, int hint) {
  if (hint & 2) { const sd_d2v w = __builtin_nontemporal_load(reinterpret_cast<const sd_d2v *>(p)); return make_double2(w.x, w.y); }
  return *p;
}
__device__ __forceinline__ void st_side(double *p, double v, int hint) {
  if (hint & 2) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ void st_side(double2 *p, double2 v, int hint) {
  if (hint & 2) { sd_d2v w; w.x = v.x; w.y = v.y; __builtin_nontemporal_store(w, reinterpret_cast<sd_d2v *>(p)); }
  else *p = v;
}

template <>
__device__ __forceinline__ void epilogue<1>(int epi, const sd_epi_args &ea, int64_t row, double acc, double own,
                                            double *out, EpiSums &sums) {
  const int hint = ea.stream_hint;
  switch (epi) {
    case SD_EPI_PLAIN:
      st_out(out + row, ea.negate ? -acc : acc, hint);
      break;
    case SD_EPI_DOT: {
      double o = ea.negate ? -acc : acc;
      st_out(out + row, o, hint);
      sums.s0 += own * o;
    } break;
    case SD_EPI_RESCALE:
      st_out(out + row, (acc - ea.b * own) / ea.a, hint);
      break;
    case SD_EPI_RESCALE_DOT: {
      double o = (acc - ea.b * own) / ea.a;
      st_out(out + row, o, hint);
      double ph = ea.phi ? ld_side((const double *)ea.phi + row, hint) : own;   // phi == null: <v_curr|v_next> (moment doubling)
      sums.s0 += ph * o;
      sums.s1 += o * o;
    } break;
    case SD_EPI_KPM: {
      double o = 2.0 * ((acc - ea.b * own) / ea.a) - ld_side((const double *)ea.prev + row, hint);
      st_out(out + row, o, hint);
      double ph = ea.phi ? ld_side((const double *)ea.phi + row, hint) : own;   // phi == null: <v_curr|v_next> (moment doubling)
      sums.s0 += ph * o;
      sums.s1 += o * o;
    } break;
    case SD_EPI_RECUR:
      st_out(out + row, 2.0 * ((acc - ea.b * own) / ea.a) - ld_side((const double *)ea.prev + row, hint), hint);
      break;
    case SD_EPI_CHEB2: {
      double o = 2.0 * ((acc - ea.b * own) / ea.a) - ld_side((const double *)ea.prev + row, hint);
      st_out(out + row, o, hint);
      double *pt = (double *)ea.accv;
      double t = ld_side(pt + row, hint);
      t += ea.c0_re * own;
      t += ea.c_re * o;
      st_side(pt + row, t, hint);
    } break;
    default: {  // SD_EPI_CHEB on real vectors: real accumulate with real coefficient
      double o = 2.0 * ((acc - ea.b * own) / ea.a) - ld_side((const double *)ea.prev + row, hint);
      st_out(out + row, o, hint);
      double *pt = (double *)ea.accv;
      double t = ld_side(pt + row, hint);
      t += ea.c_re * o;
      st_side(pt + row, t, hint);
    } break;
  }
}

template <>
__device__ __forceinline__ void epilogue<2>(int epi, const sd_epi_args &ea, int64_t row, double2 acc, double2 own,
                                            double *out, EpiSums &sums) {
  double2 *o2 = (double2 *)out;
  const int hint = ea.stream_hint;
  switch (epi) {
    case SD_EPI_PLAIN:
      st_out(o2 + row, ea.negate ? make_double2(-acc.x, -acc.y) : acc, hint);
      break;
    case SD_EPI_DOT: {
      double2 o = ea.negate ? make_double2(-acc.x, -acc.y) : acc;
      st_out(o2 + row, o, hint);
      sums.s0 += own.x * o.x + own.y * o.y;   // conj(own) * o
      sums.s1 += own.x * o.y - own.y * o.x;
    } break;
    case SD_EPI_RESCALE:
      st_out(o2 + row, make_double2((acc.x - ea.b * own.x) / ea.a, (acc.y - ea.b * own.y) / ea.a), hint);
      break;
    case SD_EPI_RESCALE_DOT: {
      double2 o = make_double2((acc.x - ea.b * own.x) / ea.a, (acc.y - ea.b * own.y) / ea.a);
      st_out(o2 + row, o, hint);
      double2 ph = ea.phi ? ld_side((const double2 *)ea.phi + row, hint) : own;   // phi == null: <v_curr|v_next> (moment doubling)
      sums.s0 += ph.x * o.x + ph.y * o.y;     // Re <phi|o>
      sums.s1 += o.x * o.x + o.y * o.y;
    } break;
    case SD_EPI_KPM: {
      double2 pv = ld_side((const double2 *)ea.prev + row, hint);
      double2 o = make_double2(2.0 * ((acc.x - ea.b * own.x) / ea.a) - pv.x,
                               2.0 * ((acc.y - ea.b * own.y) / ea.a) - pv.y);
      st_out(o2 + row, o, hint);
      double2 ph = ea.phi ? ld_side((const double2 *)ea.phi + row, hint) : own;
      sums.s0 += ph.x * o.x + ph.y * o.y;
      sums.s1 += o.x * o.x + o.y * o.y;
    } break;
    case SD_EPI_RECUR: {
      double2 pv = ld_side((const double2 *)ea.prev + row, hint);
      st_out(o2 + row, make_double2(2.0 * ((acc.x - ea.b * own.x) / ea.a) - pv.x, 2.0 * ((acc.y - ea.b * own.y) / ea.a) - pv.y), hint);
    } break;
    case SD_EPI_CHEB2: {  // two Chebyshev terms per pass over psi_t: the deferred c0*phi_k (phi_k = this apply's input) and
      // c*phi_{k+1}, added in that order with the arithmetic of two SD_EPI_CHEB passes (bit-identical), one psi_t read-modify-write
      double2 pv = ld_side((const double2 *)ea.prev + row, hint);
      double2 o = make_double2(2.0 * ((acc.x - ea.b * own.x) / ea.a) - pv.x,
                               2.0 * ((acc.y - ea.b * own.y) / ea.a) - pv.y);
      st_out(o2 + row, o, hint);
      double2 *pt = (double2 *)ea.accv;
      double2 t = ld_side(pt + row, hint);
      t.x += ea.c0_re * own.x - ea.c0_im * own.y;
      t.y += ea.c0_re * own.y + ea.c0_im * own.x;
      t.x += ea.c_re * o.x - ea.c_im * o.y;
      t.y += ea.c_re * o.y + ea.c_im * o.x;
      st_side(pt + row, t, hint);
    } break;
    default: {  // SD_EPI_CHEB  (src/TimeEvolution/Chebyshev.jl:112-117)
      double2 pv = ld_side((const double2 *)ea.prev + row, hint);
      double2 o = make_double2(2.0 * ((acc.x - ea.b * own.x) / ea.a) - pv.x,
                               2.0 * ((acc.y - ea.b * own.y) / ea.a) - pv.y);
      st_out(o2 + row, o, hint);
      double2 *pt = (double2 *)ea.accv;
      double2 t = ld_side(pt + row, hint);
      t.x += ea.c_re * o.x - ea.c_im * o.y;
      t.y += ea.c_re * o.y + ea.c_im * o.x;
      st_side(pt + row, t, hint);
    } break;
  }
}

// deterministic block reduction of two doubles; result valid in thread 0
__device__ __forceinline__ void block_reduce2(double &a, double &b, double *red /* >= 2*16 doubles LDS */) {
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) { red[2 * wv] = a; red[2 * wv + 1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double x = 0.0, y = 0.0;
    for (int w = 0; w < nw; ++w) { x += red[2 * w]; y += red[2 * w + 1]; }
    a = x; b = y;
  }
}

__device__ __forceinline__ bool epi_has_sums(int epi) {
  return epi == SD_EPI_DOT || epi == SD_EPI_KPM || epi == SD_EPI_RESCALE_DOT;
}


template <bool FMA>
__device__ __forceinline__ double acc1(double acc, double J, double v) {
  return FMA ? __builtin_fma(J, v, acc) : acc + J * v;
}
template <bool FMA>
__device__ __forceinline__ double accum(double acc, double J, double v) { return acc1<FMA>(acc, J, v); }
template <bool FMA>
__device__ __forceinline__ double2 accum(double2 acc, double J, double2 v) {
  return make_double2(acc1<FMA>(acc.x, J, v.x), acc1<FMA>(acc.y, J, v.y));
}

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int64_t rl64(int64_t v, int lane) {
  const uint32_t lo = (uint32_t)rl((int)(uint32_t)v, lane), hi = (uint32_t)rl((int)(uint32_t)((uint64_t)v >> 32), lane);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double rld(double v, int lane) { return __longlong_as_double(rl64(__double_as_longlong(v), lane)); }

struct FarBond { int64_t base; double J; int lo, n; };   // partner rows: psi[base + (i - lo)] for lo <= i < lo + n

// 128-bit buffer descriptor over [p, p + bytes): loads with a byte offset >= bytes return 0 (hardware range check),
// which replaces every per-row "is this row in range" test of the far-bond streams.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void buf_load(double &v, __amdgpu_buffer_rsrc_t r, uint32_t off) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const u2 raw = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
  v = __hiloint2double((int)raw.y, (int)raw.x);
}
__device__ __forceinline__ uint32_t buf_load_u16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0);
}
__device__ __forceinline__ void buf_load(double2 &v, __amdgpu_buffer_rsrc_t r, uint32_t off) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const u4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  v.x = __hiloint2double((int)raw.y, (int)raw.x);
  v.y = __hiloint2double((int)raw.w, (int)raw.z);
}


// counter-based N(0,1): element k of stream `seed` = Box-Muller on two uniforms hashed from (seed, k)
__host__ __device__ inline uint64_t sd_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ inline double sd_randn_at(uint64_t seed, uint64_t k) {
  const uint64_t h1 = sd_mix64(seed ^ sd_mix64(2 * k)), h2 = sd_mix64(seed ^ sd_mix64(2 * k + 1));
  const double u1 = ((double)(h1 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)(h2 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

}  // namespace sd_dev
