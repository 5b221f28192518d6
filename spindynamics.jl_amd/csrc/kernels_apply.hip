// H|psi> apply kernels for gfx950 (MI355X).  Replaces the reference's
// Threads.@threads row loop of apply_H! (src/Hamiltonian.jl:211-273), the
// separate rescale pass of apply_rescaled_H! (:286-301, fused here into the
// store), and Sz_q_vector (:307-337).
//
// Two device paths:
//  * k_apply_tiled   -- fixed-nup sector.  One workgroup per TILE (all rows
//    sharing a prefix configuration of sites 1..p, see sd_internal.hpp).  The
//    tile's psi is staged once in LDS; hops on bonds inside the suffix are LDS
//    reads at idx +- C(LS-a-1,u) (no search, no hash); hops on prefix bonds
//    are coalesced streams from another whole tile at the same in-tile offset;
//    the straddling bond is a coalesced stream from half a tile.
//  * k_apply_generic -- any model (full 2^L basis, L up to 63, arbitrary
//    bonds): one row per thread, combinadic unrank / rank per hop.
//
// Per-row operation order follows the reference exactly (fields, zz in list
// order, value = diag*psi[idx], then hops in list order, value += J*psi[idx'])
// and the file is compiled with -ffp-contract=off, so a plain apply is
// bit-identical to the CPU oracle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "sd_internal.hpp"

#define SD_BIN_STRIDE 17

namespace {

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
  for (int i = 1; i <= dm.L; ++i) d += dm.field[i - 1] * sz_of((s >> (i - 1)) & 1);
  for (int k = 0; k < dm.n_zz; ++k)
    d += (dm.zz_J[k] * sz_of((s >> (dm.zz_i[k] - 1)) & 1)) * sz_of((s >> (dm.zz_j[k] - 1)) & 1);
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

template <>
__device__ __forceinline__ void epilogue<1>(int epi, const sd_epi_args &ea, int64_t row, double acc, double own,
                                            double *out, EpiSums &sums) {
  switch (epi) {
    case SD_EPI_PLAIN:
      out[row] = ea.negate ? -acc : acc;
      break;
    case SD_EPI_DOT: {
      double o = ea.negate ? -acc : acc;
      out[row] = o;
      sums.s0 += own * o;
    } break;
    case SD_EPI_RESCALE:
      out[row] = (acc - ea.b * own) / ea.a;
      break;
    case SD_EPI_RESCALE_DOT: {
      double o = (acc - ea.b * own) / ea.a;
      out[row] = o;
      double ph = ((const double *)ea.phi)[row];
      sums.s0 += ph * o;
      sums.s1 += o * o;
    } break;
    case SD_EPI_KPM: {
      double o = 2.0 * ((acc - ea.b * own) / ea.a) - ((const double *)ea.prev)[row];
      out[row] = o;
      double ph = ((const double *)ea.phi)[row];
      sums.s0 += ph * o;
      sums.s1 += o * o;
    } break;
    default: {  // SD_EPI_CHEB on real vectors: real accumulate with real coefficient
      double o = 2.0 * ((acc - ea.b * own) / ea.a) - ((const double *)ea.prev)[row];
      out[row] = o;
      double *pt = (double *)ea.accv;
      pt[row] += ea.c_re * o;
    } break;
  }
}

template <>
__device__ __forceinline__ void epilogue<2>(int epi, const sd_epi_args &ea, int64_t row, double2 acc, double2 own,
                                            double *out, EpiSums &sums) {
  double2 *o2 = (double2 *)out;
  switch (epi) {
    case SD_EPI_PLAIN:
      o2[row] = ea.negate ? make_double2(-acc.x, -acc.y) : acc;
      break;
    case SD_EPI_DOT: {
      double2 o = ea.negate ? make_double2(-acc.x, -acc.y) : acc;
      o2[row] = o;
      sums.s0 += own.x * o.x + own.y * o.y;   // conj(own) * o
      sums.s1 += own.x * o.y - own.y * o.x;
    } break;
    case SD_EPI_RESCALE:
      o2[row] = make_double2((acc.x - ea.b * own.x) / ea.a, (acc.y - ea.b * own.y) / ea.a);
      break;
    case SD_EPI_RESCALE_DOT: {
      double2 o = make_double2((acc.x - ea.b * own.x) / ea.a, (acc.y - ea.b * own.y) / ea.a);
      o2[row] = o;
      double2 ph = ((const double2 *)ea.phi)[row];
      sums.s0 += ph.x * o.x + ph.y * o.y;     // Re <phi|o>
      sums.s1 += o.x * o.x + o.y * o.y;
    } break;
    case SD_EPI_KPM: {
      double2 pv = ((const double2 *)ea.prev)[row];
      double2 o = make_double2(2.0 * ((acc.x - ea.b * own.x) / ea.a) - pv.x,
                               2.0 * ((acc.y - ea.b * own.y) / ea.a) - pv.y);
      o2[row] = o;
      double2 ph = ((const double2 *)ea.phi)[row];
      sums.s0 += ph.x * o.x + ph.y * o.y;
      sums.s1 += o.x * o.x + o.y * o.y;
    } break;
    default: {  // SD_EPI_CHEB  (src/TimeEvolution/Chebyshev.jl:112-117)
      double2 pv = ((const double2 *)ea.prev)[row];
      double2 o = make_double2(2.0 * ((acc.x - ea.b * own.x) / ea.a) - pv.x,
                               2.0 * ((acc.y - ea.b * own.y) / ea.a) - pv.y);
      o2[row] = o;
      double2 *pt = (double2 *)ea.accv;
      double2 t = pt[row];
      t.x += ea.c_re * o.x - ea.c_im * o.y;
      t.y += ea.c_re * o.y + ea.c_im * o.x;
      pt[row] = t;
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

// =====================================================================
// tiled kernel
// =====================================================================
//
// One workgroup per tile (prefix configuration P).  Each thread owns R rows
// i = tid + r*BLOCK of the tile.
//   1. own rows + suffix configurations are requested from HBM;
//   2. every wave builds, in its own registers (lane b-1 <-> prefix bond b, lane
//      p-1 <-> the straddling bond), the list of flippable far bonds with the
//      partner tile's base offset -- no LDS, no barrier, one memory latency;
//   3. the far-bond partner rows are streamed with a two-deep ping-pong
//      pipeline (loads of bond k+1 in flight while bond k is accumulated);
//   4. own rows go to LDS, barrier, then the suffix bonds are LDS reads at
//      idx +- C(LS-a-1,u);
//   5. fused epilogue + store.
// Accumulation order per row is the reference's bond order 1..L-1.  When every
// NN hop amplitude is a power of two (XXZChain default 0.5) J*psi is exact and
// acc + J*psi is evaluated with one fma (bit-identical to the unfused form).
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
__device__ __forceinline__ void buf_load(double2 &v, __amdgpu_buffer_rsrc_t r, uint32_t off) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const u4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  v.x = __hiloint2double((int)raw.y, (int)raw.x);
  v.y = __hiloint2double((int)raw.w, (int)raw.z);
}

template <int NC, int R, int BLOCK, bool FMA, bool DIAG = false>
__global__ __launch_bounds__(BLOCK, 4) void k_apply_tiled(sd_dev_model dm, double *__restrict__ out_,
                                                       const double *__restrict__ psi_, int epi, sd_epi_args ea,
                                                       double *__restrict__ partials, int max_len) {
  using V = typename VT<NC>::type;
  constexpr uint32_t ES = sizeof(V);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V *tile = reinterpret_cast<V *>(smem);                       // max_len rows + one all-zero row at index max_len
  int *lbin = reinterpret_cast<int *>(smem + (size_t)(max_len + 1) * sizeof(V));
  double *red = reinterpret_cast<double *>(lbin + 16 * SD_BIN_STRIDE);

  const V *__restrict__ psi = reinterpret_cast<const V *>(psi_);
  const V *__restrict__ halo = reinterpret_cast<const V *>(ea.halo);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int tix = blockIdx.x;
  const sd_tile_rec rec = dm.single_rec[tix];       // one 32-byte scalar load: no dependent table look-ups at start-up
  const uint32_t P = rec.prefix;
  const int64_t base = rec.base;
  const int p = dm.p, LS = dm.LS;
  const int len = rec.len;
  const int nU = rec.nU;                            // rows whose first suffix site is up
  const uint16_t *__restrict__ sufS = dm.suf_states + rec.suf_off;
  const int nn = dm.nn_hops;
  unsigned long long *stamp = (DIAG && dm.stamps) ? dm.stamps + 8 * (size_t)tix : nullptr;
#define SD_STAMP(k)                                                                   \
  do {                                                                                \
    if (DIAG && stamp) {                                                              \
      __builtin_amdgcn_sched_barrier(0);                                              \
      unsigned long long t__;                                                         \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); \
      __builtin_amdgcn_sched_barrier(0);                                              \
      if (tid == 0) stamp[k] = t__;                                                   \
    }                                                                                 \
  } while (0)
  SD_STAMP(0);

  // ---- 1. request own rows (rows >= len read 0 through the range check) and suffix configurations ----
  V own[R];
  uint32_t sig[R];
  uint32_t ioff[R];   // byte offset of the row inside a tile-sized stream
  int irow[R];        // row index clamped into the tile (LDS addressing of idle rows)
  {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(psi + base, (uint32_t)len * ES);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = tid + r * BLOCK;
      ioff[r] = (uint32_t)i * ES;
      irow[r] = i < len ? i : len - 1;
      buf_load(own[r], rs, ioff[r]);
      sig[r] = sufS[irow[r]];
    }
  }

  // ---- 2. per-wave list of flippable far bonds (lane b-1 <-> bond b <= p-1, lane p-1 <-> straddle) ----
  uint64_t fmask = 0;
  int64_t my_base = 0;
  double my_J = 0.0;
  int my_lo = 0, my_n = len;
  if (nn > 0 && p >= 1) {
    bool fl = false;
    const int b = lane + 1;
    if (b <= p - 1) {
      fl = (((P >> (b - 1)) ^ (P >> b)) & 1u) && !(dm.dbg & 1);
      if (fl) { my_base = dm.addr[P ^ (3u << (b - 1))]; my_J = dm.hop_J[b - 1]; }
    } else if (b == p) {
      // bit p of P up:   our rows with first suffix site down (i >= nU) <-> partner rows i - nU
      // bit p of P down: our rows with first suffix site up   (i <  nU) <-> partner rows nUq + i
      const uint32_t bitp = (P >> (p - 1)) & 1u;
      const uint32_t Q = P ^ (1u << (p - 1));
      const int t2q = dm.nup - __popc(Q);
      if (t2q >= 0 && t2q <= LS && !(dm.dbg & 2)) {
        const int nUq = (int)binom_g(dm, LS - 1, t2q - 1);
        int64_t shift;
        if (bitp) { my_lo = nU; my_n = len - nU; shift = 0; }
        else { my_lo = 0; my_n = nU; shift = nUq; }
        fl = my_n > 0;
        if (fl) { my_base = dm.addr[Q] + shift; my_J = dm.hop_J[p - 1]; }
      }
    }
    fmask = __ballot(fl);
  }

  auto get_bond = [&](int ln) {
    FarBond fb;
    fb.base = rl64(my_base, ln); fb.J = rld(my_J, ln);
    fb.lo = rl(my_lo, ln); fb.n = rl(my_n, ln);
    return fb;
  };
  // rows outside [lo, lo+n) wrap to a huge unsigned offset or exceed n*ES: the load returns 0 and J*0 leaves acc unchanged
  auto issue = [&](const FarBond &fb, V(&v)[R]) {
    // partner tile lives in the owned rows, or (sharded plans) in the halo imported from its owner
    const V *__restrict__ pb = (halo && fb.base >= dm.n_local) ? halo + (fb.base - dm.n_local) : psi + fb.base;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(pb, (uint32_t)fb.n * ES);
    const uint32_t lo_b = (uint32_t)fb.lo * ES;
#pragma unroll
    for (int r = 0; r < R; ++r) buf_load(v[r], rs, ioff[r] - lo_b);
  };

  SD_STAMP(1);
  // first far bond in flight before the own rows have even arrived
  V va[R], vb[R];
  FarBond fa{}, fbb{};
  uint64_t mk = fmask;
  bool have_a = false;
  auto next_lane = [&](uint64_t &m_) { const int ln = __builtin_ctzll(m_); m_ &= m_ - 1; return ln; };   // ascending bond order
  if (mk) { fa = get_bond(next_lane(mk)); issue(fa, va); have_a = true; }

  // ---- diagonal (needs own) ----
  V acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t s = (uint64_t)P | ((uint64_t)sig[r] << p);
    acc[r] = vscale(diag_of(dm, s), own[r]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (tid + r * BLOCK < len) tile[irow[r]] = own[r];
  if (tid == 0) tile[max_len] = V{};   // the zero row read by lanes whose suffix bond is not flippable
  for (int k = tid; k < 16 * SD_BIN_STRIDE; k += BLOCK) {
    int n = k / SD_BIN_STRIDE, kk = k - n * SD_BIN_STRIDE;
    lbin[k] = (int)binom_g(dm, n, kk);
  }

  SD_STAMP(2);
  // ---- 3. far bonds: ping-pong pipeline, accumulation in bond order ----
  while (have_a) {
    bool have_b = false;
    if (mk) { fbb = get_bond(next_lane(mk)); issue(fbb, vb); have_b = true; }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = accum<FMA>(acc[r], fa.J, va[r]);
    have_a = false;
    if (!have_b) break;
    if (mk) { fa = get_bond(next_lane(mk)); issue(fa, va); have_a = true; }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = accum<FMA>(acc[r], fbb.J, vb[r]);
  }
  SD_STAMP(3);
  __syncthreads();
  SD_STAMP(4);

  // ---- 4. bonds inside the suffix: LDS reads at idx +- C(LS-a-1, u); branch-free so the R rows' reads overlap ----
  if (nn > 0 && !(dm.dbg & 4)) {
    uint32_t dw[R];   // bit a-1 set <=> suffix bond a is flippable
#pragma unroll
    for (int r = 0; r < R; ++r) dw[r] = sig[r] ^ (sig[r] >> 1);
    for (int a = 1; a <= LS - 1; ++a) {
      const double J = dm.hop_J[p + a - 1];
      const int *brow = lbin + (LS - a - 1) * SD_BIN_STRIDE;
      int d[R];
#pragma unroll
      for (int r = 0; r < R; ++r) d[r] = brow[__popc(sig[r] >> (a + 1))];
      V v[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool up = (sig[r] >> (a - 1)) & 1u;
        const bool fl = (dw[r] >> (a - 1)) & 1u;
        const int ip = up ? irow[r] + d[r] : irow[r] - d[r];
        v[r] = tile[fl ? ip : max_len];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = accum<FMA>(acc[r], J, v[r]);
    }
  }
  // ---- remaining (general) bonds: rank through the tile tables ----
  if (nn < dm.n_hop) {
    const uint64_t pmask = ((uint64_t)1 << p) - 1;
    for (int h = nn; h < dm.n_hop; ++h) {
      const int bi = dm.hop_i[h] - 1, bj = dm.hop_j[h] - 1;
      const double J = dm.hop_J[h];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint64_t s = (uint64_t)P | ((uint64_t)sig[r] << p);
        if (((s >> bi) ^ (s >> bj)) & 1) {
          const uint64_t s2 = s ^ ((uint64_t)1 << bi) ^ ((uint64_t)1 << bj);
          const int64_t idx = dm.addr[(uint32_t)(s2 & pmask)] + dm.suf_rank[(uint32_t)(s2 >> p)];
          acc[r] = accum<false>(acc[r], J, (halo && idx >= dm.n_local) ? halo[idx - dm.n_local] : psi[idx]);
        }
      }
    }
  }

  SD_STAMP(5);
  // ---- 5. epilogue + store ----
  EpiSums sums{0.0, 0.0};
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = tid + r * BLOCK;
    if (i < len) epilogue<NC>(epi, ea, base + i, acc[r], tile[i], out_, sums);
  }
  SD_STAMP(6);
  if (DIAG && stamp && tid == 0) stamp[7] = __builtin_amdgcn_s_memrealtime();
  if (epi_has_sums(epi)) {
    double a = sums.s0, b = sums.s1;
    block_reduce2(a, b, red);
    if (tid == 0) { partials[2 * (size_t)tix] = a; partials[2 * (size_t)tix + 1] = b; }
  }
}

// =====================================================================
// grouped kernel: 8 tiles related by three disjoint flippable top bonds per workgroup
// =====================================================================
//
// The three generator bonds g1<g2<g3 (odd prefix bonds, chosen on the host) map the 8 member tiles onto each other with
// identical in-tile row order.  All 8 tiles are staged in ONE LDS image, so those three far bonds -- top bonds whose
// partner tiles never survive in L2 -- are served from LDS: 3 of the ~10 far streams per row disappear for the LDS a
// three-sites-larger suffix tile would need to save 1.5.  Four teams of 256 threads each process two members; after the
// single staging barrier the teams run independently (far-bond streams, suffix bonds, store), which overlaps one
// team's memory phase with another's LDS/VALU phase.  Per-row accumulation order is unchanged (bit-identical results).
template <int NC, bool FMA, int NGEN>
__global__ __launch_bounds__(128 << NGEN, 4) void k_apply_grouped(sd_dev_model dm, double *__restrict__ out_,
                                                        const double *__restrict__ psi_, int epi, sd_epi_args ea,
                                                        double *__restrict__ partials, int max_len) {
  using V = typename VT<NC>::type;
  constexpr int R = 4, TEAM = 256, NMEM = 1 << NGEN, NTEAM = NMEM / 2, NTHR = NTEAM * TEAM;
  constexpr uint32_t ES = sizeof(V);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V *tiles = reinterpret_cast<V *>(smem);                       // NMEM * max_len rows + one all-zero row
  const int zero_row = NMEM * max_len;
  int *lbin = reinterpret_cast<int *>(smem + (size_t)(zero_row + 1) * sizeof(V));
  double *red = reinterpret_cast<double *>(lbin + 16 * SD_BIN_STRIDE);

  const V *__restrict__ psi = reinterpret_cast<const V *>(psi_);
  const int tid = threadIdx.x;
  const int team = tid / TEAM, ttid = tid - team * TEAM, lane = tid & 63;
  const int gix = blockIdx.x;
  const uint32_t P0 = dm.group_P0[gix];
  const uint32_t gens = dm.group_gens[gix];
  const int gb0 = gens & 255, gb1 = (gens >> 8) & 255, gb2 = NGEN > 2 ? (gens >> 16) & 255 : 0;
  const int p = dm.p, LS = dm.LS;
  const int t2 = dm.nup - __popc(P0);            // the same for every member
  const int len = (int)binom_g(dm, LS, t2);
  const int nU = (int)binom_g(dm, LS - 1, t2 - 1);
  const uint16_t *__restrict__ sufS = dm.suf_states + dm.suf_off[t2];

  auto member_prefix = [&](int g) {
    uint32_t P = P0;
    if (g & 1) P ^= 3u << (gb0 - 1);
    if (g & 2) P ^= 3u << (gb1 - 1);
    if (NGEN > 2 && (g & 4)) P ^= 3u << (gb2 - 1);
    return P;
  };

  uint32_t sig[R], ioff[R];
  int irow[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = ttid + r * TEAM;
    ioff[r] = (uint32_t)i * ES;
    irow[r] = i < len ? i : len - 1;
    sig[r] = sufS[irow[r]];
  }

  // ---- stage the member tiles (each team loads members team and team+NTEAM), one member at a time ----
#pragma unroll 1
  for (int mm = 0; mm < 2; ++mm) {
    const int g = team + NTEAM * mm;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(psi + dm.addr[member_prefix(g)], (uint32_t)len * ES);
    V own[R];
#pragma unroll
    for (int r = 0; r < R; ++r) buf_load(own[r], rs, ioff[r]);
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (ttid + r * TEAM < len) tiles[g * max_len + irow[r]] = own[r];
  }
  if (tid == 0) tiles[zero_row] = V{};
  for (int k = tid; k < 16 * SD_BIN_STRIDE; k += NTHR) {
    int n = k / SD_BIN_STRIDE, kk = k - n * SD_BIN_STRIDE;
    lbin[k] = (int)binom_g(dm, n, kk);
  }
  __syncthreads();

  EpiSums sums{0.0, 0.0};
#pragma unroll 1
  for (int mm = 0; mm < 2; ++mm) {
    const int g = team + NTEAM * mm;
    const uint32_t P = member_prefix(g);
    const int64_t base = dm.addr[P];
    const V *__restrict__ tile = tiles + g * max_len;

    // far-bond list of this member: lane b-1 <-> prefix bond b, lane p-1 <-> straddle; generator bonds carry the
    // LDS row offset of the partner member instead of a global base
    uint64_t fmask = 0;
    int64_t my_base = 0;
    double my_J = 0.0;
    int my_lo = 0, my_n = len, my_lds = -1;
    {
      bool fl = false;
      const int b = lane + 1;
      if (b <= p - 1) {
        fl = ((P >> (b - 1)) ^ (P >> b)) & 1u;
        if (fl) {
          my_J = dm.hop_J[b - 1];
          if (b == gb0) my_lds = (g ^ 1) * max_len;
          else if (b == gb1) my_lds = (g ^ 2) * max_len;
          else if (NGEN > 2 && b == gb2) my_lds = (g ^ 4) * max_len;
          else my_base = dm.addr[P ^ (3u << (b - 1))];
        }
      } else if (b == p) {
        const uint32_t bitp = (P >> (p - 1)) & 1u;
        const uint32_t Q = P ^ (1u << (p - 1));
        const int t2q = dm.nup - __popc(Q);
        if (t2q >= 0 && t2q <= LS) {
          const int nUq = (int)binom_g(dm, LS - 1, t2q - 1);
          int64_t shift;
          if (bitp) { my_lo = nU; my_n = len - nU; shift = 0; }
          else { my_lo = 0; my_n = nU; shift = nUq; }
          fl = my_n > 0;
          if (fl) { my_base = dm.addr[Q] + shift; my_J = dm.hop_J[p - 1]; }
        }
      }
      fmask = __ballot(fl);
    }
    struct GB { int64_t base; double J; int lo, n, lds; };
    auto get_bond = [&](int ln) {
      GB fb;
      fb.base = rl64(my_base, ln); fb.J = rld(my_J, ln);
      fb.lo = rl(my_lo, ln); fb.n = rl(my_n, ln); fb.lds = rl(my_lds, ln);
      return fb;
    };
    auto issue = [&](const GB &fb, V(&v)[R]) {
      if (fb.lds >= 0) {                                  // generator bond: partner member is in LDS
        const V *__restrict__ pt = tiles + fb.lds;
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = pt[irow[r]];
      } else {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(psi + fb.base, (uint32_t)fb.n * ES);
        const uint32_t lo_b = (uint32_t)fb.lo * ES;
#pragma unroll
        for (int r = 0; r < R; ++r) buf_load(v[r], rs, ioff[r] - lo_b);
      }
    };
    auto next_lane = [&](uint64_t &m_) { const int ln = __builtin_ctzll(m_); m_ &= m_ - 1; return ln; };

    V va[R], vb[R];
    GB fa{}, fbb{};
    uint64_t mk = fmask;
    bool have_a = false;
    if (mk) { fa = get_bond(next_lane(mk)); issue(fa, va); have_a = true; }

    V acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t s = (uint64_t)P | ((uint64_t)sig[r] << p);
      acc[r] = vscale(diag_of(dm, s), tile[irow[r]]);
    }
    while (have_a) {
      bool have_b = false;
      if (mk) { fbb = get_bond(next_lane(mk)); issue(fbb, vb); have_b = true; }
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = accum<FMA>(acc[r], fa.J, va[r]);
      have_a = false;
      if (!have_b) break;
      if (mk) { fa = get_bond(next_lane(mk)); issue(fa, va); have_a = true; }
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = accum<FMA>(acc[r], fbb.J, vb[r]);
    }

    // suffix bonds from this member's LDS tile
    {
      uint32_t dw[R];
#pragma unroll
      for (int r = 0; r < R; ++r) dw[r] = sig[r] ^ (sig[r] >> 1);
      for (int a = 1; a <= LS - 1; ++a) {
        const double J = dm.hop_J[p + a - 1];
        const int *brow = lbin + (LS - a - 1) * SD_BIN_STRIDE;
        int d[R];
#pragma unroll
        for (int r = 0; r < R; ++r) d[r] = brow[__popc(sig[r] >> (a + 1))];
        V v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const bool up = (sig[r] >> (a - 1)) & 1u;
          const bool fl = (dw[r] >> (a - 1)) & 1u;
          const int ip = up ? irow[r] + d[r] : irow[r] - d[r];
          v[r] = fl ? tile[ip] : tiles[zero_row];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = accum<FMA>(acc[r], J, v[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = ttid + r * TEAM;
      if (i < len) epilogue<NC>(epi, ea, base + i, acc[r], tile[i], out_, sums);
    }
  }
  if (epi_has_sums(epi)) {
    double a = sums.s0, b = sums.s1;
    block_reduce2(a, b, red);
    if (tid == 0) { partials[2 * (size_t)gix] = a; partials[2 * (size_t)gix + 1] = b; }
  }
}

// =====================================================================
// generic kernel: one row per thread, grid-stride
// =====================================================================
template <int NC>
__global__ __launch_bounds__(256) void k_apply_generic(sd_dev_model dm, double *__restrict__ out_,
                                                       const double *__restrict__ psi_, int epi, sd_epi_args ea,
                                                       double *__restrict__ partials) {
  using V = typename VT<NC>::type;
  __shared__ double red[32];
  const V *__restrict__ psi = reinterpret_cast<const V *>(psi_);
  const bool full = dm.nup < 0;
  EpiSums sums{0.0, 0.0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < dm.N; idx += stride) {
    const uint64_t s = full ? (uint64_t)idx : unrank_g(dm, idx);
    const V own = psi[idx];
    V acc = vscale(diag_of(dm, s), own);
    for (int h = 0; h < dm.n_hop; ++h) {
      const int bi = dm.hop_i[h] - 1, bj = dm.hop_j[h] - 1;
      if (((s >> bi) ^ (s >> bj)) & 1) {
        const uint64_t s2 = s ^ ((uint64_t)1 << bi) ^ ((uint64_t)1 << bj);
        const int64_t nidx = full ? (int64_t)s2 : rank_g(dm, s2);
        acc = vadd_mul(acc, dm.hop_J[h], psi[nidx]);
      }
    }
    epilogue<NC>(epi, ea, idx, acc, own, out_, sums);
  }
  if (epi_has_sums(epi)) {
    double a = sums.s0, b = sums.s1;
    block_reduce2(a, b, red);
    if (threadIdx.x == 0) { partials[2 * (size_t)blockIdx.x] = a; partials[2 * (size_t)blockIdx.x + 1] = b; }
  }
}

// fixed-order reduction of the per-block partial pairs -> scalars[0..1]
__global__ __launch_bounds__(1024) void k_reduce_pairs(const double *__restrict__ partials, int64_t n,
                                                       double *__restrict__ scalars) {
  __shared__ double red[32];
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) { a += partials[2 * i]; b += partials[2 * i + 1]; }
  block_reduce2(a, b, red);
  if (threadIdx.x == 0) { scalars[0] = a; scalars[1] = b; }
}

// =====================================================================
// Sz_q_vector  (src/Hamiltonian.jl:307-337)
// =====================================================================
struct SzqPhases { double re[SD_MAX_L + 1], im[SD_MAX_L + 1]; };

template <int NCIN>
__device__ __forceinline__ void szq_row(const sd_dev_model &dm, const SzqPhases &ph, double normfact, uint64_t s,
                                        const double *__restrict__ psi0, int64_t row, double2 *__restrict__ phi) {
  double sr = 0.0, si = 0.0;
  for (int r = 0; r < dm.L; ++r) {
    const double z = sz_of((s >> r) & 1);
    sr += ph.re[r] * z;
    si += ph.im[r] * z;
  }
  const double ar = normfact * sr, ai = normfact * si;
  double xr, xi;
  if (NCIN == 2) { xr = psi0[2 * row]; xi = psi0[2 * row + 1]; }
  else { xr = psi0[row]; xi = 0.0; }
  phi[row] = make_double2(ar * xr - ai * xi, ar * xi + ai * xr);
}

template <int NCIN>
__global__ __launch_bounds__(256) void k_szq_tiled(sd_dev_model dm, SzqPhases ph, double normfact,
                                                   const double *__restrict__ psi0, double2 *__restrict__ phi) {
  const int tix = blockIdx.x;
  const uint32_t P = dm.tile_prefix[tix];
  const int64_t base = dm.tile_base[tix];
  const int t2 = dm.nup - __popc(P);
  const int len = (int)binom_g(dm, dm.LS, t2);
  const uint16_t *__restrict__ sufS = dm.suf_states + dm.suf_off[t2];
  for (int i = threadIdx.x; i < len; i += blockDim.x) {
    const uint64_t s = (uint64_t)P | ((uint64_t)sufS[i] << dm.p);
    szq_row<NCIN>(dm, ph, normfact, s, psi0, base + i, phi);
  }
}

template <int NCIN>
__global__ __launch_bounds__(256) void k_szq_generic(sd_dev_model dm, SzqPhases ph, double normfact,
                                                     const double *__restrict__ psi0, double2 *__restrict__ phi) {
  const bool full = dm.nup < 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < dm.N; idx += stride) {
    const uint64_t s = full ? (uint64_t)idx : unrank_g(dm, idx);
    szq_row<NCIN>(dm, ph, normfact, s, psi0, idx, phi);
  }
}


// =====================================================================
// Observables ("next" row f2): magnetization_per_site (src/Observables.jl:14-36) and the lag sums needed by
// connected_correlations (:44-94).  The reference accumulates the full L x L matrix <S_i S_j>; C_r only needs
//   R_r = sum_i <S_i S_{mod1(i+r,L)}> = sum_rows |psi|^2 * (L - 2*popcount(s XOR rot_r(s)))/4
// (rot_r = cyclic rotation of the L-bit configuration), i.e. L sums instead of L^2.  One read stream of psi per
// chunk of 16 accumulators; per-thread register accumulators, fixed-order two-stage reduction (deterministic).
// =====================================================================
#define SD_OBS_CHUNK 16
template <int NC, int MODE>
__device__ __forceinline__ void obs_row(const sd_dev_model &dm, uint64_t s, const double *__restrict__ psi, int64_t row,
                                        int c0, int cn, double (&acc)[SD_OBS_CHUNK]) {
  double prob;
  if (NC == 2) { const double2 v = ((const double2 *)psi)[row]; prob = v.x * v.x + v.y * v.y; }
  else { const double v = psi[row]; prob = v * v; }
  if (prob == 0.0) return;                                     // src/Observables.jl:21,53
  const int L = dm.L;
  const uint64_t mask = L >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << L) - 1);
#pragma unroll
  for (int k = 0; k < SD_OBS_CHUNK; ++k) {
    if (k < cn) {
      const int c = c0 + k;
      if (MODE == 0) acc[k] += prob * sz_of((s >> c) & 1);
      else {
        const uint64_t rot = c == 0 ? s : (((s >> c) | (s << (L - c))) & mask);
        acc[k] += prob * (0.25 * (double)(L - 2 * (int)__popcll(s ^ rot)));
      }
    }
  }
}

template <int NC, int MODE>
__global__ __launch_bounds__(256) void k_obs(sd_dev_model dm, const double *__restrict__ psi, int c0, int cn,
                                             double *__restrict__ partials) {
  __shared__ double red[32];
  double acc[SD_OBS_CHUNK];
#pragma unroll
  for (int k = 0; k < SD_OBS_CHUNK; ++k) acc[k] = 0.0;
  if (dm.p >= 0) {
    for (int t = blockIdx.x; t < dm.n_tiles; t += gridDim.x) {
      const uint32_t P = dm.tile_prefix[t];
      const int64_t base = dm.tile_base[t];
      const int t2 = dm.nup - __popc(P);
      const int len = (int)binom_g(dm, dm.LS, t2);
      const uint16_t *__restrict__ sufS = dm.suf_states + dm.suf_off[t2];
      for (int i = threadIdx.x; i < len; i += blockDim.x)
        obs_row<NC, MODE>(dm, (uint64_t)P | ((uint64_t)sufS[i] << dm.p), psi, base + i, c0, cn, acc);
    }
  } else {
    const bool full = dm.nup < 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < dm.N; idx += stride)
      obs_row<NC, MODE>(dm, full ? (uint64_t)idx : unrank_g(dm, idx), psi, idx, c0, cn, acc);
  }
#pragma unroll
  for (int k = 0; k < SD_OBS_CHUNK; k += 2) {
    double a = acc[k], b = acc[k + 1];
    block_reduce2(a, b, red);
    if (threadIdx.x == 0) { partials[(size_t)blockIdx.x * SD_OBS_CHUNK + k] = a; partials[(size_t)blockIdx.x * SD_OBS_CHUNK + k + 1] = b; }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_obs_reduce(const double *__restrict__ partials, int nblocks, double *__restrict__ out) {
  // thread (k, j): accumulator k, blocks j, j+16, ... ; then a fixed-order sum over j
  __shared__ double sm[16][SD_OBS_CHUNK];
  const int k = threadIdx.x & 15, j = threadIdx.x >> 4;
  double a = 0.0;
  for (int b = j; b < nblocks; b += 16) a += partials[(size_t)b * SD_OBS_CHUNK + k];
  sm[j][k] = a;
  __syncthreads();
  if (threadIdx.x < SD_OBS_CHUNK) {
    double t = 0.0;
    for (int jj = 0; jj < 16; ++jj) t += sm[jj][threadIdx.x];
    out[threadIdx.x] = t;
  }
}

int ensure_partials(sd_ctx *ctx, size_t doubles) {
  if (ctx->partials_cap >= doubles) return SD_OK;
  if (ctx->d_partials) (void)hipFree(ctx->d_partials);
  ctx->d_partials = nullptr; ctx->partials_cap = 0;
  SD_HIP(ctx, hipMalloc((void **)&ctx->d_partials, doubles * sizeof(double)));
  ctx->partials_cap = doubles;
  return SD_OK;
}

template <int NC, int R, int BLOCK, bool FMA>
int launch_tiled_cfg(sd_ctx *ctx, const sd_dev_model &dm, int nt, size_t shmem, double *out, const double *psi, int epi,
                     const sd_epi_args &ea, int max_len) {
  // the stamped (DIAG) instantiation exists for one configuration only and is reached through sd_debug_phase_profile
  void (*kern)(sd_dev_model, double *, const double *, int, sd_epi_args, double *, int) = k_apply_tiled<NC, R, BLOCK, FMA>;
  if constexpr (NC == 2 && BLOCK == 256 && FMA)
    if (dm.stamps) kern = k_apply_tiled<NC, R, BLOCK, FMA, true>;
  static size_t attr_set = 0;
  if (shmem > 48 * 1024 && shmem > attr_set) {
    SD_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr_set = shmem;
  }
  hipLaunchKernelGGL(kern, dim3(nt), dim3(BLOCK), shmem, ctx->stream, dm, out, psi, epi, ea, ctx->d_partials, max_len);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

template <int NC, bool FMA>
int launch_tiled(sd_ctx *ctx, const sd_dev_model &dm, int nt, size_t shmem, double *out, const double *psi, int epi,
                 const sd_epi_args &ea, int max_len) {
  // 4 rows per thread; smallest block that covers the longest tile
  constexpr int R = 4;
  if (max_len <= 256 * R) return launch_tiled_cfg<NC, R, 256, FMA>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
  if (max_len <= 512 * R) return launch_tiled_cfg<NC, R, 512, FMA>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
  if (max_len <= 1024 * R) return launch_tiled_cfg<NC, R, 1024, FMA>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
  return sd_set_err(ctx, SD_EINTERNAL, "tile longer than a workgroup can hold");
}

}  // namespace

int sd_launch_apply(sd_ctx *ctx, const sd_model *m, int dtype, void *out, const void *psi, int epi,
                    const sd_epi_args &ea) {
  if (!m->dev_ready) return sd_set_err(ctx, SD_EARG, "model has no device tables (created without a context)");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "dtype must be SD_F64 or SD_C128");
  const bool sums = (epi == SD_EPI_DOT || epi == SD_EPI_KPM || epi == SD_EPI_RESCALE_DOT);
  const sd_dev_model &dm = m->dm;
  if (dm.n_local == 0) return SD_OK;
  if (m->p >= 0) {
    const int nt = dm.n_singles, ng = dm.n_groups;
    if (sums) { int rc = ensure_partials(ctx, 2 * (size_t)(nt + ng)); if (rc) return rc; }
    const int max_len = m->max_tile_len;
    const size_t esz = dtype == SD_C128 ? 16 : 8;
    int rc = SD_OK;
    if (ng > 0) {
      const int ngen = m->group_ngen;
      const size_t shg = (size_t)((1 << ngen) * max_len + 1) * esz + 16 * SD_BIN_STRIDE * sizeof(int) + 32 * sizeof(double) + 16;
      void (*kg)(sd_dev_model, double *, const double *, int, sd_epi_args, double *, int);
      if (ngen == 3)
        kg = dtype == SD_C128 ? (m->hop_pow2 ? k_apply_grouped<2, true, 3> : k_apply_grouped<2, false, 3>)
                              : (m->hop_pow2 ? k_apply_grouped<1, true, 3> : k_apply_grouped<1, false, 3>);
      else
        kg = dtype == SD_C128 ? (m->hop_pow2 ? k_apply_grouped<2, true, 2> : k_apply_grouped<2, false, 2>)
                              : (m->hop_pow2 ? k_apply_grouped<1, true, 2> : k_apply_grouped<1, false, 2>);
      SD_HIP(ctx, hipFuncSetAttribute((const void *)kg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shg));
      hipLaunchKernelGGL(kg, dim3(ng), dim3(128 << ngen), shg, ctx->stream, dm, (double *)out, (const double *)psi, epi, ea,
                         ctx->d_partials, max_len);
      SD_HIP(ctx, hipGetLastError());
    }
    if (nt > 0) {
      const size_t shmem = (size_t)(max_len + 1) * esz + 16 * SD_BIN_STRIDE * sizeof(int) + 32 * sizeof(double) + 16;
      double *saved = ctx->d_partials;
      ctx->d_partials = saved ? saved + 2 * (size_t)ng : saved;      // singles write their partials after the groups'
      if (dtype == SD_C128)
        rc = m->hop_pow2 ? launch_tiled<2, true>(ctx, dm, nt, shmem, (double *)out, (const double *)psi, epi, ea, max_len)
                         : launch_tiled<2, false>(ctx, dm, nt, shmem, (double *)out, (const double *)psi, epi, ea, max_len);
      else
        rc = m->hop_pow2 ? launch_tiled<1, true>(ctx, dm, nt, shmem, (double *)out, (const double *)psi, epi, ea, max_len)
                         : launch_tiled<1, false>(ctx, dm, nt, shmem, (double *)out, (const double *)psi, epi, ea, max_len);
      ctx->d_partials = saved;
      if (rc) return rc;
    }
    if (sums) {
      hipLaunchKernelGGL(k_reduce_pairs, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_partials, (int64_t)(nt + ng), ctx->d_scalars);
      SD_HIP(ctx, hipGetLastError());
    }
  } else {
    int64_t nb = (dm.N + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (sums) { int rc = ensure_partials(ctx, 2 * (size_t)nb); if (rc) return rc; }
    if (dtype == SD_C128)
      hipLaunchKernelGGL(k_apply_generic<2>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dm, (double *)out,
                         (const double *)psi, epi, ea, ctx->d_partials);
    else
      hipLaunchKernelGGL(k_apply_generic<1>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dm, (double *)out,
                         (const double *)psi, epi, ea, ctx->d_partials);
    SD_HIP(ctx, hipGetLastError());
    if (sums) {
      hipLaunchKernelGGL(k_reduce_pairs, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_partials, nb, ctx->d_scalars);
      SD_HIP(ctx, hipGetLastError());
    }
  }
  return SD_OK;
}


// mode 0: out[L] = magnetization per site; mode 1: out[L] = lag sums R_r.  psi is a device vector.
int sd_launch_observable(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi, int mode, double *out_host) {
  if (!m->dev_ready) return sd_set_err(ctx, SD_EARG, "model has no device tables (created without a context)");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "dtype must be SD_F64 or SD_C128");
  const sd_dev_model &dm = m->dm;
  int nb = m->p >= 0 ? std::min(dm.n_tiles, 2048) : (int)std::min<int64_t>(2048, (dm.N + 255) / 256);
  if (nb < 1) nb = 1;
  int rc = ensure_partials(ctx, (size_t)nb * SD_OBS_CHUNK);
  if (rc) return rc;
  for (int c0 = 0; c0 < dm.L; c0 += SD_OBS_CHUNK) {
    const int cn = std::min(SD_OBS_CHUNK, dm.L - c0);
    if (dtype == SD_C128) {
      if (mode == 0) hipLaunchKernelGGL((k_obs<2, 0>), dim3(nb), dim3(256), 0, ctx->stream, dm, (const double *)psi, c0, cn, ctx->d_partials);
      else hipLaunchKernelGGL((k_obs<2, 1>), dim3(nb), dim3(256), 0, ctx->stream, dm, (const double *)psi, c0, cn, ctx->d_partials);
    } else {
      if (mode == 0) hipLaunchKernelGGL((k_obs<1, 0>), dim3(nb), dim3(256), 0, ctx->stream, dm, (const double *)psi, c0, cn, ctx->d_partials);
      else hipLaunchKernelGGL((k_obs<1, 1>), dim3(nb), dim3(256), 0, ctx->stream, dm, (const double *)psi, c0, cn, ctx->d_partials);
    }
    hipLaunchKernelGGL(k_obs_reduce, dim3(1), dim3(256), 0, ctx->stream, ctx->d_partials, nb, ctx->d_scalars);
    SD_HIP(ctx, hipGetLastError());
    double tmp[SD_OBS_CHUNK];
    rc = sd_read_scalars(ctx, 0, SD_OBS_CHUNK, tmp);
    if (rc) return rc;
    for (int k = 0; k < cn; ++k) out_host[c0 + k] = tmp[k];
  }
  return SD_OK;
}

int sd_launch_szq(sd_ctx *ctx, const sd_model *m, int dtype_in, const void *psi0, double q, void *phi) {
  if (!m->dev_ready) return sd_set_err(ctx, SD_EARG, "model has no device tables (created without a context)");
  if (dtype_in != SD_F64 && dtype_in != SD_C128) return sd_set_err(ctx, SD_EARG, "dtype must be SD_F64 or SD_C128");
  const sd_dev_model &dm = m->dm;
  if (dm.n_local == 0) return SD_OK;
  SzqPhases ph;
  // phases = exp.(im*q*(0:L-1))  (src/Hamiltonian.jl:317), computed on the host in double
  for (int r = 0; r < dm.L; ++r) { double x = q * (double)r; ph.re[r] = cos(x); ph.im[r] = sin(x); }
  const double normfact = 1.0 / sqrt((double)dm.L);
  if (m->p >= 0) {
    if (dtype_in == SD_C128)
      hipLaunchKernelGGL(k_szq_tiled<2>, dim3(dm.n_tiles), dim3(256), 0, ctx->stream, dm, ph, normfact,
                         (const double *)psi0, (double2 *)phi);
    else
      hipLaunchKernelGGL(k_szq_tiled<1>, dim3(dm.n_tiles), dim3(256), 0, ctx->stream, dm, ph, normfact,
                         (const double *)psi0, (double2 *)phi);
  } else {
    int64_t nb = (dm.N + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (dtype_in == SD_C128)
      hipLaunchKernelGGL(k_szq_generic<2>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dm, ph, normfact,
                         (const double *)psi0, (double2 *)phi);
    else
      hipLaunchKernelGGL(k_szq_generic<1>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dm, ph, normfact,
                         (const double *)psi0, (double2 *)phi);
  }
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
