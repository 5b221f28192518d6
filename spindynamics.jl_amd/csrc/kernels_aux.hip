// Auxiliary kernels of the hot path: Sz_q_vector (src/Hamiltonian.jl:307-337) and the Observables reductions
// (src/Observables.jl:14-109).  Single-stream passes over psi with the state decoded from the tile tables.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "device_common.hpp"

using namespace sd_dev;

namespace {

// =====================================================================
// Sz_q_vector  (src/Hamiltonian.jl:307-337)
// =====================================================================
struct SzqPhases { double re[SD_MAX_L + 1], im[SD_MAX_L + 1]; };

// The site sum runs in site order 1..L (src/Hamiltonian.jl:318-326).  Sites first..last-1 of configuration s are added to
// (sr, si): a tile passes the sum over its prefix sites -- the FIRST terms of every row's sum, the same for all its rows --
// and each row adds its LS suffix terms, which are the same additions in the same order as the full loop.
__device__ __forceinline__ void szq_sum(const SzqPhases &ph, uint64_t s, int first, int last, double &sr, double &si) {
  for (int r = first; r < last; ++r) {
    const double z = sz_of((s >> r) & 1);
    sr += ph.re[r] * z;
    si += ph.im[r] * z;
  }
}
template <int NCIN>
__device__ __forceinline__ void szq_store(double sr, double si, double normfact, const double *__restrict__ psi0, int64_t row,
                                          double2 *__restrict__ phi) {
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
  double pr = 0.0, pi = 0.0;
  szq_sum(ph, (uint64_t)P, 0, dm.p, pr, pi);                      // wave-uniform: once per tile
  for (int i = threadIdx.x; i < len; i += blockDim.x) {
    double sr = pr, si = pi;
    szq_sum(ph, (uint64_t)sufS[i] << dm.p, dm.p, dm.L, sr, si);
    szq_store<NCIN>(sr, si, normfact, psi0, base + i, phi);
  }
}

// Full 2^L basis (idx = state), L >= 12.  Site r+1 is bit r of the row index, and the site sum runs r = 0..L-1, so rows that
// share their LOW k index bits share the FIRST k terms of the sum.  A thread keeps one value b of those bits, forms the first
// k terms once and carries 16 rows -- 16 consecutive values of the high bits -- in registers: the remaining L-k terms are
// added site by site, each one add of +-(phase/2) per row (exactly the product phase * (+-0.5) of the reference: scaling by
// 0.5 is exact), with the sign known at compile time for the four sites that count the 16 rows and uniform for the sites
// above.  Same additions in the same order as the one-row-per-thread loop, ~45 instead of ~200 lane-operations per row;
// consecutive lanes hold consecutive rows (16-B accesses), 16 loads and 16 stores in flight per thread.
template <int NCIN>
__global__ __launch_bounds__(256) void k_szq_full(sd_dev_model dm, SzqPhases ph, double normfact, int k,
                                                  const double *__restrict__ psi0, double2 *__restrict__ phi) {
  constexpr int J = 16;
  const uint32_t nlb = (1u << k) >> 8;                           // workgroups per 16 values of the high bits (2^k / 256)
  const uint32_t bb = blockIdx.x % nlb;
  const int64_t j0 = (int64_t)(blockIdx.x / nlb) * J;
  const uint32_t b = (bb << 8) + threadIdx.x;
  double pr = 0.0, pi = 0.0;
  szq_sum(ph, (uint64_t)b, 0, k, pr, pi);
  double sr[J], si[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { sr[j] = pr; si[j] = pi; }
  const int L = dm.L;
  // sharded by the top index bits: this rank's high bits start at row_lo >> k, a multiple of 16 like j0
  const uint64_t hs0 = (uint64_t)(dm.row_lo >> k) + (uint64_t)j0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {                                  // sites k+1 .. k+4: bit t of the row counter j
    const double hr = 0.5 * ph.re[k + t], hi = 0.5 * ph.im[k + t];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      if ((j >> t) & 1) { sr[j] += hr; si[j] += hi; }
      else { sr[j] += -hr; si[j] += -hi; }
    }
  }
  for (int r = k + 4; r < L; ++r) {                              // sites above: the same for the 16 rows
    const bool up = (hs0 >> (r - k)) & 1;
    const double hr = up ? 0.5 * ph.re[r] : -(0.5 * ph.re[r]), hi = up ? 0.5 * ph.im[r] : -(0.5 * ph.im[r]);
#pragma unroll
    for (int j = 0; j < J; ++j) { sr[j] += hr; si[j] += hi; }
  }
#pragma unroll
  for (int j = 0; j < J; ++j) szq_store<NCIN>(sr[j], si[j], normfact, psi0, ((j0 + j) << k) + (int64_t)b, phi);
}

template <int NCIN>
__global__ __launch_bounds__(256) void k_szq_generic(sd_dev_model dm, SzqPhases ph, double normfact,
                                                     const double *__restrict__ psi0, double2 *__restrict__ phi) {
  const bool full = dm.nup < 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < dm.n_local; idx += stride) {
    const uint64_t s = full ? (uint64_t)(dm.row_lo + idx) : unrank_g(dm, idx);     // full basis: state = global row
    double sr = 0.0, si = 0.0;
    szq_sum(ph, s, 0, dm.L, sr, si);
    szq_store<NCIN>(sr, si, normfact, psi0, idx, phi);
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
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < dm.n_local; idx += stride)
      obs_row<NC, MODE>(dm, full ? (uint64_t)(dm.row_lo + idx) : unrank_g(dm, idx), psi, idx, c0, cn, acc);
  }
#pragma unroll
  for (int k = 0; k < SD_OBS_CHUNK; k += 2) {
    double a = acc[k], b = acc[k + 1];
    block_reduce2(a, b, red);
    if (threadIdx.x == 0) { partials[(size_t)blockIdx.x * SD_OBS_CHUNK + k] = a; partials[(size_t)blockIdx.x * SD_OBS_CHUNK + k + 1] = b; }
    __syncthreads();
  }
}

// ---- one-pass forms for the tiled layouts (round 3) ----
// A tile's rows share the "uniform" sites (sector: the prefix sites 1..p; full basis: the sites above the 2^10 rows of a
// tile) and differ in the "variable" ones (suffix sites / the low 10 index bits).
//   MODE 0, magnetisation: sum_rows prob * sz(site).  Variable sites: one signed add per row and site; uniform sites: the
//     thread adds its rows' total probability of the tile, signed by the tile's bit, once per tile -- all L sites in ONE pass
//     over psi (the chunked kernel above needs ceil(L/16) passes and 4 lane-operations per row and site).
//   MODE 1, lag sums R_r = sum_rows prob * (L - 2 popcount(s ^ rot_r s)) / 4: popcount(s ^ rot_r s) = popcount(s ^ rot_{L-r} s),
//     so only r = 1 .. L/2 are accumulated (R_0 = L/4 * sum prob), in 32-bit arithmetic when L <= 32: one pass as well.
// The sums run in another order than the reference's row loop (src/Observables.jl:19-30, 56-72): tolerance 1e-13 in the tests.
#define SD_OBS2_COLS 64
template <int NC, bool FULL, int MODE, bool WIDE>
__global__ __launch_bounds__(256) void k_obs2(sd_dev_model dm, const double *__restrict__ psi, double *__restrict__ partials) {
  constexpr int NV = 16, NU = 32, NLAG = 32;
  __shared__ double red[4][SD_OBS2_COLS];
  double accV[MODE == 0 ? NV : 1], accU[MODE == 0 ? NU : 1], lag[MODE == 1 ? NLAG : 1];
  double total = 0.0;
#pragma unroll
  for (int k = 0; k < (MODE == 0 ? NV : 1); ++k) accV[k] = 0.0;
#pragma unroll
  for (int k = 0; k < (MODE == 0 ? NU : 1); ++k) accU[k] = 0.0;
#pragma unroll
  for (int k = 0; k < (MODE == 1 ? NLAG : 1); ++k) lag[k] = 0.0;
  const int L = dm.L;
  const int nv = FULL ? 10 : dm.LS, nu = L - nv;           // variable / uniform site counts
  const int vsh = FULL ? 0 : dm.p;                         // bit position of the first variable site
  const int nl = L / 2;
  const int64_t ntiles = FULL ? (dm.n_local >> 10) : (int64_t)dm.n_tiles;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    uint64_t ubits;                                        // the uniform sites' bits, site order, from bit 0
    int64_t base;
    int len;
    const uint16_t *__restrict__ sufS = nullptr;
    if (FULL) {
      base = t << 10; len = 1024;
      ubits = (uint64_t)((dm.row_lo + base) >> 10);
    } else {
      const uint32_t P = dm.tile_prefix[t];
      base = dm.tile_base[t];
      const int t2 = dm.nup - __popc(P);
      len = (int)binom_g(dm, dm.LS, t2);
      sufS = dm.suf_states + dm.suf_off[t2];
      ubits = P;
    }
    double tsum = 0.0;
    for (int i = threadIdx.x; i < len; i += 256) {
      double prob;
      if (NC == 2) { const double2 v = ((const double2 *)psi)[base + i]; prob = v.x * v.x + v.y * v.y; }
      else { const double v = psi[base + i]; prob = v * v; }
      const uint32_t var = FULL ? (uint32_t)i : (uint32_t)sufS[i];
      tsum += prob;
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
          if (k < nv) accV[k] += ((var >> k) & 1u) ? prob : -prob;
      } else {
        const uint64_t s64 = FULL ? ((ubits << 10) | var) : (ubits | ((uint64_t)var << vsh));
        if (!WIDE) {
          const uint32_t s = (uint32_t)s64, mask = L >= 32 ? 0xffffffffu : ((1u << L) - 1u);
#pragma unroll
          for (int r = 1; r <= NLAG; ++r)
            if (r <= nl) {
              const uint32_t rot = ((s >> r) | (s << (L - r))) & mask;
              lag[r - 1] += prob * (double)(L - 2 * (int)__popc(s ^ rot));
            }
        } else {
          const uint64_t mask = ((uint64_t)1 << L) - 1;
#pragma unroll
          for (int r = 1; r <= NLAG; ++r)
            if (r <= nl) {
              const uint64_t rot = ((s64 >> r) | (s64 << (L - r))) & mask;
              lag[r - 1] += prob * (double)(L - 2 * (int)__popcll(s64 ^ rot));
            }
        }
      }
    }
    total += tsum;
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < NU; ++k)
        if (k < nu) accU[k] += ((ubits >> k) & 1) ? tsum : -tsum;
    }
  }
  // columns: MODE 0: [0, nv) variable sites, [16, 16+nu) uniform sites; MODE 1: [0, nl) lags 1..nl; column 63: sum prob
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  auto put = [&](int col, double a) {
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if (lane == 0) red[wv][col] = a;
  };
  if (threadIdx.x < 4 * SD_OBS2_COLS) (&red[0][0])[threadIdx.x] = 0.0;
  __syncthreads();
  if (MODE == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) put(k, accV[k]);
#pragma unroll
    for (int k = 0; k < NU; ++k) put(NV + k, accU[k]);
  } else {
#pragma unroll
    for (int k = 0; k < NLAG; ++k) put(k, lag[k]);
  }
  put(SD_OBS2_COLS - 1, total);
  __syncthreads();
  if (threadIdx.x < SD_OBS2_COLS)
    partials[(size_t)blockIdx.x * SD_OBS2_COLS + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// out[c] = sum over blocks of partials[block][c], fixed order (16 strided partial sums per column, then their sum)
__global__ __launch_bounds__(1024) void k_obs2_reduce(const double *__restrict__ partials, int nblocks, double *__restrict__ out) {
  __shared__ double sm[16][SD_OBS2_COLS];
  const int c = threadIdx.x & 63, j = threadIdx.x >> 6;
  double a = 0.0;
  for (int b = j; b < nblocks; b += 16) a += partials[(size_t)b * SD_OBS2_COLS + c];
  sm[j][c] = a;
  __syncthreads();
  if (threadIdx.x < SD_OBS2_COLS) {
    double t = 0.0;
    for (int jj = 0; jj < 16; ++jj) t += sm[jj][threadIdx.x];
    out[threadIdx.x] = t;
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



// ---- the list-order diagonal of every local row, once per model (sd_dev_model::diag_cache) ----
__global__ __launch_bounds__(256) void k_build_diag(sd_dev_model dm, double *__restrict__ out) {
  for (int t = blockIdx.x; t < dm.n_tiles; t += gridDim.x) {
    const uint32_t P = dm.tile_prefix[t];
    const int64_t base = dm.tile_base[t];
    const int t2 = dm.nup - __popc(P);
    const int len = (int)binom_g(dm, dm.LS, t2);
    const uint16_t *__restrict__ sufS = dm.suf_states + dm.suf_off[t2];
    for (int i = threadIdx.x; i < len; i += blockDim.x) out[base + i] = diag_of(dm, (uint64_t)P | ((uint64_t)sufS[i] << dm.p));
  }
}

// ---- sharded plans: pack the tiles peers need into the contiguous send buffer; fill a local vector by GLOBAL index ----
template <int NC>
__global__ __launch_bounds__(256) void k_pack(sd_dev_model dm, const double *__restrict__ psi_, double *__restrict__ send_) {
  using V = typename VT<NC>::type;
  const V *__restrict__ psi = reinterpret_cast<const V *>(psi_);
  V *__restrict__ send = reinterpret_cast<V *>(send_);
  for (int t = blockIdx.x; t < dm.n_pack; t += gridDim.x) {
    const int64_t src = dm.pack_src[t], dst = dm.pack_dst[t];
    const int len = dm.pack_len[t];
    for (int i = threadIdx.x; i < len; i += blockDim.x) send[dst + i] = psi[src + i];
  }
}

__global__ __launch_bounds__(256) void k_fill_randn_tiles(sd_dev_model dm, double *__restrict__ x, int per, uint64_t seed) {
  for (int t = blockIdx.x; t < dm.n_tiles; t += gridDim.x) {
    const int64_t lb = dm.tile_base[t], gb = dm.tile_gbase[t];
    const int len = (int)binom_g(dm, dm.LS, dm.nup - __popc(dm.tile_prefix[t]));
    for (int i = threadIdx.x; i < len * per; i += blockDim.x)
      x[lb * per + i] = sd_randn_at(seed, (uint64_t)(gb * per + i));
  }
}

// ---- create_spin_operator(site, op)  (src/Hamiltonian.jl:49-136) in gather form ----
// op: 0 z, 1 plus, 2 minus, 3 x, 4 y.  z is diagonal; the others change the magnetisation and exist in the full basis
// only (the reference throws in a sector, :68-73), where the flipped configuration is idx ^ (1 << (site-1)).
// out has psi's element type; y needs ComplexF64 (the host rejects a real psi, where the reference raises InexactError).
template <int NC>
__global__ __launch_bounds__(256) void k_spin_op(sd_dev_model dm, int bit_pos, int op, const double *__restrict__ psi,
                                                 double *__restrict__ out) {
  const bool full = dm.nup < 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < dm.N; idx += stride) {
    uint64_t s;
    if (full) s = (uint64_t)idx;
    else s = unrank_g(dm, idx);
    const bool up = (s >> bit_pos) & 1;
    double xr, xi = 0.0;
    if (op == 0) {
      const double z = up ? 0.5 : -0.5;
      xr = z * psi[idx * NC];
      if (NC == 2) xi = z * psi[idx * NC + 1];
    } else {
      const int64_t j = idx ^ ((int64_t)1 << bit_pos);        // full basis only
      const double pr = psi[j * NC], pi = NC == 2 ? psi[j * NC + 1] : 0.0;
      if (op == 1) { xr = up ? pr : 0.0; xi = up ? pi : 0.0; }            // S+ : result[flip(s)] += psi[s] for bit(s) = 0
      else if (op == 2) { xr = up ? 0.0 : pr; xi = up ? 0.0 : pi; }       // S-
      else if (op == 3) { xr = 0.5 * pr; xi = 0.5 * pi; }                 // Sx
      else {                                                              // Sy: -0.5i from a down source, +0.5i from an up source
        const double c = up ? -0.5 : 0.5;                                 // result[j'] += c*i * psi[s], j' has the opposite bit of s
        xr = -c * pi; xi = c * pr;
      }
    }
    out[idx * NC] = xr;
    if (NC == 2) out[idx * NC + 1] = xi;
  }
}
}  // namespace

int sd_k_build_diag(const sd_dev_model &dm, double *out) {
  if (dm.n_tiles <= 0) return SD_OK;
  hipLaunchKernelGGL(k_build_diag, dim3((unsigned)std::min(dm.n_tiles, 1 << 16)), dim3(256), 0, 0, dm, out);
  if (hipGetLastError() != hipSuccess) return SD_EHIP;
  return hipDeviceSynchronize() == hipSuccess ? SD_OK : SD_EHIP;
}

// mode 0: out[L] = magnetization per site; mode 1: out[L] = lag sums R_r.  psi is a device vector.
int sd_launch_observable(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi, int mode, double *out_host) {
  if (!m->dev_ready) return sd_set_err(ctx, SD_EARG, "model has no device tables (created without a context)");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "dtype must be SD_F64 or SD_C128");
  const sd_dev_model &dm = m->dm;
  const bool tiled = m->p >= 0 && dm.n_tiles > 0 && dm.LS <= 15 && dm.p <= 32;
  const bool fullt = m->p < 0 && m->full_ls > 0 && dm.L >= 12 && dm.L <= 40 && (dm.n_local >> 10) > 0;
  if ((tiled || fullt) && !getenv("SD_OBS_CHUNKED")) {
    // one pass over psi for all L sites / all lags (k_obs2)
    const int64_t ntiles = fullt ? (dm.n_local >> 10) : (int64_t)dm.n_tiles;
    const int nb2 = (int)std::min<int64_t>(ntiles, 2048);
    int rc2 = sd_ensure_partials(ctx, (size_t)nb2 * SD_OBS2_COLS + SD_OBS2_COLS);
    if (rc2) return rc2;
    double *res = ctx->d_partials + (size_t)nb2 * SD_OBS2_COLS;
    const bool wide = dm.L > 32, c = dtype == SD_C128;
#define SD_OBS2_LAUNCH(NC_, FULL_, MODE_, WIDE_)                                                                              \
    hipLaunchKernelGGL((k_obs2<NC_, FULL_, MODE_, WIDE_>), dim3(nb2), dim3(256), 0, ctx->stream, dm, (const double *)psi, ctx->d_partials)
    if (mode == 0) {
      if (fullt) { if (c) SD_OBS2_LAUNCH(2, true, 0, false); else SD_OBS2_LAUNCH(1, true, 0, false); }
      else { if (c) SD_OBS2_LAUNCH(2, false, 0, false); else SD_OBS2_LAUNCH(1, false, 0, false); }
    } else if (wide) {
      if (fullt) { if (c) SD_OBS2_LAUNCH(2, true, 1, true); else SD_OBS2_LAUNCH(1, true, 1, true); }
      else { if (c) SD_OBS2_LAUNCH(2, false, 1, true); else SD_OBS2_LAUNCH(1, false, 1, true); }
    } else {
      if (fullt) { if (c) SD_OBS2_LAUNCH(2, true, 1, false); else SD_OBS2_LAUNCH(1, true, 1, false); }
      else { if (c) SD_OBS2_LAUNCH(2, false, 1, false); else SD_OBS2_LAUNCH(1, false, 1, false); }
    }
#undef SD_OBS2_LAUNCH
    hipLaunchKernelGGL(k_obs2_reduce, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_partials, nb2, res);
    SD_HIP(ctx, hipGetLastError());
    double h[SD_OBS2_COLS];
    SD_HIP(ctx, hipMemcpyAsync(h, res, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int L = dm.L, nv = fullt ? 10 : dm.LS, nu = L - nv;
    if (mode == 0) {
      for (int k = 0; k < nv; ++k) out_host[(fullt ? 0 : dm.p) + k] = 0.5 * h[k];          // prob * (+-1/2)
      for (int k = 0; k < nu; ++k) out_host[(fullt ? 10 : 0) + k] = 0.5 * h[16 + k];
    } else {
      out_host[0] = 0.25 * (double)L * h[SD_OBS2_COLS - 1];
      for (int r = 1; r <= L / 2; ++r) out_host[r] = out_host[L - r] = 0.25 * h[r - 1];
    }
    return SD_OK;
  }
  int nb = m->p >= 0 ? std::min(dm.n_tiles, 2048) : (int)std::min<int64_t>(2048, (dm.n_local + 255) / 256);
  if (nb < 1) nb = 1;
  int rc = sd_ensure_partials(ctx, (size_t)nb * SD_OBS_CHUNK);
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

int sd_launch_pack(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi, void *sendbuf) {
  if (!m->dev_ready) return sd_set_err(ctx, SD_EARG, "model has no device tables (created without a context)");
  const sd_dev_model &dm = m->dm;
  if (dm.n_pack == 0) return SD_OK;
  const int nb = std::min(dm.n_pack, 8192);
  if (dtype == SD_C128) hipLaunchKernelGGL(k_pack<2>, dim3(nb), dim3(256), 0, ctx->stream, dm, (const double *)psi, (double *)sendbuf);
  else hipLaunchKernelGGL(k_pack<1>, dim3(nb), dim3(256), 0, ctx->stream, dm, (const double *)psi, (double *)sendbuf);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

int sd_launch_fill_randn_local(sd_ctx *ctx, const sd_model *m, int dtype, void *x, uint64_t seed) {
  if (!m->dev_ready) return sd_set_err(ctx, SD_EARG, "model has no device tables (created without a context)");
  const sd_dev_model &dm = m->dm;
  const int per = dtype == SD_C128 ? 2 : 1;
  if (m->p < 0) return sd_k_fill_randn(ctx, (double *)x, dm.n_local * per, seed, (uint64_t)(dm.row_lo * per));   // keyed by the global element index
  if (dm.n_tiles == 0) return SD_OK;
  hipLaunchKernelGGL(k_fill_randn_tiles, dim3(std::min(dm.n_tiles, 8192)), dim3(256), 0, ctx->stream, dm, (double *)x, per, seed);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

int sd_launch_spin_op(sd_ctx *ctx, const sd_model *m, int dtype, int site, int op, const void *psi, void *out) {
  if (!m->dev_ready) return sd_set_err(ctx, SD_EARG, "model has no device tables (created without a context)");
  const sd_dev_model &dm = m->dm;
  if (m->nranks > 1 && m->p < 0) return sd_set_err(ctx, SD_EARG, "spin operators act on an unsharded full basis (the flipped row may live on another rank)");
  int64_t nb = (dm.N + 255) / 256;
  if (nb > 16384) nb = 16384;
  if (nb < 1) nb = 1;
  if (dtype == SD_C128) hipLaunchKernelGGL(k_spin_op<2>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dm, site - 1, op, (const double *)psi, (double *)out);
  else hipLaunchKernelGGL(k_spin_op<1>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dm, site - 1, op, (const double *)psi, (double *)out);
  SD_HIP(ctx, hipGetLastError());
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
  } else if (m->full_ls > 0 && dm.L >= 12 && (dm.n_local >> 12) > 0 && !getenv("SD_SZQ_FULL_GENERIC")) {
    // full basis: k low index bits per thread-constant prefix sum (2^k >= 256 rows per workgroup pass), 16 values of the high
    // bits per thread (a rank of a sharded basis owns 2^(L-d) rows: at least 16 x 256 of them, else the row loop below)
    int k = dm.L / 2;
    if (k < 8) k = 8;
    while (k > 8 && (dm.n_local >> k) < 16) --k;
    const int64_t n_high = dm.n_local >> k;                      // a power of two >= 16
    const int64_t nb = (n_high / 16) * (int64_t)((1u << k) >> 8);
    if (dtype_in == SD_C128)
      hipLaunchKernelGGL(k_szq_full<2>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dm, ph, normfact, k,
                         (const double *)psi0, (double2 *)phi);
    else
      hipLaunchKernelGGL(k_szq_full<1>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dm, ph, normfact, k,
                         (const double *)psi0, (double2 *)phi);
  } else {
    int64_t nb = (dm.n_local + 255) / 256;
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
