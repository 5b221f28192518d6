// H|psi> apply kernels for gfx950 (MI355X).  Replaces the reference's
// Threads.@threads row loop of apply_H! (src/Hamiltonian.jl:211-273), the
// separate rescale pass of apply_rescaled_H! (:286-301, fused here into the
// store), and Sz_q_vector (:307-337).
//
// Device paths:
//  * k_apply_tiled   -- fixed-nup sector.  One workgroup per TILE (all rows
//    sharing a prefix configuration of sites 1..p, see sd_internal.hpp).  The
//    tile's psi is staged once in LDS; hops on bonds inside the suffix are LDS
//    reads at idx +- C(LS-a-1,u) (no search, no hash); hops on prefix bonds
//    are coalesced streams from another whole tile at the same in-tile offset;
//    the straddling bond is a coalesced stream from half a tile.
//  * k_apply_fulltile -- full 2^L basis, L >= 12: idx = state, 2^10 consecutive
//    rows are a tile; chain bonds inside are LDS reads at i ^ (3 << (a-1)),
//    higher bonds whole-tile streams from T ^ (3 << b).
//  * k_apply_generic -- any other model (L up to 63, arbitrary bonds, huge
//    prefix spaces): one row per thread, combinadic unrank / rank per hop.
// sd_launch_apply picks the path and, for tiled plans with many tiles, issues one
// launch per tile length class (64 / 128 / 256 / 512 / 1024 threads) and per part
// (interior / boundary tiles of a sharded plan).
//
// Per-row operation order follows the reference exactly (fields, zz in list
// order, value = diag*psi[idx], then hops in list order, value += J*psi[idx'])
// and the file is compiled with -ffp-contract=off, so a plain apply is
// bit-identical to the CPU oracle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "device_common.hpp"


using namespace sd_dev;

namespace {

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
//   3. the first far-bond stream is requested; own rows and the binomial table
//      go to LDS and the only barrier follows at once (all waves are at the same
//      point; afterwards each wave runs on its own clock);
//   4. the far-bond partner rows are streamed with a two-deep ping-pong
//      pipeline (loads of bond k+1 in flight while bond k is accumulated);
//   5. the gathers of the first general bond (e.g. the periodic (L,1) bond) are
//      requested into the idle stream registers; the suffix bonds are LDS reads at
//      idx +- C(LS-a-1,u); then the general bonds;
//   6. fused epilogue + store.
// Accumulation order per row is the reference's bond order 1..L-1.  When every
// NN hop amplitude is a power of two (XXZChain default 0.5) J*psi is exact and
// acc + J*psi is evaluated with one fma (bit-identical to the unfused form).
// minimum waves per SIMD the register allocation must allow (build-time experiments: -DSD_LB_C128=6 -DSD_LB_F64=5 spill)
#ifndef SD_SKIP_DEAD_ROWS
#define SD_SKIP_DEAD_ROWS 1
#endif
#ifndef SD_SKIP_DEAD_C128
#define SD_SKIP_DEAD_C128 0     // (round 4, wave-contiguous rows: A/B in profiles/ablation_r04.md)
#endif
#ifndef SD_SKIP_DEAD_C128_SHORT
#define SD_SKIP_DEAD_C128_SHORT 1   // ... for the one-wave workgroups of the short tiles only (dilute sectors: most tiles fill one or two row groups)
#endif
#ifndef SD_LB_C128
#define SD_LB_C128 4
#endif
#ifndef SD_LB_F64
#define SD_LB_F64 5
#endif
#ifndef SD_LB_GEN
#define SD_LB_GEN 5         // the form with the general-bond plan (GEN)
#endif
#ifndef SD_GEN_DEPTH
#define SD_GEN_DEPTH 1      // register sets of its prefix-prefix streams
#endif
#ifndef SD_FAR_DEPTH
#define SD_FAR_DEPTH 2      // register sets of the far-bond streams (1: no ping-pong, fewer registers, more waves)
#endif
// Rows of a thread.  A WAVE owns R*64 consecutive rows of the tile, row group r of a lane is row wave*R*64 + r*64 + lane:
// consecutive lanes read consecutive rows (coalesced) and the R row groups of a stream differ by a compile-time byte offset
// of r*64*sizeof(V) <= 3584, which fits the buffer instructions' 12-bit immediate -- no per-row address arithmetic.
//
// PK (LS <= 12, every default plan): the suffix bonds' partner rows come from the packed per-sector table dm.suf_part (one
// 16-byte load per row, L2/L1 resident: 64 KB for all sectors): a bond costs a shift, a mask, an LDS read and the
// multiply-add -- the binomial form (!PK: popcount, LDS binomial look-up, two bit tests, signed offset, select) cost ~14
// lane-operations per row and bond, half of the kernel's VALU stream (VERDICT r03, weak point 5).
template <int NC, int R, int BLOCK, bool FMA, bool PK, bool DIAG = false, bool GEN = false>
__global__ __launch_bounds__(BLOCK, (GEN ? SD_LB_GEN : NC == 2 ? SD_LB_C128 : SD_LB_F64)) void k_apply_tiled(sd_dev_model dm, double *__restrict__ out_,
                                                       const double *__restrict__ psi_, int epi, sd_epi_args ea,
                                                       double *__restrict__ partials, int max_len) {
  using V = typename VT<NC>::type;
  constexpr uint32_t ES = sizeof(V);
  constexpr int WR = R * 64;                                   // rows of a wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V *tile = reinterpret_cast<V *>(smem);                       // row 0: all zero; rows 1..len: the tile's psi
  int *lbin = reinterpret_cast<int *>(smem + (size_t)(max_len + 1) * sizeof(V));   // !PK only
  double *red = reinterpret_cast<double *>(lbin + 16 * SD_BIN_STRIDE);
  V *tile2 = reinterpret_cast<V *>(red + 32);                  // wrap bond (dm.wrap_hop): row 0 zero, rows 1.. the partner tile

  if (ea.batch > 1) {                                          // vector blockIdx.y of a batch (sd_epi_args::batch)
    const int64_t boff = (int64_t)blockIdx.y * ea.bstride * NC;      // in doubles
    psi_ += boff; out_ += boff;
    if (ea.prev) ea.prev = (const double *)ea.prev + boff;
    if (ea.phi) ea.phi = (const double *)ea.phi + boff;
    if (ea.accv) ea.accv = (double *)ea.accv + boff;
    partials += 2 * (size_t)blockIdx.y * (size_t)dm.n_singles;
    ea.negate = (ea.negate >> blockIdx.y) & 1;                 // a batch carries one negate bit per vector
  }
  const V *__restrict__ psi = reinterpret_cast<const V *>(psi_);
  const V *__restrict__ halo = reinterpret_cast<const V *>(ea.halo);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int i0 = (tid >> 6) * WR + lane;                       // the thread's first row; row group r: i0 + 64 r
  const int tix = blockIdx.x + dm.tile_off;
  const sd_tile_rec rec = dm.single_rec[tix];       // one 32-byte scalar load: no dependent table look-ups at start-up
  const uint32_t P = rec.prefix;
  const int64_t base = rec.base;
  const int p = dm.p, LS = dm.LS;
  const int len = rec.len;
  const int nU = rec.nU;                            // rows whose first suffix site is up
  const int nn = dm.nn_hops;
  // GEN: the model's general bonds run from the host-resolved plan dm.gen (section 4b); the suffix configurations are then needed
  // for the diagonal at most
  static_assert(!GEN || (PK && !DIAG), "general-bond plan: packed tables only");
  const bool need_sig = GEN ? (dm.need_sig_gen != 0) : (!PK || dm.need_sig);
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

  // ---- 1. request own rows (rows >= len read 0 through the range check), the partner table and, if needed, the configurations ----
  V own[R];
  uint32_t sig[R];
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  u4 pt[PK ? R : 1];
  uint32_t dg[R];
  const uint32_t off0 = (uint32_t)i0 * ES;   // byte offset of the thread's first row inside a tile-sized stream; row group r: + r*64*ES
  {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(psi + base, (uint32_t)len * ES);
#pragma unroll
    for (int r = 0; r < R; ++r) buf_load(own[r], rs, off0 + (uint32_t)(r * 64) * ES);
    if (PK && !need_sig && !dm.diag_cache) {
      // one byte per row: anti-parallel pairs inside the suffix (bits 0..3) and the first suffix site (bit 4)
      const __amdgpu_buffer_rsrc_t rg = make_rsrc(dm.suf_dg + rec.suf_off, (uint32_t)len);
#pragma unroll
      for (int r = 0; r < R; ++r) dg[r] = (uint32_t)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rg, (uint32_t)i0 + (uint32_t)(r * 64), 0, 0);
    }
    if (need_sig) {
      const __amdgpu_buffer_rsrc_t rsig = make_rsrc(dm.suf_states + rec.suf_off, (uint32_t)len * 2u);
#pragma unroll
      for (int r = 0; r < R; ++r) sig[r] = buf_load_u16(rsig, (uint32_t)i0 * 2u + (uint32_t)(r * 64) * 2u);   // rows >= len: 0 -> no suffix bond flips
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) sig[r] = 0;
    }
  }

  // ---- 2. per-wave list of flippable far bonds (lane b-1 <-> bond b <= p-1, lane p-1 <-> straddle) ----
  // The partner bases were resolved on the host (basis.cpp, far_base; -1: the bond cannot flip): one coalesced load per wave
  // that needs nothing but the block index, so it is in flight together with the tile record.  Everything else a bond needs is
  // wave-uniform and scalar: its amplitude (a scalar load), and for the straddling bond the window of rows that have the hop.
  uint64_t fmask = 0;
  int64_t my_base = -1;
  const uint32_t bitp = p >= 1 ? ((P >> (p - 1)) & 1u) : 0u;
  if (nn > 0 && p >= 1) {
    const int b = lane + 1;
    if (b <= p) my_base = dm.far_base[(size_t)tix * (size_t)p + (size_t)lane];
    bool fl = my_base >= 0;
    if (DIAG) {
      if (b <= p - 1) fl = fl && !((dm.dbg & 1) || b <= (dm.dbg >> 8));      // dbg >> 8 = m: prefix bonds 1..m left to another pass (two-pass probe)
      else if (b == p) fl = fl && !(dm.dbg & 2);
    }
    fmask = __ballot(fl);
  }
  // bit p of P up:   our rows with first suffix site down (i >= nU) <-> partner rows i - nU
  // bit p of P down: our rows with first suffix site up   (i <  nU) <-> partner rows nUq + i (the shift is in the base)
  const int st_lo = bitp ? nU : 0, st_n = bitp ? len - nU : nU;
  auto get_bond = [&](int ln) {
    FarBond fb;
    fb.base = rl64(my_base, ln); fb.J = dm.hop_J[ln];
    const bool st = ln == p - 1;
    fb.lo = st ? st_lo : 0; fb.n = st ? st_n : len;
    return fb;
  };
  // rows outside [lo, lo+n) wrap to a huge unsigned offset or exceed n*ES: the load returns 0 and J*0 leaves acc unchanged
  // Row groups of this wave that lie wholly beyond the tile's last row issue no far-bond loads at all: their stream registers
  // stay zero.  The range check would return zeros for them anyway, but every such load still costs the address unit its
  // cycles (TA busy 70 % of the launch, ablation_r03.md section 5).
  // Float64 only: measured -2...3 % there, but +5 % for ComplexF64 (round 3).
  constexpr bool SKIP_DEAD = SD_SKIP_DEAD_ROWS && (NC == 1 || SD_SKIP_DEAD_C128 || (SD_SKIP_DEAD_C128_SHORT == 1 && BLOCK == 64) || (SD_SKIP_DEAD_C128_SHORT == 2 && BLOCK <= 128));
  uint32_t live = 0;
  if (SKIP_DEAD) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if ((tid & ~63) / 64 * WR + r * 64 < len) live |= 1u << r;
    live = (uint32_t)__builtin_amdgcn_readfirstlane((int)live);
  }
  auto issue = [&](const FarBond &fb, V(&v)[R]) {
    // partner tile lives in the owned rows, or (sharded plans) in the halo imported from its owner
    const V *__restrict__ pb = (halo && fb.base >= dm.n_local) ? halo + (fb.base - dm.n_local) : psi + fb.base;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(pb, (uint32_t)fb.n * ES);
    const uint32_t rel = off0 - (uint32_t)fb.lo * ES;      // wraps for rows below the window: the range check returns 0
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (!SKIP_DEAD || ((live >> r) & 1u)) buf_load(v[r], rs, rel + (uint32_t)(r * 64) * ES);
  };

  SD_STAMP(1);
  // first far bond in flight before the own rows have even arrived
  V va[R], vb[R];
  if (SKIP_DEAD) {
#pragma unroll
    for (int r = 0; r < R; ++r) { va[r] = V{}; vb[r] = V{}; }
  }
  FarBond fa{}, fbb{};
  uint64_t mk = fmask;
  bool have_a = false;
  auto next_lane = [&](uint64_t &m_) { const int ln = __builtin_ctzll(m_); m_ &= m_ - 1; return ln; };   // ascending bond order
  if (mk) { fa = get_bond(next_lane(mk)); issue(fa, va); have_a = true; }

  // ---- diagonal (needs own) ----
  V acc[R];
  if (dm.diag_cache) {                                         // general couplings: the diagonal was summed once per model
    double dd[R];
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dm.diag_cache + base, (uint32_t)len * (uint32_t)sizeof(double));
#pragma unroll
    for (int r = 0; r < R; ++r) buf_load(dd[r], rd, (uint32_t)i0 * (uint32_t)sizeof(double) + (uint32_t)(r * 64) * (uint32_t)sizeof(double));
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = vscale(dd[r], own[r]);
  } else if (!need_sig) {
    // uniform chain zz, exact partial sums (diag_mode 1; diag_of's q * (n_zz - 2 anti)): anti-parallel pairs inside the prefix
    // (once per thread) + the pair across the prefix | suffix cut + those inside the suffix (4 bits of the packed table)
    const int antiP = p >= 2 ? __popc((P ^ (P >> 1)) & ((1u << (p - 1)) - 1u)) : 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int anti = antiP + (int)(dg[r] & 15u) + (p >= 1 ? (int)(bitp ^ (dg[r] >> 4)) : 0);
      if (dm.n_zz_nn == 0) anti = 0;
      acc[r] = vscale(dm.diag_q * (double)(dm.n_zz - 2 * anti), own[r]);
    }
  } else if (dm.diag_mode == 0) {     // no cache (SD_DIAG_CACHE=0 or no memory): the prefix part of the list-order sum once per thread, the rest per row
    const DiagHead dh = diag_head(dm, P, p);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t s = (uint64_t)P | ((uint64_t)sig[r] << p);
      acc[r] = vscale(diag_tail(dm, dh, s, p), own[r]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t s = (uint64_t)P | ((uint64_t)sig[r] << p);
      acc[r] = vscale(diag_of(dm, s), own[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (i0 + r * 64 < len) tile[1 + i0 + r * 64] = own[r];
  if (tid == 0) tile[0] = V{};         // the zero row read by lanes whose suffix bond is not flippable
  if (!PK)
    for (int k = tid; k < 16 * SD_BIN_STRIDE; k += BLOCK) {
      int n = k / SD_BIN_STRIDE, kk = k - n * SD_BIN_STRIDE;
      lbin[k] = (int)binom_g(dm, n, kk);
    }

  SD_STAMP(2);
  // The only barrier: own rows (and binomials) are in LDS.  Placed BEFORE the far-bond streams (all waves are at the same
  // point here, their own rows have just arrived), so that afterwards every wave runs its stream and suffix phases on
  // its own clock and one wave's LDS/VALU phase overlaps its neighbours' memory phase.
  __syncthreads();
  // ---- 3. far bonds: ping-pong pipeline, accumulation in bond order ----
  // (The issues are conditional, so the compiler cannot count the loads in flight and drains them all before each
  // accumulation: the two register sets overlap less than the source suggests.  Three rewrites with counted waits -- loads
  // always issued, EMPTY bonds at the end -- were measured and lose or tie: profiles/ablation_r03.md section 3.)
#if SD_FAR_DEPTH == 1
  // one register set: a wave waits for each bond's rows before it asks for the next; the overlap comes from the other waves
  // of the SIMD (the registers saved buy two more of them)
  while (have_a) {
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = accum<FMA>(acc[r], fa.J, va[r]);
    have_a = false;
    if (mk) { fa = get_bond(next_lane(mk)); issue(fa, va); have_a = true; }
  }
  (void)fbb; (void)vb;
#else
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
#endif
  SD_STAMP(3);
  // The packed partner table of the suffix bonds is requested only now: its 4 registers per row would otherwise be live
  // across the stream phase and cost a wave per SIMD (89 instead of 74 VGPRs for ComplexF64).  An L2 / L1 hit.
  if (PK && nn > 0) {
    const __amdgpu_buffer_rsrc_t rp = make_rsrc(dm.suf_part + 4 * (size_t)rec.suf_off, (uint32_t)len * 16u);
#pragma unroll
    for (int r = 0; r < R; ++r) pt[PK ? r : 0] = __builtin_amdgcn_raw_buffer_load_b128(rp, (uint32_t)i0 * 16u + (uint32_t)(r * 64) * 16u, 0, 0);
  }
  // Wrap bond (the first general bond, one site in the prefix and one in the suffix: the periodic chain's (L, 1)).  Its partner
  // rows are scattered over the whole of ONE tile, P ^ (1 << wrap_pb) -- as a per-row gather every fetched line is half used and the
  // address unit pays per line (+19 % on the apply).  The partner tile is streamed instead, coalesced like a far bond, into a
  // second LDS image (the stream registers are free now) and read there after the suffix bonds, at the row the packed table names.
  // (not for the 8-row Float64 form: its 96 registers -- five waves -- have no room, the code would spill; it keeps the gather)
  const bool wrap = !GEN && PK && !(NC == 1 && R == 8) && nn > 0 && dm.wrap_hop == nn;
  bool wrap_on = false;
  if (wrap) {
    const uint32_t Q = P ^ (1u << dm.wrap_pb);
    const int t2q = dm.nup - __popc(Q);
    if (t2q >= 0 && t2q <= LS) {
      wrap_on = true;
      const int64_t qb = dm.addr[Q];
      const int lenq = (int)dm.binom[LS * (SD_MAX_L + 1) + t2q];
      const V *__restrict__ pbq = (halo && qb >= dm.n_local) ? halo + (qb - dm.n_local) : psi + qb;
      const __amdgpu_buffer_rsrc_t rq = make_rsrc(pbq, (uint32_t)lenq * ES);
      // (the partner tile belongs to the suffix sector t' +- 1: it may be longer than the BLOCK * R rows this workgroup covers)
      for (int c0 = 0; c0 < lenq; c0 += BLOCK * R) {
#pragma unroll
        for (int r = 0; r < R; ++r) buf_load(va[r], rq, off0 + (uint32_t)(c0 + r * 64) * ES);
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (c0 + i0 + r * 64 < lenq) tile2[1 + c0 + i0 + r * 64] = va[r];
      }
      if (tid == 0) tile2[0] = V{};
    }
  }
  SD_STAMP(4);

  // ---- general bonds (anything after the leading chain bonds: the periodic (L,1) bond, long-range lists) ----
  // Where the partner row lives follows from where the two sites are (wave-uniform per bond):
  //   both in the prefix : the whole tile maps onto tile P ^ bits at the same row offset (a coalesced stream, like a chain bond);
  //   both in the suffix : the partner row is in this tile -- an LDS read at suf_rank[sigma ^ bits];
  //   one in each        : every partner row is in the ONE tile P ^ prefix bit (its base is fetched once per wave), at row
  //                        suf_rank[sigma ^ suffix bit]; the suffix sector changes with the prefix filling.
  // The value for rows without the hop is never used.
  struct GBond { int64_t base; uint32_t smask; int pb; int kind; };     // kind 0 prefix-prefix, 1 suffix-suffix, 2 mixed, -1 nothing to do
  auto gbond = [&](int h) {
    GBond g{0, 0u, 0, -1};
    const int bi = dm.hop_i[h] - 1, bj = dm.hop_j[h] - 1;
    const bool ip = bi < p, jp = bj < p;
    if (ip && jp) {
      if (((P >> bi) ^ (P >> bj)) & 1u) { g.kind = 0; g.base = dm.addr[P ^ (1u << bi) ^ (1u << bj)]; }
    } else if (!ip && !jp) {
      g.kind = 1; g.smask = (1u << (bi - p)) | (1u << (bj - p));
    } else {
      g.pb = ip ? bi : bj;
      g.smask = 1u << ((ip ? bj : bi) - p);
      const uint32_t Q = P ^ (1u << g.pb);
      const int t2q = dm.nup - __popc(Q);
      if (t2q >= 0 && t2q <= LS) { g.kind = 2; g.base = dm.addr[Q]; }
    }
    return g;
  };
  auto gflip = [&](const GBond &g, int r) -> bool {       // does row r have this hop?
    if (g.kind == 0) return true;
    if (g.kind == 1) return __popc(sig[r] & g.smask) == 1;
    if (g.kind == 2) return ((P >> g.pb) & 1u) != ((sig[r] & g.smask) ? 1u : 0u);
    return false;
  };
  auto gvalue = [&](const GBond &g, int r) -> V {         // psi at the partner row of row r (call only when gflip)
    if (g.kind == 1) return tile[1 + dm.suf_rank[sig[r] ^ g.smask]];
    const int64_t idx = g.base + (g.kind == 0 ? (int64_t)(i0 + r * 64) : (int64_t)dm.suf_rank[sig[r] ^ g.smask]);
    return (halo && idx >= dm.n_local) ? halo[idx - dm.n_local] : psi[idx];
  };
  // the first general bond's values are requested now, into the idle stream registers, so that their latency hides behind
  // the suffix phase; they are accumulated in their turn, after the suffix bonds
  const bool have_g = !GEN && nn < dm.n_hop;
  double gJ = 0.0;
  GBond g0{0, 0u, 0, -1};
  auto request_g0 = [&]() {
    g0 = gbond(nn);
    gJ = dm.hop_J[nn];
#pragma unroll
    for (int r = 0; r < R; ++r) va[r] = (i0 + r * 64 < len && gflip(g0, r)) ? gvalue(g0, r) : V{};
  };
  // (PK: requested after the suffix phase instead -- values held across it would sit on top of the packed table's registers
  // and cost every model, with or without general bonds, a wave per SIMD)
  if (have_g && !PK) request_g0();
  // ---- 4. bonds inside the suffix ----
  if (nn > 0 && !(DIAG && (dm.dbg & 4))) {
    if constexpr (PK) {
      // partner row + 1 from the packed table (0: no hop -> the zero row); branch-free, the R rows' LDS reads overlap
      const unsigned char *tb = reinterpret_cast<const unsigned char *>(tile);
      constexpr uint32_t SH = NC == 2 ? 4 : 3;            // log2(sizeof(V))
#pragma unroll
      for (int a = 1; a <= 11; ++a) if (a <= LS - 1) {     // wave-uniform guard (LS is a run-time value <= 12)
        const double J = dm.hop_J[p + a - 1];
        const int w = (a - 1) / 3, sh = 10 * ((a - 1) % 3);
        V v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const u4 q = pt[PK ? r : 0];
          const uint32_t word = w == 0 ? q.x : w == 1 ? q.y : w == 2 ? q.z : q.w;
          const uint32_t ad = ((word >> sh) & 0x3FFu) << SH;
          v[r] = *reinterpret_cast<const V *>(tb + ad);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = accum<FMA>(acc[r], J, v[r]);
      }
    } else {
      // LS > 12: LDS reads at idx +- C(LS-a-1, u), u = ups beyond the bond (binomials in LDS)
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
          const int ip = (i0 + r * 64) + (up ? d[r] : -d[r]);
          v[r] = tile[fl ? ip + 1 : 0];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = accum<FMA>(acc[r], J, v[r]);
      }
    }
  }
  // ---- 4b. the general bonds in list order, from the plan (sd_gbond) ----
  // prefix-prefix bonds: whole-tile streams; the next flippable one is requested as soon as the current one has been accumulated,
  // whatever lies between them in the list -- requesting early does not change the order of the sums.  (One register set: with two
  // in flight the form needs 113 registers, four waves, and is 6-10 % slower: profiles/ablation_r04.md section 10.)  Suffix-suffix
  // bonds: LDS reads at the row the packed table gen_ss_part names (one 4-byte load per row and three bonds); mixed bonds: the
  // ONE partner tile streamed into the second LDS image and read at the row mix_part names.  Rows without the hop add J * 0.
  if constexpr (GEN) {
    constexpr uint32_t SH = NC == 2 ? 4 : 3;            // log2(sizeof(V))
    const unsigned char *tb = reinterpret_cast<const unsigned char *>(tile);
    const unsigned char *tb2 = reinterpret_cast<const unsigned char *>(tile2);
    uint32_t ptw[R];                                     // the current word of the packed table: three suffix-suffix bonds
    int cur_word = -1;
    auto gissue = [&](int64_t gbase, V(&v)[R]) {
      const V *__restrict__ pbp = (halo && gbase >= dm.n_local) ? halo + (gbase - dm.n_local) : psi + gbase;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(pbp, (uint32_t)len * ES);      // same prefix filling: same length, same row offsets
#pragma unroll
      for (int r = 0; r < R; ++r) buf_load(v[r], rs, off0 + (uint32_t)(r * 64) * ES);
    };
    for (int hb = 0; hb < dm.n_gen; hb += 64) {
      const int nb = dm.n_gen - hb < 64 ? dm.n_gen - hb : 64;
      // lane k <-> bond hb + k: the partner bases of the flippable prefix-prefix bonds of this block, one gather per wave
      // (the descriptors stay in the lanes and are read with v_readlane per bond: a scalar load per bond would put a memory
      // latency at the head of every one of them)
      int64_t base_k = -1;
      sd_gbond gl{-1, 0u, 0, 0, 0.0};
      if (lane < nb) {
        gl = dm.gen[hb + lane];
        if (gl.kind == 0 && __popc(P & gl.pmask) == 1) base_k = dm.addr[P ^ gl.pmask];
      }
      uint64_t todo = __ballot(base_k >= 0);
      int in_a = -1;                                       // the bond whose rows are in flight in va
#if SD_GEN_DEPTH == 2
      int in_b = -1;                                       // ... and in vb (consumed in this order)
#endif
      if (todo) { in_a = next_lane(todo); gissue(rl64(base_k, in_a), va); }
#if SD_GEN_DEPTH == 2
      if (todo) { in_b = next_lane(todo); gissue(rl64(base_k, in_b), vb); }
#endif
#pragma nounroll
      for (int k = 0; k < nb; ++k) {
        sd_gbond gb;                                      // wave-uniform
        gb.kind = rl(gl.kind, k); gb.pmask = (uint32_t)rl((int)gl.pmask, k); gb.slot = rl(gl.slot, k); gb.pb = rl(gl.pb, k);
        gb.J = rld(gl.J, k);
        if (gb.kind == 0) {
          if (k == in_a) {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = accum<false>(acc[r], gb.J, va[r]);
            in_a = -1;
            if (todo) { in_a = next_lane(todo); gissue(rl64(base_k, in_a), va); }
          }
#if SD_GEN_DEPTH == 2
          else if (k == in_b) {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = accum<false>(acc[r], gb.J, vb[r]);
            in_b = -1;
            if (todo) { in_b = next_lane(todo); gissue(rl64(base_k, in_b), vb); }
          }
#endif
        } else if (gb.kind == 1) {
          // (one 4-byte word = three bonds per load: a whole 16-byte entry per row, as the chain bonds keep, would hold 16
          // registers across the streams and cost this form a wave per SIMD)
          const int wi = gb.slot / 3, sh = 10 * (gb.slot - 3 * wi);
          if (wi != cur_word) {
            cur_word = wi;
            const int chunk = wi >> 2;
            const __amdgpu_buffer_rsrc_t rp2 = make_rsrc(dm.gen_ss_part + 4 * ((size_t)chunk * (size_t)dm.n_suf_rows + (size_t)rec.suf_off), (uint32_t)len * 16u);
            const uint32_t wo = (uint32_t)(wi & 3) * 4u;
#pragma unroll
            for (int r = 0; r < R; ++r) ptw[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rp2, (uint32_t)i0 * 16u + (uint32_t)(r * 64) * 16u + wo, 0, 0);
          }
          V v[R];
#pragma unroll
          for (int r = 0; r < R; ++r) v[r] = *reinterpret_cast<const V *>(tb + ((((ptw[r] >> sh) & 0x3FFu)) << SH));
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r] = accum<false>(acc[r], gb.J, v[r]);
        } else if (gb.kind == 2) {
          const uint32_t Q = P ^ gb.pmask;
          const int t2q = dm.nup - __popc(Q);
          const bool ok = t2q >= 0 && t2q <= LS;
          __syncthreads();                                 // the previous mixed bond's readers of the second image are done
          if (ok) {
            const int64_t qb = dm.addr[Q];
            const int lenq = (int)dm.binom[LS * (SD_MAX_L + 1) + t2q];
            const V *__restrict__ pbq = (halo && qb >= dm.n_local) ? halo + (qb - dm.n_local) : psi + qb;
            const __amdgpu_buffer_rsrc_t rq = make_rsrc(pbq, (uint32_t)lenq * ES);
            for (int c0 = 0; c0 < lenq; c0 += BLOCK * R) {   // (the partner tile's sector is t' +- 1: it may be longer than BLOCK * R rows)
              V tmp[R];
#pragma unroll
              for (int r = 0; r < R; ++r) buf_load(tmp[r], rq, off0 + (uint32_t)(c0 + r * 64) * ES);
#pragma unroll
              for (int r = 0; r < R; ++r)
                if (c0 + i0 + r * 64 < lenq) tile2[1 + c0 + i0 + r * 64] = tmp[r];
            }
            if (tid == 0) tile2[0] = V{};
          }
          __syncthreads();
          if (ok) {
            const __amdgpu_buffer_rsrc_t rm = make_rsrc(dm.mix_part + ((size_t)gb.slot * (size_t)dm.n_suf_rows + (size_t)rec.suf_off), (uint32_t)len * 2u);
            const uint32_t bpb = (P >> gb.pb) & 1u;
            uint32_t w16[R];
#pragma unroll
            for (int r = 0; r < R; ++r) w16[r] = buf_load_u16(rm, (uint32_t)i0 * 2u + (uint32_t)(r * 64) * 2u);
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const bool has = (w16[r] >> 15) != bpb;       // the row's suffix site differs from the tile's prefix site
              const uint32_t ad = has ? ((w16[r] & 0x3FFu) << SH) : 0u;
              acc[r] = accum<false>(acc[r], gb.J, *reinterpret_cast<const V *>(tb2 + ad));
            }
          }
        }
      }
    }
  }
  // ---- the general bonds in list order (without a plan) ----
  if (wrap) {
    __syncthreads();                                     // the second image is complete (every wave takes this branch or none)
    if (wrap_on) {
      const double Jw = dm.hop_J[nn];
      const uint32_t bpb = (P >> dm.wrap_pb) & 1u;
      const unsigned char *tb2 = reinterpret_cast<const unsigned char *>(tile2);
      constexpr uint32_t SH = NC == 2 ? 4 : 3;
#pragma unroll
      for (int r = 0; r < R; ++r) {                      // (row by row: a batch of R values would cost the Float64 form a register spill)
        const uint32_t w3 = pt[PK ? r : 0].w;
        const bool has = ((w3 >> 30) & 1u) != bpb;       // the row's suffix site differs from the tile's prefix site
        const uint32_t ad = has ? (((w3 >> 20) & 0x3FFu) << SH) : 0u;
        acc[r] = accum<false>(acc[r], Jw, *reinterpret_cast<const V *>(tb2 + ad));     // (rows without the hop add J * 0)
      }
    }
  }
  if (have_g) {
    if (!wrap) {
      if (PK) request_g0();
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (gflip(g0, r)) acc[r] = accum<false>(acc[r], gJ, va[r]);   // rows without the hop keep acc untouched
    }
    for (int h = nn + 1; h < dm.n_hop; ++h) {
      const GBond g = gbond(h);
      if (g.kind < 0) continue;
      const double J = dm.hop_J[h];
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (i0 + r * 64 < len && gflip(g, r)) acc[r] = accum<false>(acc[r], J, gvalue(g, r));
    }
  }

  SD_STAMP(5);
  // ---- 5. epilogue + store ----
  EpiSums sums{0.0, 0.0};
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = i0 + r * 64;
    if (i < len) epilogue<NC>(epi, ea, base + i, acc[r], tile[1 + i], out_, sums);
  }
  SD_STAMP(6);
  if (DIAG && stamp && tid == 0) stamp[7] = __builtin_amdgcn_s_memrealtime();
  if (epi_has_sums(epi)) {
    double a = sums.s0, b = sums.s1;
    block_reduce2(a, b, red);
    // one pair per tile of the plan (interior and boundary launches fill disjoint slots of one buffer sized for all tiles)
    if (tid == 0) { partials[2 * (size_t)tix] = a; partials[2 * (size_t)tix + 1] = b; }
  }
}

// =====================================================================
// full-basis tiled kernel (nup = nothing: idx = state, src/Hamiltonian.jl:223,255-257)
// =====================================================================
//
// Site i is bit i-1 of the row index, so 2^LF consecutive rows form a tile without any table: sites 1..LF vary inside
// the tile, sites LF+1..L are the tile number T.  A hop on the chain bond (a, a+1) flips two index bits:
//   a+1 <= LF : partner row i ^ (3 << (a-1)) of the same tile                              -> LDS
//   a   == LF : the half of the tile whose bit LF-1 differs from T's bit 0 reads the other half of tile T ^ 1
//   a   >  LF : whole tile T ^ (3 << (a-LF-1)) at the same in-tile offset (when T's two bits differ)
// Accumulation per row is the reference's bond order 1..L-1: the LDS bonds first, then the streams (requested before the
// LDS phase, consumed after it), then any further bonds of the hop list as gathers.
template <int NC, bool FMA>
__global__ __launch_bounds__(256, 4) void k_apply_fulltile(sd_dev_model dm, double *__restrict__ out_,
                                                        const double *__restrict__ psi_, int epi, sd_epi_args ea,
                                                        double *__restrict__ partials) {
  using V = typename VT<NC>::type;
  constexpr uint32_t ES = sizeof(V);
  constexpr int R = 4, BLOCK = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int LF = dm.full_ls, TL = 1 << LF;            // TL == R * BLOCK
  V *tile = reinterpret_cast<V *>(smem);               // TL rows + one all-zero row at index TL
  double *red = reinterpret_cast<double *>(smem + (size_t)(TL + 1) * sizeof(V));
  const V *__restrict__ psi = reinterpret_cast<const V *>(psi_);
  const V *__restrict__ halo = reinterpret_cast<const V *>(ea.halo);
  const int tid = threadIdx.x, lane = tid & 63;
  // blocks are dealt round-robin to the 8 XCDs: hand each XCD runs of 32 consecutive tiles (neighbouring tiles are partners)
  uint32_t T = blockIdx.x;
  if (gridDim.x >= 512 && (gridDim.x & 255u) == 0) {
    const uint32_t x = T & 7u, j = T >> 3, g = j >> 5, i = j & 31u;
    T = (g << 8) + (x << 5) + i;
  }
  const int64_t base = (int64_t)T << LF;                       // local row of the tile's first row
  // sharded by the top index bits: this rank's tiles are the global tiles T + row_lo / 2^LF (the rank is their top bits)
  const uint32_t Tg = T + (uint32_t)(dm.row_lo >> LF);
  const int64_t gbase = (int64_t)Tg << LF;                     // its state (= global row)
  const int nn = dm.nn_hops;

  V own[R];
  uint32_t ioff[R];
  {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(psi + base, (uint32_t)TL * ES);
#pragma unroll
    for (int r = 0; r < R; ++r) { ioff[r] = (uint32_t)(tid + r * BLOCK) * ES; buf_load(own[r], rs, ioff[r]); }
  }
  // per-wave list of flippable stream bonds: lane 0 <-> bond LF (straddle), lane k <-> bond LF + k
  uint64_t fmask = 0;
  int64_t my_base = 0;
  double my_J = 0.0;
  int my_lo = 0, my_n = TL;
  if (nn > 0) {
    const int a = LF + lane;
    bool fl = false;
    if (a <= dm.L - 1) {
      if (lane == 0) {
        const int t = (int)(T & 1u);
        my_lo = (1 - t) * (TL / 2); my_n = TL / 2;
        my_base = ((int64_t)(T ^ 1u) << LF) + (TL / 2 - my_lo);
        fl = true;
      } else {
        const int b = a - LF - 1;
        fl = ((Tg >> b) ^ (Tg >> (b + 1))) & 1u;
        // partner tile: in this rank's rows, or (a bond that reaches into the rank bits) in the slab imported from rank q
        const uint32_t Tp = Tg ^ (3u << b);
        const int tb = dm.L - LF - dm.fs_dbits;                  // tile-index bits below the rank bits
        const uint32_t q = Tp >> tb;
        const int64_t at_peer = (int64_t)(Tp & ((1u << tb) - 1u)) << LF;
        my_base = at_peer;
        if (q != (Tg >> tb)) {
          int64_t ho = 0, plo = 0;
#pragma unroll
          for (int q2 = 0; q2 < SD_FS_MAX_RANKS; ++q2)
            if ((uint32_t)q2 == q) { ho = dm.fs_halo_off[q2]; plo = dm.fs_peer_lo[q2]; }
          my_base = dm.n_local + ho + at_peer - plo;
        }
      }
      my_J = dm.hop_J[a - 1];
    }
    fmask = __ballot(fl);
  }
  auto get_bond = [&](int ln) {
    FarBond fb;
    fb.base = rl64(my_base, ln); fb.J = rld(my_J, ln);
    fb.lo = rl(my_lo, ln); fb.n = rl(my_n, ln);
    return fb;
  };
  auto issue = [&](const FarBond &fb, V(&v)[R]) {
    const V *__restrict__ pb = (halo && fb.base >= dm.n_local) ? halo + (fb.base - dm.n_local) : psi + fb.base;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(pb, (uint32_t)fb.n * ES);
    const uint32_t lo_b = (uint32_t)fb.lo * ES;
#pragma unroll
    for (int r = 0; r < R; ++r) buf_load(v[r], rs, ioff[r] - lo_b);
  };
  V va[R], vb[R];
  FarBond fa{}, fbb{};
  uint64_t mk = fmask;
  bool have_a = false;
  auto next_lane = [&](uint64_t &m_) { const int ln = __builtin_ctzll(m_); m_ &= m_ - 1; return ln; };
  if (mk) { fa = get_bond(next_lane(mk)); issue(fa, va); have_a = true; }

  V acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    acc[r] = vscale(diag_of(dm, (uint64_t)gbase + (uint64_t)(tid + r * BLOCK)), own[r]);
    tile[tid + r * BLOCK] = own[r];
  }
  if (tid == 0) tile[TL] = V{};
  __syncthreads();

  // bonds inside the tile (LDS)
  const int n_in = nn > 0 ? LF - 1 : 0;
  for (int a = 1; a <= n_in; ++a) {
    const double J = dm.hop_J[a - 1];
    V v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = tid + r * BLOCK;
      const bool fl = ((i >> (a - 1)) ^ (i >> a)) & 1;
      v[r] = tile[fl ? (i ^ (3 << (a - 1))) : TL];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = accum<FMA>(acc[r], J, v[r]);
  }
  // streams, ascending bond order
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
  // remaining (general) bonds: idx' = idx ^ (two bits)
  for (int h = nn; h < dm.n_hop; ++h) {
    const int bi = dm.hop_i[h] - 1, bj = dm.hop_j[h] - 1;
    const double J = dm.hop_J[h];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t s = (uint64_t)gbase + (uint64_t)(tid + r * BLOCK);       // (sharded plans carry chain bonds only)
      if (((s >> bi) ^ (s >> bj)) & 1) acc[r] = accum<false>(acc[r], J, psi[s ^ ((uint64_t)1 << bi) ^ ((uint64_t)1 << bj)]);
    }
  }
  EpiSums sums{0.0, 0.0};
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = tid + r * BLOCK;
    epilogue<NC>(epi, ea, base + i, acc[r], tile[i], out_, sums);
  }
  if (epi_has_sums(epi)) {
    double a = sums.s0, b = sums.s1;
    block_reduce2(a, b, red);
    if (tid == 0) { partials[2 * (size_t)blockIdx.x] = a; partials[2 * (size_t)blockIdx.x + 1] = b; }
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
  // the binomials of the combinadic order in LDS (a sector's unrank / rank walk L of them per call), 64 x 64 entries
  __shared__ int64_t lb[64 * 64];
  const V *__restrict__ psi = reinterpret_cast<const V *>(psi_);
  const bool full = dm.nup < 0;
  const int L = dm.L;
  if (!full) {
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
      const int n = i >> 6, k = i & 63;
      lb[i] = (n <= L && k <= n) ? dm.binom[n * (SD_MAX_L + 1) + k] : 0;
    }
    __syncthreads();
  }
  auto bl = [&](int n, int k) -> int64_t { return (k < 0 || k > n) ? 0 : lb[(n << 6) + k]; };
  auto unrank_l = [&](int64_t idx) {
    uint64_t s = 0;
    int r = dm.nup;
    for (int k = 1; k <= L && r > 0; ++k) {
      const int64_t c = bl(L - k, r - 1);
      if (idx < c) { s |= (uint64_t)1 << (k - 1); --r; }
      else idx -= c;
    }
    return s;
  };
  auto rank_l = [&](uint64_t s) {
    int64_t idx = 0;
    int r = dm.nup;
    for (int k = 1; k <= L && r > 0; ++k) {
      if ((s >> (k - 1)) & 1) --r;
      else idx += bl(L - k, r - 1);
    }
    return idx;
  };
  const int nn = full ? 0 : dm.nn_hops;       // leading chain bonds (1,2),(2,3),...: the partner index in closed form
  EpiSums sums{0.0, 0.0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < dm.N; idx += stride) {
    const uint64_t s = full ? (uint64_t)idx : unrank_l(idx);
    const V own = psi[idx];
    V acc = vscale(diag_of(dm, s), own);
    // chain bond (b, b+1): the flip moves one up spin across one site, past nothing -- the index changes by C(L-b-1, u),
    // u = ups beyond site b+1, upwards when site b holds the up spin; no rank walk
    for (int b = 1; b <= nn; ++b)
      if (((s >> (b - 1)) ^ (s >> b)) & 1) {
        const int64_t d = lb[((L - b - 1) << 6) + __popcll(s >> (b + 1))];
        const int64_t nidx = ((s >> (b - 1)) & 1) ? idx + d : idx - d;
        acc = vadd_mul(acc, dm.hop_J[b - 1], psi[nidx]);
      }
    for (int h = nn; h < dm.n_hop; ++h) {
      const int bi = dm.hop_i[h] - 1, bj = dm.hop_j[h] - 1;
      if (((s >> bi) ^ (s >> bj)) & 1) {
        const uint64_t s2 = s ^ ((uint64_t)1 << bi) ^ ((uint64_t)1 << bj);
        const int64_t nidx = full ? (int64_t)s2 : rank_l(s2);
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

// =====================================================================
// short tiles: sixteen lanes per tile, one row per lane
// =====================================================================
// The tiles of suffix fillings 0, 1, LS-1 and LS hold 1 or LS <= 12 rows; in a dilute sector they are most of the tiles (L=36,
// nup=9: 2.0 M of 2.6 M) and a workgroup each is nearly all set-up.  Here a 256-thread block takes sixteen of them: lane i of a
// 16-lane group owns row i of the group's tile -- configuration from the tile record and the suffix table, diagonal by diag_of (or
// the cached value), a chain bond's partner row by the closed form of the combinadic order, idx +- C(L-b-1, u) (SURVEY appendix B;
// unsharded plans: local row = global row), other bonds by a rank walk over binomials held in LDS.  Same per-row arithmetic and
// order as everywhere else; the tile's two partial sums are reduced over its sixteen lanes and filed under the tile's index.
template <int NC, int LPT>      // LPT lanes per tile: 16 (tiles of 2..16 rows) or 1 (one-row tiles)
__global__ __launch_bounds__(256) void k_apply_short(sd_dev_model dm, double *__restrict__ out_, const double *__restrict__ psi_, int epi,
                                                     sd_epi_args ea, double *__restrict__ partials, int first, int count) {
  using V = typename VT<NC>::type;
  __shared__ int64_t lb[64 * 64];
  if (ea.batch > 1) {                                          // vector blockIdx.y of a batch (as in k_apply_tiled)
    const int64_t boff = (int64_t)blockIdx.y * ea.bstride * NC;
    psi_ += boff; out_ += boff;
    if (ea.prev) ea.prev = (const double *)ea.prev + boff;
    if (ea.phi) ea.phi = (const double *)ea.phi + boff;
    if (ea.accv) ea.accv = (double *)ea.accv + boff;
    partials += 2 * (size_t)blockIdx.y * (size_t)dm.n_singles;
    ea.negate = (ea.negate >> blockIdx.y) & 1;
  }
  const V *__restrict__ psi = reinterpret_cast<const V *>(psi_);
  const int L = dm.L, p = dm.p;
  for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {    // (paid once per block: the blocks stride over the tile list)
    const int n = i >> 6, k = i & 63;
    lb[i] = (n <= L && k <= n) ? dm.binom[n * (SD_MAX_L + 1) + k] : 0;
  }
  __syncthreads();
  auto bl = [&](int n, int k) -> int64_t { return (k < 0 || k > n) ? 0 : lb[(n << 6) + k]; };
  auto rank_l = [&](uint64_t s) {
    int64_t idx = 0;
    int r = dm.nup;
    for (int k = 1; k <= L && r > 0; ++k) {
      if ((s >> (k - 1)) & 1) --r;
      else idx += bl(L - k, r - 1);
    }
    return idx;
  };
  constexpr int TPB = 256 / LPT;                               // tiles of a block per sweep
  const int i = threadIdx.x % LPT;
  const int nn = dm.nn_hops;
  const int sweeps = (count + (int)gridDim.x * TPB - 1) / ((int)gridDim.x * TPB);      // the same for every lane: the shuffles below stay converged
  for (int sw = 0; sw < sweeps; ++sw) {
    const int g = (sw * (int)gridDim.x + (int)blockIdx.x) * TPB + (int)threadIdx.x / LPT;
    const bool have_tile = g < count;
    const int tix = first + (have_tile ? g : 0);
    const sd_tile_rec rec = dm.single_rec[tix];
    EpiSums sums{0.0, 0.0};
    if (have_tile && i < rec.len) {
      const int64_t idx = rec.base + i;
      const uint64_t s = (uint64_t)rec.prefix | ((uint64_t)dm.suf_states[rec.suf_off + i] << p);
      const V own = psi[idx];
      V acc = vscale(dm.diag_cache ? dm.diag_cache[idx] : diag_of(dm, s), own);
      for (int b = 1; b <= nn; ++b)
        if (((s >> (b - 1)) ^ (s >> b)) & 1) {
          const int64_t d = lb[((L - b - 1) << 6) + __popcll(s >> (b + 1))];
          const int64_t nidx = ((s >> (b - 1)) & 1) ? idx + d : idx - d;
          acc = vadd_mul(acc, dm.hop_J[b - 1], psi[nidx]);
        }
      for (int h = nn; h < dm.n_hop; ++h) {
        const int bi = dm.hop_i[h] - 1, bj = dm.hop_j[h] - 1;
        if (((s >> bi) ^ (s >> bj)) & 1) {
          const uint64_t s2 = s ^ ((uint64_t)1 << bi) ^ ((uint64_t)1 << bj);
          acc = vadd_mul(acc, dm.hop_J[h], psi[rank_l(s2)]);
        }
      }
      epilogue<NC>(epi, ea, idx, acc, own, out_, sums);
    }
    if (epi_has_sums(epi)) {
      double a = sums.s0, b = sums.s1;
      if (LPT > 1)
        for (int off = LPT / 2; off > 0; off >>= 1) {          // over the tile's lanes, fixed order
          a += __shfl_down(a, off, LPT);
          b += __shfl_down(b, off, LPT);
        }
      if (have_tile && i == 0) { partials[2 * (size_t)tix] = a; partials[2 * (size_t)tix + 1] = b; }
    }
  }
}

// epilogue alone, for H psi produced by a caller's operator (sd_ctx_set_apply_callback): same arithmetic per element as the
// fused form; out may be hpsi
template <int NC>
__global__ __launch_bounds__(256) void k_epilogue_only(int epi, sd_epi_args ea, int64_t n, double *out_,
                                                       const double *hpsi_, const double *__restrict__ psi_,
                                                       double *__restrict__ partials) {
  using V = typename VT<NC>::type;
  __shared__ double red[32];
  const V *hpsi = reinterpret_cast<const V *>(hpsi_);
  const V *__restrict__ psi = reinterpret_cast<const V *>(psi_);
  EpiSums sums{0.0, 0.0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const V acc = hpsi[i], own = psi[i];
    epilogue<NC>(epi, ea, i, acc, own, out_, sums);
  }
  if (epi_has_sums(epi)) {
    double a = sums.s0, b = sums.s1;
    block_reduce2(a, b, red);
    if (threadIdx.x == 0) { partials[2 * (size_t)blockIdx.x] = a; partials[2 * (size_t)blockIdx.x + 1] = b; }
  }
}

// fixed-order reduction of the per-block partial pairs -> scalars[0..1]; block k of a batched call reduces list k
__global__ __launch_bounds__(1024) void k_reduce_pairs(const double *__restrict__ partials, int64_t n,
                                                       double *__restrict__ scalars, int64_t dstride) {
  __shared__ double red[32];
  partials += 2 * (size_t)blockIdx.x * (size_t)n;
  scalars += (size_t)blockIdx.x * (size_t)dstride;
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) { a += partials[2 * i]; b += partials[2 * i + 1]; }
  block_reduce2(a, b, red);
  if (threadIdx.x == 0) { scalars[0] = a; scalars[1] = b; }
}

// first stage for long lists (one partial pair per tile: 10^6 pairs at L=32, too long for one workgroup): block j sums
// the pairs [j*chunk, (j+1)*chunk) into stage[j]; still a fixed order for a given n
__global__ __launch_bounds__(256) void k_reduce_pairs_stage(const double *__restrict__ partials, int64_t n, int64_t chunk,
                                                            double *__restrict__ stage) {
  __shared__ double red[32];
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  const double2 *__restrict__ p2 = reinterpret_cast<const double2 *>(partials);
  double a = 0.0, b = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) { const double2 v = p2[i]; a += v.x; b += v.y; }
  block_reduce2(a, b, red);
  if (threadIdx.x == 0) { stage[2 * blockIdx.x] = a; stage[2 * blockIdx.x + 1] = b; }
}

// partials[0 .. 2n) -> ctx->d_scalars[0..1]; the caller reserved 2n + 2*SD_RED_STAGE_BLOCKS doubles of ctx->d_partials
}  // namespace
int sd_reduce_pairs(sd_ctx *ctx, int64_t n, double *dst) {
  if (!dst) dst = ctx->d_scalars;
  const double *src = ctx->d_partials;
  if (n > 16384) {
    double *stage = ctx->d_partials + 2 * n;
    const int64_t chunk = (n + SD_RED_STAGE_BLOCKS - 1) / SD_RED_STAGE_BLOCKS;
    hipLaunchKernelGGL(k_reduce_pairs_stage, dim3(SD_RED_STAGE_BLOCKS), dim3(256), 0, ctx->stream, src, n, chunk, stage);
    SD_HIP(ctx, hipGetLastError());
    src = stage; n = SD_RED_STAGE_BLOCKS;
  }
  hipLaunchKernelGGL(k_reduce_pairs, dim3(1), dim3(1024), 0, ctx->stream, src, n, dst, (int64_t)0);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
int sd_reduce_pairs_batched(sd_ctx *ctx, int64_t n, int batch, double *dst, int64_t dstride) {
  if (!dst || n > 16384 || batch < 1) return sd_set_err(ctx, SD_EINTERNAL, "bad batched reduction");
  hipLaunchKernelGGL(k_reduce_pairs, dim3((unsigned)batch), dim3(1024), 0, ctx->stream, ctx->d_partials, n, dst, dstride);   // same geometry as one list alone: same bits
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
int sd_launch_epilogue_only(sd_ctx *ctx, int dtype, int64_t n, void *out, const void *hpsi, const void *psi, int epi,
                            const sd_epi_args &ea_in) {
  const bool sums = (epi == SD_EPI_DOT || epi == SD_EPI_KPM || epi == SD_EPI_RESCALE_DOT);
  sd_epi_args ea = ea_in;
  ea.stream_hint = 0;
  if (n <= 0) {
    if (sums) SD_HIP(ctx, hipMemsetAsync(ea.sums_dst ? ea.sums_dst : ctx->d_scalars, 0, 2 * sizeof(double), ctx->stream));
    return SD_OK;
  }
  const int64_t nb = std::min<int64_t>((n + 255) / 256, 4096);
  if (sums) { int rc = sd_ensure_partials(ctx, 2 * (size_t)nb + 2 * SD_RED_STAGE_BLOCKS); if (rc) return rc; }
  if (dtype == SD_C128)
    hipLaunchKernelGGL(k_epilogue_only<2>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, epi, ea, n, (double *)out,
                       (const double *)hpsi, (const double *)psi, ctx->d_partials);
  else
    hipLaunchKernelGGL(k_epilogue_only<1>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, epi, ea, n, (double *)out,
                       (const double *)hpsi, (const double *)psi, ctx->d_partials);
  SD_HIP(ctx, hipGetLastError());
  if (sums) return sd_reduce_pairs(ctx, nb, ea.sums_dst);
  return SD_OK;
}
namespace {

template <int NC, int R, int BLOCK, bool FMA, bool PK, bool GEN = false>
int launch_tiled_cfg(sd_ctx *ctx, const sd_dev_model &dm, int nt, size_t shmem, double *out, const double *psi, int epi,
                     const sd_epi_args &ea, int max_len) {
  // the stamped (DIAG) instantiation exists for one configuration only and is reached through sd_debug_phase_profile
  void (*kern)(sd_dev_model, double *, const double *, int, sd_epi_args, double *, int) = k_apply_tiled<NC, R, BLOCK, FMA, PK, false, GEN>;
  // ... and for the SD_DEBUG_SKIP timing ablations: the production instantiations carry no run-time debug branches
  if constexpr (NC == 2 && FMA && PK && !GEN && (BLOCK == 256 || BLOCK == 128 || BLOCK == 64))
    if (dm.stamps || dm.dbg) kern = k_apply_tiled<NC, R, BLOCK, FMA, PK, true>;
  // per kernel AND per device, so no cache: cheap next to a launch, and only the SD_SUFFIX_BITS >= 13 tiles get here
  if (shmem > 48 * 1024)
    SD_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kern, dim3(nt, ea.batch > 1 ? ea.batch : 1), dim3(BLOCK), shmem, ctx->stream, dm, out, psi, epi, ea, ctx->d_partials, max_len);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

template <int NC, bool FMA>
int launch_tiled(sd_ctx *ctx, const sd_dev_model &dm, int nt, int cls, size_t shmem, double *out, const double *psi, int epi,
                 const sd_epi_args &ea, int max_len) {
  // 4 rows per thread; cls selects the workgroup size (64 << cls threads) that covers the segment's longest tile
  constexpr int R = 4;
  if ((64 << cls) * R < max_len) return sd_set_err(ctx, SD_EINTERNAL, "tile longer than its workgroup can hold");
  if (!dm.suf_part) {
    // LS > 12 (SD_SUFFIX_BITS experiments): partners from binomials, the two largest workgroups only (basis.cpp puts every
    // tile of such a plan into one class >= 3)
    switch (cls) {
      case 3: return launch_tiled_cfg<NC, R, 512, FMA, false>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
      case 4: return launch_tiled_cfg<NC, R, 1024, FMA, false>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
    }
    return sd_set_err(ctx, SD_EINTERNAL, "bad tile length class for a plan without the packed partner table");
  }
  if (dm.n_gen > 0)     // general bonds from the host-resolved plan (4 rows per thread for both element types)
    switch (cls) {
      case 0: return launch_tiled_cfg<NC, R, 64, FMA, true, true>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
      case 1: return launch_tiled_cfg<NC, R, 128, FMA, true, true>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
      case 2: return launch_tiled_cfg<NC, R, 256, FMA, true, true>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
      default: return sd_set_err(ctx, SD_EINTERNAL, "bad tile length class");
    }
  if constexpr (NC == 1) {
    // Float64: 8 rows per thread in workgroups of half the size (same registers as 4 ComplexF64 rows; the per-thread
    // set-up -- far-bond list, descriptors -- is paid once per 8 rows).  SD_F64_ROWS=4 keeps 4 rows per thread.
    static const int rows = getenv("SD_F64_ROWS") ? atoi(getenv("SD_F64_ROWS")) : 8;   // measured at L=30: 1.80 -> 1.74 ms
    if (rows == 8 && cls >= 1 && cls <= 2)
      switch (cls) {
        case 1: return launch_tiled_cfg<NC, 8, 64, FMA, true>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
        case 2: return launch_tiled_cfg<NC, 8, 128, FMA, true>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
      }
  }
  switch (cls) {      // LS <= 12: at most C(12,6) = 924 rows, i.e. class 2
    case 0: return launch_tiled_cfg<NC, R, 64, FMA, true>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
    case 1: return launch_tiled_cfg<NC, R, 128, FMA, true>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
    case 2: return launch_tiled_cfg<NC, R, 256, FMA, true>(ctx, dm, nt, shmem, out, psi, epi, ea, max_len);
  }
  return sd_set_err(ctx, SD_EINTERNAL, "bad tile length class");
}

}  // namespace

int sd_launch_apply(sd_ctx *ctx, const sd_model *m, int dtype, void *out, const void *psi, int epi,
                    const sd_epi_args &ea_in, int part) {
  if (!m->dev_ready) return sd_set_err(ctx, SD_EARG, "model has no device tables (created without a context)");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "dtype must be SD_F64 or SD_C128");
  const bool sums = (epi == SD_EPI_DOT || epi == SD_EPI_KPM || epi == SD_EPI_RESCALE_DOT);
  sd_dev_model dm = m->dm;
  sd_epi_args ea = ea_in;
  {
    static const int hint = getenv("SD_STREAM_HINT") ? atoi(getenv("SD_STREAM_HINT")) : 3;   // measured best: device_common.hpp, "Streams of the epilogues"
    ea.stream_hint = hint;
  }
  if (dm.n_local == 0) {      // a rank without rows still takes part in the reductions: its sums are zero
    if (sums) SD_HIP(ctx, hipMemsetAsync(ea.sums_dst ? ea.sums_dst : ctx->d_scalars, 0, 2 * sizeof(double), ctx->stream));
    return SD_OK;
  }
  if (ea.no_reduce && (m->p < 0 || part != 0)) return sd_set_err(ctx, SD_EINTERNAL, "unreduced sums: whole tiled plans only");
  if (ea.batch > 1 && (m->p < 0 || m->nranks != 1 || part != 0 || dm.n_singles > 16384 || (sums && !ea.sums_dst && !ea.no_reduce)))
    return sd_set_err(ctx, SD_EINTERNAL, "batched apply: unsharded tiled plans with at most 16384 tiles only");
  if (m->p >= 0) {
    // part 0: every tile; 1: interior tiles only (no halo read); 2: boundary tiles only.  A sum epilogue run in two parts
    // (1, then 2 with the same epilogue) files its per-tile partial sums into disjoint slots of one buffer and reduces
    // them once, after part 2: part 1 alone leaves the sums unreduced.
    int nt = dm.n_singles;
    if (part == 1) nt = dm.n_interior;
    else if (part == 2) nt = dm.n_singles - dm.n_interior;
    if (sums) { int rc = sd_ensure_partials(ctx, 2 * (size_t)dm.n_singles * (size_t)std::max(ea.batch, 1) + 2 * SD_RED_STAGE_BLOCKS); if (rc) return rc; }
    const int max_len = m->max_tile_len;
    const size_t esz = dtype == SD_C128 ? 16 : 8;
    int rc = SD_OK;
    if (nt > 0) {
      // one launch per non-empty (part, tile length class) segment; a segment's LDS image is sized by its own longest tile
      const int s0 = part == 2 ? SD_N_LEN_CLASS : 0, s1 = part == 1 ? SD_N_LEN_CLASS : 2 * SD_N_LEN_CLASS;
      for (int sg = s0; sg < s1 && rc == SD_OK; ++sg) {
        const int cnt = m->seg_off[sg + 1] - m->seg_off[sg];
        if (cnt <= 0) continue;
        const int cls = m->seg_cls[sg];
        const int seg_max = std::min(max_len, (64 << cls) * 4);
        size_t shmem = (size_t)(seg_max + 1) * esz + 16 * SD_BIN_STRIDE * sizeof(int) + 32 * sizeof(double) + 16;
        if ((dm.wrap_hop >= 0 && dm.wrap_hop == dm.nn_hops) || dm.n_gen_mixed > 0) shmem += (size_t)(m->max_tile_len_all + 1) * esz;   // the wrap bond's partner tile: any length class, owned or imported
        // (the 8-row Float64 form does not use it: launch_tiled; asking for it there would only cost occupancy -- but it is 7.4 KB)
        {   // experiment knob: SD_LDS_MIN_KB_<cls> raises the LDS request of a class, i.e. lowers its workgroups per CU
          static int min_kb[SD_N_LEN_CLASS] = {-1, -1, -1, -1, -1};
          if (min_kb[cls] < 0) {
            char name[32]; snprintf(name, sizeof(name), "SD_LDS_MIN_KB_%d", cls);
            const char *e = getenv(name);
            min_kb[cls] = e ? atoi(e) : 0;
          }
          if ((size_t)min_kb[cls] * 1024 > shmem) shmem = (size_t)min_kb[cls] * 1024;
        }
        dm.tile_off = m->seg_off[sg];
        if (dtype == SD_C128)
          rc = m->hop_pow2 ? launch_tiled<2, true>(ctx, dm, cnt, cls, shmem, (double *)out, (const double *)psi, epi, ea, seg_max)
                           : launch_tiled<2, false>(ctx, dm, cnt, cls, shmem, (double *)out, (const double *)psi, epi, ea, seg_max);
        else
          rc = m->hop_pow2 ? launch_tiled<1, true>(ctx, dm, cnt, cls, shmem, (double *)out, (const double *)psi, epi, ea, seg_max)
                           : launch_tiled<1, false>(ctx, dm, cnt, cls, shmem, (double *)out, (const double *)psi, epi, ea, seg_max);
      }
      if (rc) return rc;
      if (dm.n_short > 0 && part == 0) {          // the short tiles of an unsharded plan (k_apply_short)
        const unsigned by = ea.batch > 1 ? (unsigned)ea.batch : 1u;
        auto blocks = [](int tiles, int per_block) { return (unsigned)std::min(4096, std::max(1, (tiles + per_block - 1) / per_block)); };
        const int n16 = dm.n_short_multi, n1 = dm.n_short - dm.n_short_multi;
        if (n16 > 0) {
          if (dtype == SD_C128) hipLaunchKernelGGL((k_apply_short<2, 16>), dim3(blocks(n16, 16), by), dim3(256), 0, ctx->stream, dm, (double *)out, (const double *)psi, epi, ea, ctx->d_partials, dm.short_off, n16);
          else hipLaunchKernelGGL((k_apply_short<1, 16>), dim3(blocks(n16, 16), by), dim3(256), 0, ctx->stream, dm, (double *)out, (const double *)psi, epi, ea, ctx->d_partials, dm.short_off, n16);
        }
        if (n1 > 0) {
          if (dtype == SD_C128) hipLaunchKernelGGL((k_apply_short<2, 1>), dim3(blocks(n1, 256), by), dim3(256), 0, ctx->stream, dm, (double *)out, (const double *)psi, epi, ea, ctx->d_partials, dm.short_off + n16, n1);
          else hipLaunchKernelGGL((k_apply_short<1, 1>), dim3(blocks(n1, 256), by), dim3(256), 0, ctx->stream, dm, (double *)out, (const double *)psi, epi, ea, ctx->d_partials, dm.short_off + n16, n1);
        }
        SD_HIP(ctx, hipGetLastError());
      }
    }
    if (sums && part != 1 && !ea.no_reduce) {
      int rc2 = ea.batch > 1 ? sd_reduce_pairs_batched(ctx, (int64_t)dm.n_singles, ea.batch, ea.sums_dst, ea.sums_bstride)
                             : sd_reduce_pairs(ctx, (int64_t)dm.n_singles, ea.sums_dst);
      if (rc2) return rc2;
    }
  } else if (m->full_ls > 0) {
    // sharded by the top index bits: every tile may read the halo, so there is no interior part (part 1 launches nothing)
    if (part == 1) return SD_OK;
    const int64_t nb = dm.n_local >> m->full_ls;
    if (sums) { int rc = sd_ensure_partials(ctx, 2 * (size_t)nb + 2 * SD_RED_STAGE_BLOCKS); if (rc) return rc; }
    const size_t esz = dtype == SD_C128 ? 16 : 8;
    const size_t shmem = (((size_t)1 << m->full_ls) + 1) * esz + 32 * sizeof(double) + 16;
    void (*kf)(sd_dev_model, double *, const double *, int, sd_epi_args, double *) =
        dtype == SD_C128 ? (m->hop_pow2 ? k_apply_fulltile<2, true> : k_apply_fulltile<2, false>)
                         : (m->hop_pow2 ? k_apply_fulltile<1, true> : k_apply_fulltile<1, false>);
    hipLaunchKernelGGL(kf, dim3((unsigned)nb), dim3(256), shmem, ctx->stream, dm, (double *)out, (const double *)psi, epi, ea,
                       ctx->d_partials);
    SD_HIP(ctx, hipGetLastError());
    if (sums) {
      int rc2 = sd_reduce_pairs(ctx, (int64_t)nb, ea.sums_dst);
      if (rc2) return rc2;
    }
  } else {
    int64_t nb = (dm.N + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (sums) { int rc = sd_ensure_partials(ctx, 2 * (size_t)nb + 2 * SD_RED_STAGE_BLOCKS); if (rc) return rc; }
    if (dtype == SD_C128)
      hipLaunchKernelGGL(k_apply_generic<2>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dm, (double *)out,
                         (const double *)psi, epi, ea, ctx->d_partials);
    else
      hipLaunchKernelGGL(k_apply_generic<1>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, dm, (double *)out,
                         (const double *)psi, epi, ea, ctx->d_partials);
    SD_HIP(ctx, hipGetLastError());
    if (sums) {
      int rc2 = sd_reduce_pairs(ctx, (int64_t)nb, ea.sums_dst);
      if (rc2) return rc2;
    }
  }
  return SD_OK;
}


