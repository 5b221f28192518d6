// k_apply_orbit: H|psi> for unsharded open-chain sectors of large systems (gfx950).  Same operator, same per-row
// operation order and the same bits as k_apply_tiled (reference: src/Hamiltonian.jl:211-273); what changes is which
// rows a workgroup keeps on the chip.
//
// Layout facts used (sd_internal.hpp): with the sites split into a prefix (sites 1..p) and a suffix of LSG sites, the
// rows sharing a prefix configuration P form a contiguous TILE of C(LSG, t') rows, and a hop on a prefix bond maps a
// whole tile onto another whole tile at the same in-tile offset.  The odd prefix bonds (1,2),(3,4),... are pairwise
// disjoint, so flipping one never changes whether another is flippable: the first NGEN flippable ones of a tile (its
// GENERATORS) span an orbit of 2^NGEN tiles of identical length and row order.
//
// One workgroup per orbit ("group"), thread i <-> in-tile row i of EVERY member tile:
//   1. the 2^NGEN member tiles are loaded once (2^NGEN 16-B loads in flight per thread) into one LDS image;
//      wave 0 meanwhile builds, from binomials only, the table of far-bond partner offsets of every (bond, member);
//   2. per sub-batch of TB members the accumulators live in registers; bonds are walked in the reference's order 1..L-1:
//        generator bond  -> the partner row is row i of another member: an LDS read, no memory traffic at all
//                           (these are the top bonds whose partner tiles never survive in L2);
//        other prefix bond, straddling bond -> TB coalesced streams (range-checked buffer loads, two-deep ping-pong);
//        suffix bond     -> LDS read at a partner row taken from a per-(sector,row) byte table built on the host
//                           (the same row for every member: one address, TB reads at constant offsets);
//   3. fused epilogue + store (device_common.hpp), one partial-sum pair per workgroup.
// Per row this removes NGEN of the ~(L-LSG)/2 far reads and, since the partner offsets, the diagonal's suffix part and
// the far-bond list are formed once per thread or per group instead of once per row, about two thirds of the vector
// instructions of k_apply_tiled.  Groups are dealt to the XCDs in runs of consecutive row ranges, so that the LOW prefix
// bonds find their partner tiles in the XCD's L2 (basis.cpp, sd_build_orbit_plan).
// Tiles with fewer than NGEN flippable odd bonds form smaller orbits handled by the same kernel (member mask).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_common.hpp"

using namespace sd_dev;

namespace {

__device__ __forceinline__ int uni32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni64(int64_t v) {
  const uint32_t lo = (uint32_t)uni32((int)(uint32_t)v), hi = (uint32_t)uni32((int)(uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
// buffer descriptor from wave-uniform operands, forced into scalar registers (a descriptor left in vector registers
// costs a waterfall loop and a full vmcnt(0) drain per load)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_u(const void *p, uint32_t bytes) {
  return make_rsrc(reinterpret_cast<const void *>(uni64(reinterpret_cast<int64_t>(p))), (uint32_t)uni32((int)bytes));
}
__device__ __forceinline__ uint32_t low_mask(int nbits) { return nbits >= 32 ? 0xffffffffu : ((1u << nbits) - 1u); }

template <int NC, bool FMA, int BLOCK, int NGEN, int TB, int LSG>
__global__ __launch_bounds__(BLOCK, 2) void k_apply_orbit(sd_dev_model dm, double *__restrict__ out_,
                                                         const double *__restrict__ psi_, int epi, sd_epi_args ea,
                                                         double *__restrict__ partials, int seg_off) {
  using V = typename VT<NC>::type;
  constexpr uint32_t ES = sizeof(V);
  constexpr int NT = 1 << NGEN;
  static_assert(NT % TB == 0 && TB >= 1 && TB <= 16, "sub-batch size");
  static_assert(LSG >= 3 && LSG <= 12, "suffix sites");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V *tiles = reinterpret_cast<V *>(smem);                                     // NT x BLOCK rows; rows >= len are zero
  int64_t *ftab = reinterpret_cast<int64_t *>(smem + (size_t)NT * BLOCK * ES); // [32][NT] partner base of (bond, member), -1: none
  int64_t *tbase = ftab + 32 * NT;                                            // [NT] first row of member t
  double *thead = reinterpret_cast<double *>(tbase + NT);                     // [NT] prefix part of the list-order diagonal
  uint32_t *tpre = reinterpret_cast<uint32_t *>(thead + NT);                  // [NT] prefix configuration of member t
  double *red = reinterpret_cast<double *>(tpre + NT);                        // 32 doubles

  const V *__restrict__ psi = reinterpret_cast<const V *>(psi_);
  const int tid = threadIdx.x, lane = tid & 63;
  const int gid = blockIdx.x + seg_off;
  // one 64-byte record per group: nothing below waits for a second table.  Every field is wave-uniform; the explicit
  // readfirstlane keeps them (and everything derived: buffer descriptors, bond masks) in scalar registers
  const sd_orb_rec *__restrict__ rp = dm.orb_groups + gid;
  sd_orb_rec rec;
  rec.base0 = uni64(rp->base0);
#pragma unroll
  for (int k = 0; k < SD_ORB_NGEN; ++k) rec.dg[k] = uni64(rp->dg[k]);
  rec.P0 = (uint32_t)uni32((int)rp->P0);
  rec.gens = (uint32_t)uni32((int)rp->gens);
  rec.len = uni32(rp->len); rec.nU = uni32(rp->nU); rec.suf_off = uni32(rp->suf_off);
  const uint32_t P0 = rec.P0;
  const int64_t base0 = rec.base0;
  const uint32_t gens = rec.gens;                     // generator bond numbers g_k, ascending, 6 bits each, 0 = unused
  const int L = dm.L, nup = dm.nup, p = dm.orb_p;
  const int kp = nup - __popc(P0);                    // up spins in the suffix: the same for every member
  const int len = rec.len;                            // C(LSG, kp)
  const int nU = rec.nU;                              // rows whose first suffix site is up = C(LSG-1, kp-1)

  // rec.dg[k]: row offset of flipping generator k from (up,down) to (down,up) = C(L-g-1, u), u = up spins beyond site
  // g+1, which no other generator changes -- so member t starts at base0 + sum of dg[k] over the set bits of t
  uint32_t present = 0;
#pragma unroll
  for (int k = 0; k < NGEN; ++k)
    if ((gens >> (6 * k)) & 63) present |= 1u << k;
  const uint32_t absent = (NT - 1) & ~present;        // member t exists iff (t & absent) == 0

  // ---- per-thread row data: partner rows of the suffix bonds (bytes), suffix configuration ----
  const int i = tid;
  const int irow = i < len ? i : len - 1;
  const uint4 pt = reinterpret_cast<const uint4 *>(dm.orb_ptab)[rec.suf_off + irow];
  const uint32_t sig = pt.w & 0xffffu;
  const uint32_t ioff = (uint32_t)i * ES;
  const double my_J = lane < dm.n_hop ? dm.hop_J[lane] : 0.0;   // lane l <-> hop l+1 (the chain bond (l+1, l+2)); read with v_readlane

  // ---- 1. member tiles -> registers -> LDS (rows >= len read 0 through the range check: the zero rows of the image) ----
  {
    V own[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      own[t] = V{};
      if ((t & absent) == 0) {
        int64_t bt = base0;
#pragma unroll
        for (int k = 0; k < NGEN; ++k)
          if ((t >> k) & 1) bt += rec.dg[k];
        buf_load(own[t], make_rsrc_u(psi + bt, (uint32_t)len * ES), ioff);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if ((t & absent) == 0) tiles[t * BLOCK + i] = own[t];
  }

  // ---- wave 0: far-bond table.  lane <-> bond b = (lane & 31) + 1 (b <= p-1 prefix bond, b == p straddling bond) ----
  if (tid < 64) {
    const int b = (lane & 31) + 1;
    // partner tile of member t through prefix bond b = its own tile +- C(L-b-1, u), u = up spins beyond site b+1
    // (SURVEY appendix B).  u differs between members only when site b+1 is the FIRST site of a generator (then that
    // generator's state decides whether site b+1 is up): two binomials per bond cover every member, one memory latency.
    int kdep = -1;
    bool isgen = false;
#pragma unroll
    for (int k = 0; k < NGEN; ++k) {
      const int g = (gens >> (6 * k)) & 63;
      if (g && g == b + 1) kdep = k;
      isgen |= (g == b);
    }
    const int u0 = nup - __popc(P0 & low_mask(b + 1));
    const int64_t c0 = b <= p - 1 ? binom_g(dm, L - b - 1, u0) : 0;
    const int64_t c1 = b <= p - 1 ? binom_g(dm, L - b - 1, u0 + 1) : 0;
    for (int t = lane >> 5; t < NT; t += 2) {
      const bool valid = (t & absent) == 0;
      uint32_t Pt = P0;
      int64_t bt = base0;
#pragma unroll
      for (int k = 0; k < NGEN; ++k) {
        const int g = (gens >> (6 * k)) & 63;
        if (g && ((t >> k) & 1)) { Pt ^= 3u << (g - 1); bt += rec.dg[k]; }
      }
      int64_t e = -1;
      if (valid && b <= p - 1) {
        if ((((Pt >> (b - 1)) ^ (Pt >> b)) & 1u) && !isgen) {
          const int64_t d = (kdep >= 0 && ((t >> kdep) & 1)) ? c1 : c0;
          e = ((Pt >> (b - 1)) & 1u) ? bt + d : bt - d;
        }
      } else if (valid && b == p) {
        // site p up:   our rows with first suffix site down (i >= nU) <-> rows bt + len + (i - nU) (the next tile)
        // site p down: our rows with first suffix site up   (i <  nU) <-> rows bt - nU + i        (inside the previous tile)
        const uint32_t bitp = (Pt >> (p - 1)) & 1u;
        const int t2q = bitp ? kp + 1 : kp - 1;
        const int n = bitp ? len - nU : nU;
        if (t2q >= 0 && t2q <= LSG && n > 0) e = (bitp ? bt + len : bt - nU) | ((int64_t)bitp << 62);
      }
      if (b <= p) ftab[(b - 1) * NT + t] = e;
    }
    if (lane < NT) {
      const int t = lane;
      uint32_t Pt = P0;
      int64_t bt = base0;
#pragma unroll
      for (int k = 0; k < NGEN; ++k) {
        const int g = (gens >> (6 * k)) & 63;
        if (g && ((t >> k) & 1)) { Pt ^= 3u << (g - 1); bt += rec.dg[k]; }
      }
      tbase[t] = bt;
      tpre[t] = Pt;
      thead[t] = dm.diag_mode == 0 ? diag_head(dm, Pt, p).d0 : 0.0;
    }
  }
  __syncthreads();

  const int zz_from = (dm.diag_mode == 0 && dm.field_zero && dm.n_zz_nn > 0 && p >= 2) ? p - 1 : 0;   // as diag_head
  EpiSums sums{0.0, 0.0};

#pragma unroll 1
  for (int t0 = 0; t0 < NT; t0 += TB) {
    // members of this sub-batch that exist (scalar mask)
    uint32_t vm = 0;
#pragma unroll
    for (int j = 0; j < TB; ++j)
      if (((t0 + j) & absent) == 0) vm |= 1u << j;
    if (vm == 0) continue;

    // far-bond entries of this sub-batch in registers: lane l <-> bond l+1
    int64_t ent[TB];
    bool any = false;
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      ent[j] = lane < p ? ftab[lane * NT + t0 + j] : -1;
      any |= ent[j] >= 0;
    }
    uint64_t mk = (uint64_t)uni64((int64_t)__ballot(any));   // bit b-1 <=> bond b has a stream for some member

    // diagonal * own
    V acc[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int t = t0 + j;
      const uint64_t s = (uint64_t)tpre[t] | ((uint64_t)sig << p);
      double d;
      if (dm.diag_mode == 0) d = diag_tail(dm, DiagHead{thead[t], zz_from}, s, p);
      else d = diag_of(dm, s);
      acc[j] = vscale(d, tiles[t * BLOCK + i]);
    }

    // One ordered stream of "batches" over the bonds 1..p that do anything for this sub-batch: a far bond (TB streams)
    // or a generator bond (partner row = row i of member t ^ (1 << k), an LDS read at consume time).  Every batch issues
    // exactly TB buffer loads -- members without the hop, and generator batches, get an empty buffer (no memory access,
    // returns 0) -- so the number of loads in flight is a compile-time constant and the waits are counted (vmcnt(TB)).
    uint64_t genbits = 0;
#pragma unroll
    for (int k = 0; k < NGEN; ++k) {
      const int g = (gens >> (6 * k)) & 63;
      if (g) genbits |= (uint64_t)1 << (g - 1);
    }
    mk = (uint64_t)uni64((int64_t)(mk | genbits));     // wave-uniform: keep the bond walk (and the descriptors built from it) scalar
    auto issue = [&](int b, V(&land)[TB], bool live = true) -> uint32_t {
      uint32_t m = 0;
      const bool strad = b == p;
#pragma unroll
      for (int j = 0; j < TB; ++j) {
        const int64_t e = rl64(ent[j], b - 1);          // generator bonds carry -1 for every member
        const bool on = live && e >= 0;
        if (on) m |= 1u << j;
        const int hb = strad ? (int)((e >> 62) & 1) : 0;
        const int n = !on ? 0 : (strad ? (hb ? len - nU : nU) : len);
        const uint32_t lo_b = hb ? (uint32_t)nU * ES : 0u;
        // rows outside [lo, lo+n) wrap to a huge unsigned offset or exceed n*ES: the load returns 0
        buf_load(land[j], make_rsrc_u(psi + (on ? (e & ~((int64_t)1 << 62)) : 0), (uint32_t)n * ES), ioff - lo_b);
      }
      return m;
    };
    auto consume = [&](int b, uint32_t m, const V(&land)[TB]) {
      const double J = rld(my_J, b - 1);
      if ((genbits >> (b - 1)) & 1) {
        int k = 0;
#pragma unroll
        for (int q = 1; q < NGEN; ++q)
          if ((int)((gens >> (6 * q)) & 63) == b) k = q;
        const int flip = 1 << k;
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[j] = accum<FMA>(acc[j], J, tiles[((t0 + j) ^ flip) * BLOCK + i]);
      } else {
#pragma unroll
        for (int j = 0; j < TB; ++j)
          if ((m >> j) & 1u) acc[j] = accum<FMA>(acc[j], J, land[j]);
      }
    };
    auto next_bond = [&](uint64_t &m_) { const int ln = uni32(__builtin_ctzll(m_)); m_ &= m_ - 1; return ln + 1; };   // ascending

    // ---- 2. the batches two at a time: 2*TB loads in flight per thread, consumed in bond order.  Nothing but the
    // accumulators lives across iterations (register sets carried around a loop cost copies and full drains) ----
    while (mk) {
      V la[TB], lb[TB];
      const int ba = next_bond(mk);
      const uint32_t ma = issue(ba, la);
      const int bb = mk ? next_bond(mk) : 0;
      const uint32_t mb = issue(bb ? bb : 1, lb, bb != 0);
      consume(ba, ma, la);
      if (bb) consume(bb, mb, lb);
    }

    // ---- suffix bonds: one partner row per (thread, bond) for all members; reads of bond a+1 in flight while bond a
    // is accumulated (the loop is kept rolled: fully unrolled, the scheduler hoists all (LSG-1)*TB LDS reads and spills) ----
    {
      const V *__restrict__ tb0 = tiles + t0 * BLOCK;
      const uint64_t plo = (uint64_t)pt.x | ((uint64_t)pt.y << 32);
      auto prow = [&](int a) -> uint32_t {      // partner row of suffix bond a, or the zero row BLOCK-1
        return a <= 8 ? (uint32_t)(plo >> (8 * (a - 1))) & 255u : (pt.z >> (8 * (a - 9))) & 255u;
      };
      auto lds_rows = [&](int a, V(&v)[TB]) {
        const uint32_t pr = prow(a);
#pragma unroll
        for (int j = 0; j < TB; ++j) v[j] = tb0[j * BLOCK + pr];
      };
      auto add_rows = [&](int a, const V(&v)[TB]) {
        const double J = rld(my_J, p + a - 1);
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[j] = accum<FMA>(acc[j], J, v[j]);
      };
      V va[TB], vb[TB];
      lds_rows(1, va);
#pragma unroll 1
      for (int a = 1; a <= LSG - 1; a += 2) {
        if (a + 1 <= LSG - 1) lds_rows(a + 1, vb);
        add_rows(a, va);
        if (a + 2 <= LSG - 1) lds_rows(a + 2, va);
        if (a + 1 <= LSG - 1) add_rows(a + 1, vb);
      }
    }

    // ---- 3. epilogue + store ----
    if (i < len) {
#pragma unroll
      for (int j = 0; j < TB; ++j)
        if ((vm >> j) & 1u) {
          const int t = t0 + j;
          epilogue<NC>(epi, ea, tbase[t] + i, acc[j], tiles[t * BLOCK + i], out_, sums);
        }
    }
  }
  if (epi_has_sums(epi)) {
    double a = sums.s0, b = sums.s1;
    block_reduce2(a, b, red);
    if (tid == 0) { partials[2 * (size_t)gid] = a; partials[2 * (size_t)gid + 1] = b; }
  }
}

template <int NC, bool FMA, int BLOCK, int TB>
int launch_cfg(sd_ctx *ctx, const sd_dev_model &dm, int ng, int seg_off, double *out, const double *psi, int epi,
               const sd_epi_args &ea) {
  constexpr int NGEN = SD_ORB_NGEN, LSG = SD_ORB_LS, NT = 1 << NGEN;
  const size_t shmem = (size_t)NT * BLOCK * (NC * 8) + (size_t)32 * NT * 8 + (size_t)NT * (8 + 8 + 4) + 32 * 8 + 16;
  auto kern = k_apply_orbit<NC, FMA, BLOCK, NGEN, TB, LSG>;
  if (shmem > 48 * 1024)
    SD_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kern, dim3(ng), dim3(BLOCK), shmem, ctx->stream, dm, out, psi, epi, ea, ctx->d_partials, seg_off);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

template <int NC, bool FMA>
int launch_block(sd_ctx *ctx, const sd_dev_model &dm, int block, int tb, int ng, int seg_off, double *out, const double *psi,
                 int epi, const sd_epi_args &ea) {
  if (tb == 4) {
    switch (block) {
      case 64: return launch_cfg<NC, FMA, 64, 4>(ctx, dm, ng, seg_off, out, psi, epi, ea);
      case 128: return launch_cfg<NC, FMA, 128, 4>(ctx, dm, ng, seg_off, out, psi, epi, ea);
      case 256: return launch_cfg<NC, FMA, 256, 4>(ctx, dm, ng, seg_off, out, psi, epi, ea);
    }
  } else {
    switch (block) {
      case 64: return launch_cfg<NC, FMA, 64, 8>(ctx, dm, ng, seg_off, out, psi, epi, ea);
      case 128: return launch_cfg<NC, FMA, 128, 8>(ctx, dm, ng, seg_off, out, psi, epi, ea);
      case 256: return launch_cfg<NC, FMA, 256, 8>(ctx, dm, ng, seg_off, out, psi, epi, ea);
    }
  }
  return sd_set_err(ctx, SD_EINTERNAL, "bad orbit workgroup size");
}

}  // namespace

int sd_launch_apply_orbit(sd_ctx *ctx, const sd_model *m, int dtype, void *out, const void *psi, int epi,
                          const sd_epi_args &ea) {
  const bool sums = (epi == SD_EPI_DOT || epi == SD_EPI_KPM || epi == SD_EPI_RESCALE_DOT);
  const int ng_all = (int)m->orb_groups.size();
  if (sums) { int rc = sd_ensure_partials(ctx, 2 * (size_t)ng_all + 2 * SD_RED_STAGE_BLOCKS); if (rc) return rc; }
  const char *tbe = getenv("SD_ORB_TB");            // members per register sub-batch: 8 (default) or 4
  const int tb = (tbe && atoi(tbe) == 4) ? 4 : 8;
  for (int c = 0; c < SD_ORB_N_CLASS; ++c) {
    const int ng = m->orb_seg_off[c + 1] - m->orb_seg_off[c];
    if (ng <= 0) continue;
    const int block = m->orb_seg_block[c];
    int rc;
    if (dtype == SD_C128)
      rc = m->hop_pow2 ? launch_block<2, true>(ctx, m->dm, block, tb, ng, m->orb_seg_off[c], (double *)out, (const double *)psi, epi, ea)
                       : launch_block<2, false>(ctx, m->dm, block, tb, ng, m->orb_seg_off[c], (double *)out, (const double *)psi, epi, ea);
    else
      rc = m->hop_pow2 ? launch_block<1, true>(ctx, m->dm, block, tb, ng, m->orb_seg_off[c], (double *)out, (const double *)psi, epi, ea)
                       : launch_block<1, false>(ctx, m->dm, block, tb, ng, m->orb_seg_off[c], (double *)out, (const double *)psi, epi, ea);
    if (rc) return rc;
  }
  if (sums) return sd_reduce_pairs(ctx, (int64_t)ng_all, ea.sums_dst);
  return SD_OK;
}
