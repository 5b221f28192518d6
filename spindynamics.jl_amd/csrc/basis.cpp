// Host-side construction of the basis tables and the (optionally sharded)
// tile plan.  Reproduces the order of build_sector_basis (reference
// src/Basis.jl:37-53) from binomial coefficients only: no states[] array and
// no hash map are ever built for the full dimension.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sd_internal.hpp"

int64_t sd_binom(int n, int k) {
  if (k < 0 || k > n || n < 0) return 0;
  if (k > n - k) k = n - k;
  __int128 r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return (int64_t)r;
}

void sd_fill_binom(std::vector<int64_t> &tab) {
  const int W = SD_MAX_L + 1;
  tab.assign((size_t)W * W, 0);
  for (int n = 0; n < W; ++n)
    for (int k = 0; k <= n; ++k) tab[(size_t)n * W + k] = sd_binom(n, k);
}

static inline int64_t B(const sd_model *m, int n, int k) {
  if (k < 0 || n < 0 || k > n) return 0;
  return m->binom[(size_t)n * (SD_MAX_L + 1) + k];
}

// idx0 -> state: walk sites 1..L, "up" child first.
uint64_t sd_unrank_host(const sd_model *m, int64_t idx0) {
  if (m->nup < 0) return (uint64_t)idx0;
  uint64_t s = 0;
  int r = m->nup;
  for (int k = 1; k <= m->L && r > 0; ++k) {
    int64_t c = B(m, m->L - k, r - 1);  // rows with site k up
    if (idx0 < c) { s |= (uint64_t)1 << (k - 1); --r; }
    else idx0 -= c;
  }
  return s;
}

// state -> idx0, -1 when s is not a basis state (get(idxmap, s, 0) == 0)
int64_t sd_rank_host(const sd_model *m, uint64_t s) {
  if (m->L < 64 && (s >> m->L) != 0) return -1;
  if (m->nup < 0) return (int64_t)s;
  if (__builtin_popcountll(s) != m->nup) return -1;
  int64_t idx = 0;
  int r = m->nup;
  for (int k = 1; k <= m->L && r > 0; ++k) {
    if ((s >> (k - 1)) & 1) --r;
    else idx += B(m, m->L - k, r - 1);
  }
  return idx;
}

// Enumerate the sector (n sites, t ups) in reference order into out (as n-bit ints).
static void enum_sector(int n, int t, std::vector<uint16_t> &out) {
  if (t < 0 || t > n) return;
  std::vector<int> c(t);
  for (int k = 0; k < t; ++k) c[k] = k + 1;
  for (;;) {
    uint32_t s = 0;
    for (int k = 0; k < t; ++k) s |= 1u << (c[k] - 1);
    out.push_back((uint16_t)s);
    int k = t - 1;
    while (k >= 0 && c[k] == n - t + k + 1) --k;
    if (k < 0) break;
    ++c[k];
    for (int q = k + 1; q < t; ++q) c[q] = c[q - 1] + 1;
  }
}

static int suffix_bits_from_env() {
  int ls = 12;   // C(12,6) = 924-row tiles: 14.8 KB of LDS in ComplexF64, 5 workgroups of 256 threads per CU
  if (const char *e = getenv("SD_SUFFIX_BITS")) ls = atoi(e);
  if (ls < 2) ls = 2;
  if (ls > 15) ls = 15;
  return ls;
}

// base row (global) of the tile with prefix P
static int64_t tile_base_global(const sd_model *m, uint32_t P) {
  int64_t idx = 0;
  int r = m->nup;
  for (int k = 1; k <= m->p && r > 0; ++k) {
    if ((P >> (k - 1)) & 1) --r;
    else idx += B(m, m->L - k, r - 1);
  }
  return idx;
}

struct TileRef { int64_t base; uint32_t P; };

// marks in need[] every tile (by prefix) that the rows of tiles [k_lo,k_hi) read through a hop
static void collect_needs(const sd_model *m, const std::vector<TileRef> &tiles, const std::vector<int> &owner, int q,
                          int nn_hops, std::vector<uint8_t> &need) {
  const int p = m->p;
  for (size_t k = 0; k < tiles.size(); ++k) {
    if (owner[k] != q) continue;
    uint32_t P = tiles[k].P;
    if (nn_hops > 0) {
      for (int b = 1; b <= p - 1; ++b)
        if (((P >> (b - 1)) & 1) != ((P >> b) & 1)) need[P ^ (3u << (b - 1))] = 1;
      if (p >= 1) {
        uint32_t Q = P ^ (1u << (p - 1));
        int t2 = m->nup - __builtin_popcount(Q);
        if (t2 >= 0 && t2 <= m->LS) need[Q] = 1;
      }
    }
    for (size_t h = (size_t)nn_hops; h < m->hop_i.size(); ++h) {
      int i = m->hop_i[h], j = m->hop_j[h];
      bool ip = i <= p, jp = j <= p;
      uint32_t Q = P;
      if (ip && jp) {
        if (((P >> (i - 1)) & 1) == ((P >> (j - 1)) & 1)) continue;
        Q = P ^ (1u << (i - 1)) ^ (1u << (j - 1));
      } else if (ip) Q = P ^ (1u << (i - 1));
      else if (jp) Q = P ^ (1u << (j - 1));
      else continue;
      int t2 = m->nup - __builtin_popcount(Q);
      if (t2 >= 0 && t2 <= m->LS) need[Q] = 1;
    }
  }
}

static int count_nn_hops(const sd_model *m) {
  int L = m->L;
  if (L < 2 || (int)m->hop_i.size() < L - 1) return 0;
  for (int k = 0; k < L - 1; ++k)
    if (m->hop_i[k] != k + 1 || m->hop_j[k] != k + 2) return 0;
  return L - 1;
}

// Reorders one launch segment (tp_io, tb_io) in place.  first_seen: 2^p scratch entries, all -1 on entry and on return.
// XCD-aware processing order.  Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8), each with
// a private 4 MiB L2.  A far bond maps tile P onto tile P' = P ^ bond and both read each other.  Tiles related by
// DISJOINT flippable top bonds (the odd prefix bonds (1,2),(3,4),..: flipping one never changes another) form orbits
// of 2^f tiles; the members of an orbit are queued back to back on ONE XCD, so that a member's partner reads for
// those bonds -- the first far bonds it processes, right after its own rows were fetched by the partner -- hit
// that XCD's L2 (or merge with the in-flight fetch) instead of going to the fabric again.  SD_XCD_ORBIT = max f
// (0 disables, then SD_XCD_CHUNK consecutive tiles per XCD are used).  Speed only: a bijection of the tile list.
static void xcd_order(const sd_model *m, int p, std::vector<uint32_t> &tp_io, std::vector<int64_t> &tb_io,
                      std::vector<int64_t> &first_seen) {
  {
    int FO = 5;     // measured with 5 workgroups per CU in flight (profiles/ablation_r02.md §14): 5 beats 6 by 1 % at L=28..32
    if (const char *e = getenv("SD_XCD_ORBIT")) FO = atoi(e);
    if (FO > 8) FO = 8;
    int CH = 32;
    if (const char *e = getenv("SD_XCD_CHUNK")) CH = atoi(e);
    int B0 = 1;     // first bond considered as an orbit generator (two-pass probe: the bonds below belong to the other pass)
    if (const char *e = getenv("SD_XCD_ORBIT_FROM")) B0 = std::max(1, atoi(e));
    const size_t nt = tp_io.size();
    // (Tried: the "transposed" order -- tiles sorted by (filling of the prefix, prefix as an integer: site 1 varies fastest), dealt to
    // the XCDs in runs of K.  The LRU model of profiles/l2_order_lab.py (pasc<K>) promises 25 % fewer read misses; the hardware
    // counters show none (FETCH_SIZE +3 % / +0.4 % at K = 256 / 1024) and the apply is 1-2 % slower, 9-19 % with K >= 4096.
    // Removed; profiles/ablation_r03.md section 11.)
    // (Tried: with a wrap bond (1, j), j in the suffix -- the periodic chain's (L, 1) -- site 1 alone as an extra generator, so
    // that P and P ^ 1 are queued next to each other and the wrap bond's gather finds its partner tile in L2.  Periodic L=28 /
    // L=30: 0.775 / 3.127 ms with it, 0.770 / 3.117 without -- no gain, removed; profiles/ablation_r03.md section 5.)
    const bool wrap = false;
    const int Bp = wrap ? 2 : B0, FOp = wrap ? std::max(FO - 1, 0) : FO;
    // canonical orbit representative: chosen pairs set to (up, down) (site 1 down with a wrap bond); member id = what is flipped
    auto canon = [&](uint32_t P, int *member_out) {
      uint32_t C0 = P; int member = 0, ng = 0, sh = 0;
      if (wrap) { member = (int)(P & 1u); C0 &= ~1u; sh = 1; }
      for (int b = Bp; b + 1 <= p && ng < FOp; b += 2)
        if (((P >> (b - 1)) ^ (P >> b)) & 1u) {
          if (!((P >> (b - 1)) & 1u)) { C0 ^= 3u << (b - 1); member |= 1 << (ng + sh); }
          ++ng;
        }
      if (member_out) *member_out = member;
      return C0;
    };
    auto first_seen_reset = [&](uint32_t P) { first_seen[canon(P, nullptr)] = -1; };
    if (FO > 0 && nt >= 64 && p >= 3 && count_nn_hops(m) > 0) {
      std::vector<uint64_t> key(nt);     // (first-seen rank of the orbit) << 8 | member id
      int64_t n_orb = 0;
      for (size_t k = 0; k < nt; ++k) {
        int member = 0;
        const uint32_t C0 = canon(tp_io[k], &member);
        if (first_seen[C0] < 0) first_seen[C0] = n_orb++;
        key[k] = ((uint64_t)first_seen[C0] << 8) | (uint64_t)member;
      }
      std::vector<size_t> idx(nt);
      for (size_t k = 0; k < nt; ++k) idx[k] = k;
      std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return key[a] < key[b]; });
      // deal whole orbits to the 8 XCD queues, then interleave the queues (position j of queue x -> block 8j + x)
      std::vector<std::vector<size_t>> q(8);
      size_t o = 0;
      size_t OC = 1;   // consecutive orbits handed to the same XCD
      if (const char *e = getenv("SD_XCD_ORBIT_RUN")) OC = (size_t)std::max(1, atoi(e));
      for (size_t k = 0; k < nt;) {
        size_t e = k;
        while (e < nt && (key[idx[e]] >> 8) == (key[idx[k]] >> 8)) ++e;
        for (size_t t = k; t < e; ++t) q[(o / OC) % 8].push_back(idx[t]);
        ++o; k = e;
      }
      std::vector<uint32_t> tp; std::vector<int64_t> tb;
      tp.reserve(nt); tb.reserve(nt);
      size_t longest = 0;
      for (auto &v : q) longest = std::max(longest, v.size());
      for (size_t j = 0; j < longest; ++j)
        for (int x = 0; x < 8; ++x)
          if (j < q[x].size()) { tp.push_back(tp_io[q[x][j]]); tb.push_back(tb_io[q[x][j]]); }
      for (size_t k = 0; k < nt; ++k) first_seen_reset(tp_io[k]);
      tp_io.swap(tp); tb_io.swap(tb);
    } else if (CH > 0 && nt >= (size_t)16 * CH) {
      std::vector<uint32_t> tp(nt);
      std::vector<int64_t> tb(nt);
      const size_t group = (size_t)8 * CH, full = nt / group * group;
      for (size_t b = 0; b < nt; ++b) {
        size_t t = b;
        if (b < full) {
          const size_t x = b % 8, sl = b / 8, g = sl / CH, i = sl % CH;
          t = g * group + x * CH + i;
        }
        tp[b] = tp_io[t]; tb[b] = tb_io[t];
      }
      tp_io.swap(tp); tb_io.swap(tb);
    }
  }
}

// ---- ownership by nested bisections of popcount cells (nranks = 2, 4, 8) ----
// d nested site blocks 1..s[d] > 1..s[d-1] > .. > 1..s[1]; rank bit i says on which side of a threshold the number of up
// spins of one of these blocks lies, longest block first, the threshold of each further block being the conditional median
// given the sides chosen before it (every rank gets about N/nranks rows).  A boundary bond (s[i], s[i]+1) changes the count
// of exactly one block by one, so only rows whose count sits at that block's threshold cross a cut, every cut has one
// partner rank, and a cut costs about the probability of sitting at the threshold (~ 1/sqrt(block length)): the longer the
// blocks the better, hence nested blocks.  Evaluated from binomials over the disjoint pieces of the blocks (no loop over
// rows or tiles); used when it imports less than the run-of-cells cut below.
struct BisectPlan {
  int d = 0;
  int s[4] = {0, 0, 0, 0};            // piece i = sites s[i]+1 .. s[i+1]; block j = sites 1..s[j]
  int thr[3][4] = {{0}};              // thr[level][sides chosen so far]: side = (count of block d-level > thr)
  double score = 1e300;               // worst rank's imported rows per owned row
  double busiest = 0;                 // rows on the busiest (receiver, owner) pair
};

static int bisect_owner(const BisectPlan &bp, const int *k) {   // k[i] = up spins of piece i
  int cum[4] = {0, 0, 0, 0};
  for (int i = 0; i < bp.d; ++i) cum[i + 1] = cum[i] + k[i];
  int r = 0;
  for (int lev = 0; lev < bp.d; ++lev) r = 2 * r + (cum[bp.d - lev] > bp.thr[lev][r] ? 1 : 0);
  return r;
}

static bool eval_bisect(const sd_model *m, BisectPlan &bp) {
  const int d = bp.d, L = m->L, nup = m->nup, P = 1 << d;
  int len[3] = {0, 0, 0};
  for (int i = 0; i < d; ++i) len[i] = bp.s[i + 1] - bp.s[i];
  const int rest = L - bp.s[d];
  auto weight = [&](const int *k) {
    int sum = 0; double w = 1;
    for (int i = 0; i < d; ++i) { w *= (double)B(m, len[i], k[i]); sum += k[i]; }
    return w * (double)B(m, rest, nup - sum);
  };
  auto for_cells = [&](auto &&fn) {
    int k[3] = {0, 0, 0};
    for (k[0] = 0; k[0] <= len[0]; ++k[0])
      for (k[1] = 0; k[1] <= (d > 1 ? len[1] : 0); ++k[1])
        for (k[2] = 0; k[2] <= (d > 2 ? len[2] : 0); ++k[2]) fn(k);
  };
  // thresholds level by level: the weighted median of the block count among the cells of each side pattern so far
  for (int lev = 0; lev < d; ++lev)
    for (int pre = 0; pre < (1 << lev); ++pre) {
      const int blk = d - lev;                       // block 1..s[blk]
      std::vector<double> hist(bp.s[blk] + 1, 0.0);
      double tot = 0;
      for_cells([&](const int *k) {
        int cum[4] = {0, 0, 0, 0};
        for (int i = 0; i < d; ++i) cum[i + 1] = cum[i] + k[i];
        int r = 0;
        for (int q = 0; q < lev; ++q) r = 2 * r + (cum[d - q] > bp.thr[q][r] ? 1 : 0);
        if (r != pre) return;
        const double w = weight(k);
        hist[cum[blk]] += w; tot += w;
      });
      if (tot <= 0) return false;
      double acc = 0, bestd = 1e300; int t = 0;
      for (int v = 0; v < bp.s[blk]; ++v) {          // side 0 = counts <= v
        acc += hist[v];
        const double dd = std::fabs(acc - tot / 2);
        if (dd < bestd) { bestd = dd; t = v; }
      }
      bp.thr[lev][pre] = t;
    }
  std::vector<double> size(P, 0.0), vol(P, 0.0), pair((size_t)P * P, 0.0);
  for_cells([&](const int *k) {
    int sum = 0;
    for (int i = 0; i < d; ++i) sum += k[i];
    const int kr = nup - sum;
    if (kr < 0 || kr > rest) return;
    const int me = bisect_owner(bp, k);
    size[me] += weight(k);
    for (int i = 0; i < d; ++i) {                   // boundary bond (s[i+1], s[i+1]+1): piece i | piece i+1 (or the rest)
      const bool last = i == d - 1;
      const int lr = last ? rest : len[i + 1], kright = last ? kr : k[i + 1];
      double others = 1;
      for (int q = 0; q < d; ++q) if (q != i && (last || q != i + 1)) others *= (double)B(m, len[q], k[q]);
      if (!last) others *= (double)B(m, rest, kr);
      for (int dir = 0; dir < 2; ++dir) {            // 0: (up, down) -> (down, up): k_i - 1, k_right + 1;  1: the reverse
        const double n = dir == 0 ? (double)B(m, len[i] - 1, k[i] - 1) * (double)B(m, lr - 1, kright) * others
                                  : (double)B(m, len[i] - 1, k[i]) * (double)B(m, lr - 1, kright - 1) * others;
        if (n <= 0) continue;
        int k2[3] = {k[0], k[1], k[2]};
        k2[i] += dir == 0 ? -1 : 1;
        if (!last) k2[i + 1] += dir == 0 ? 1 : -1;
        const int o = bisect_owner(bp, k2);
        if (o != me) { vol[me] += n; pair[(size_t)me * P + o] += n; }
      }
    }
  });
  double tot = 0;
  for (int r = 0; r < P; ++r) tot += size[r];
  bp.score = 0; bp.busiest = 0;
  for (int r = 0; r < P; ++r) {
    if (size[r] <= 0 || size[r] > 1.15 * tot / P) return false;
    bp.score = std::max(bp.score, vol[r] / size[r]);
  }
  for (double v : pair) bp.busiest = std::max(bp.busiest, v);
  return true;
}

static bool best_bisect_plan(const sd_model *m, int p, int nranks, BisectPlan &best) {
  int d = 0;
  while ((1 << d) < nranks) ++d;
  if ((1 << d) != nranks || d < 1 || d > 3 || m->nup < 0) return false;
  bool found = false;
  BisectPlan bp; bp.d = d;
  for (int s1 = 2; s1 <= p - 1; ++s1)
    for (int s2 = (d > 1 ? s1 + 2 : s1); s2 <= (d > 1 ? p - 1 : s1); ++s2)
      for (int s3 = (d > 2 ? s2 + 2 : s2); s3 <= (d > 2 ? p - 1 : s2); ++s3) {
        bp.s[0] = 0; bp.s[1] = s1; bp.s[2] = s2; bp.s[3] = s3;
        if (!eval_bisect(m, bp)) continue;
        if (getenv("SD_SHARD_DEBUG") && atoi(getenv("SD_SHARD_DEBUG")) > 1)
          fprintf(stderr, "[sd shard]   s=(%d,%d,%d) import %.3f busiest %.3g\n", s1, s2, s3, bp.score, bp.busiest);
        // the worst rank's import first, then the busiest pair (xGMI is point to point)
        if (!found || bp.score < best.score * (1 - 1e-12) ||
            (bp.score <= best.score * (1 + 1e-12) && bp.busiest < best.busiest)) { best = bp; found = true; }
      }
  return found;
}

int sd_build_plan(sd_model *m, int rank, int nranks, std::string &err) {
  if (nranks < 1 || rank < 0 || rank >= nranks) { err = "bad shard rank/nranks"; return SD_EARG; }
  m->rank = rank; m->nranks = nranks;
  m->tile_prefix.clear(); m->tile_base.clear(); m->addr.clear();
  m->suf_states.clear(); m->suf_rank.clear(); m->suf_off.clear(); m->suf_part.clear();
  m->recv_slabs.clear(); m->send_slabs.clear();
  m->n_halo = 0; m->max_tile_len = 0;
  m->n_short = 0; m->short_off = 0; m->n_short_multi = 0;

  const int L = m->L;
  int LS = std::min(L, suffix_bits_from_env());
  if (m->nup >= 0) {
    // the largest workgroup (1024 threads x 4 rows) must hold the longest tile, C(LS, t) over the feasible suffix fillings t
    auto longest = [&](int ls) {
      int64_t mx = 0;
      for (int t = std::max(0, m->nup - (L - ls)); t <= std::min(ls, m->nup); ++t) mx = std::max(mx, B(m, ls, t));
      return mx;
    };
    while (LS > 2 && longest(LS) > 4096) --LS;
  }
  int p = L - LS;
  m->max_tile_len_all = 0;
  if (m->nup >= 0)
    for (int t = std::max(0, m->nup - (L - LS)); t <= std::min(LS, m->nup); ++t)
      m->max_tile_len_all = std::max<int>(m->max_tile_len_all, (int)std::min<int64_t>(B(m, LS, t), 1 << 30));
  // (Prefix spaces beyond 2^26 -- dilute sectors of chains with L >= 39 at LS = 12 -- keep the per-row path: the plan's dense prefix
  // tables cost 13 B x 2^p on the host and its tile order visits all 2^p prefixes; measured with the cap at 28: L=40, nup=10 plans in
  // 110 s for an apply of ~50 instead of 528 ms, profiles/ablation_r04.md section 11.  SD_PLAN_TIMING=1 prints where a plan's seconds go.)
  // Small, very dilute sectors: with fewer than 32 rows per tile on average and fewer than 1.5 M rows in all the apply is a handful of
  // launches of mostly one-row workgroups, and the per-row path's single launch is as fast or faster (L=30, nup=6: 12.0 against 9.3 G
  // rows/s; L=32, nup=6: 13.4 against 13.2; larger dilute sectors go to the tiles, whose short ones have a kernel of their own:
  // L=36, nup=6: 17.1 against 14.2 -- profiles/ablation_r04.md section 11).  Unsharded plans only: a shard is a union of tiles.
  // SD_FORCE_PER_ROW / SD_FORCE_TILED: A/B.
  bool short_tiles = false;
  if (m->nup >= 0 && p >= 0 && p <= SD_MAX_PREFIX_BITS && nranks == 1) {
    double nt = 0;
    for (int k = std::max(0, m->nup - LS); k <= std::min(p, m->nup); ++k) nt += (double)B(m, p, k);
    short_tiles = nt > 0 && (double)m->N / nt < 32.0 && p >= 1 && m->N < 1500000;
    if (getenv("SD_FORCE_PER_ROW")) short_tiles = true;
    if (getenv("SD_FORCE_TILED")) short_tiles = false;
  }
  if (m->nup < 0 || p > SD_MAX_PREFIX_BITS || short_tiles) {
    // generic (untiled) path: per-row rank/unrank on device
    m->p = -1; m->LS = 0;
    m->row_lo = 0; m->row_hi = m->N; m->n_local = m->N;
    m->shard_mode = 0; m->n_interior = 0; m->fs_dbits = 0;
    m->pack_src.clear(); m->pack_dst.clear(); m->pack_len.clear(); m->n_send = 0;
    // full 2^L basis: idx = state, so 2^10 consecutive rows form a tile without any table (k_apply_fulltile)
    m->full_ls = (m->nup < 0 && L >= 12 && L <= 40 && !getenv("SD_NO_FULLTILE")) ? 10 : 0;
    if (nranks == 1) return SD_OK;
    // Full basis over 2^d ranks (src/Hamiltonian.jl:223,255-257: idx = state): the rank is the top d index bits, i.e. the
    // configuration of sites L-d+1..L; a rank owns the contiguous rows [r N/P, (r+1) N/P).  Chain bonds below site L-d stay
    // inside a rank.  The bond (L-d, L-d+1) straddles the cut: the rows whose site L-d differs from the rank's lowest bit
    // read, from rank r^1, the contiguous HALF of its vector whose site L-d equals this rank's lowest bit.  A bond between
    // two rank bits that differ maps the whole vector onto the whole vector of rank r ^ (3 << k), at the same offset.
    int d = 0;
    while ((1 << d) < nranks) ++d;
    const int nnh = count_nn_hops(m);
    if (m->nup >= 0) { err = "sharding needs a fixed-nup sector with at most 2^26 prefix tiles, or the full basis"; return SD_EARG; }
    if ((1 << d) != nranks || nranks > SD_FS_MAX_RANKS) { err = "the full 2^L basis shards over 2, 4 or 8 ranks (top index bits)"; return SD_EARG; }
    if (m->full_ls == 0 || L - d < m->full_ls + 1) { err = "full-basis sharding needs L >= 12 and at least two tiles per rank"; return SD_EARG; }
    if (nnh == 0 || (int)m->hop_i.size() != nnh) { err = "full-basis sharding needs the open chain's hop list (bonds (i, i+1) in order)"; return SD_EARG; }
    m->fs_dbits = d;
    m->n_local = m->N / nranks;
    m->row_lo = (int64_t)rank * m->n_local; m->row_hi = m->row_lo + m->n_local;
    for (int q = 0; q < SD_FS_MAX_RANKS; ++q) { m->fs_halo_off[q] = -1; m->fs_peer_lo[q] = 0; }
    int64_t hoff = 0;
    auto add_pair = [&](int me, int peer, int64_t peer_lo, int64_t count, bool recv) {
      if (recv) {
        m->fs_halo_off[peer] = hoff; m->fs_peer_lo[peer] = peer_lo;
        m->recv_slabs.push_back({peer, m->n_local + hoff, count, (int64_t)peer * m->n_local + peer_lo});
        hoff += count;
      } else {
        m->send_slabs.push_back({peer, peer_lo, count, (int64_t)me * m->n_local + peer_lo});
      }
    };
    // what this rank receives, in a fixed order (straddling bond first, then the bonds between rank bits, ascending)
    add_pair(rank, rank ^ 1, (int64_t)(rank & 1) * (m->n_local / 2), m->n_local / 2, true);
    for (int k = 0; k + 1 < d; ++k)
      if (((rank >> k) ^ (rank >> (k + 1))) & 1) add_pair(rank, rank ^ (3 << k), 0, m->n_local, true);
    // what the peers receive from this rank: the same rule seen from their side, in THEIR order (slab k of a pair pairs up)
    add_pair(rank, rank ^ 1, (int64_t)((rank ^ 1) & 1) * (m->n_local / 2), m->n_local / 2, false);
    for (int k = 0; k + 1 < d; ++k)
      if (((rank >> k) ^ (rank >> (k + 1))) & 1) add_pair(rank, rank ^ (3 << k), 0, m->n_local, false);
    m->n_halo = hoff;
    return SD_OK;
  }
  m->p = p; m->LS = LS; m->full_ls = 0;

  // suffix sector tables
  m->suf_off.assign(LS + 2, 0);
  m->suf_rank.assign((size_t)1 << LS, 0);
  for (int t = 0; t <= LS; ++t) {
    m->suf_off[t] = (int32_t)m->suf_states.size();
    enum_sector(LS, t, m->suf_states);
    int32_t n = (int32_t)m->suf_states.size() - m->suf_off[t];
    for (int32_t i = 0; i < n; ++i) m->suf_rank[m->suf_states[m->suf_off[t] + i]] = (uint16_t)i;
  }
  m->suf_off[LS + 1] = (int32_t)m->suf_states.size();
  // packed partner table of the suffix bonds (sd_dev_model::suf_part): 10-bit fields hold partner row + 1 <= C(12,6) = 924
  m->suf_part.clear(); m->suf_dg.clear();
  m->wrap_hop = -1;
  if (LS <= 12) {
    m->suf_part.assign(4 * m->suf_states.size(), 0u);
    m->suf_dg.assign(m->suf_states.size(), 0);
    const uint32_t inner = LS >= 2 ? ((1u << (LS - 1)) - 1u) : 0u;
    for (size_t k = 0; k < m->suf_states.size(); ++k) {
      const uint32_t sg = m->suf_states[k];
      uint32_t *w = m->suf_part.data() + 4 * k;
      for (int a = 1; a <= LS - 1; ++a)
        if (((sg >> (a - 1)) ^ (sg >> a)) & 1u) {
          const uint32_t partner = (uint32_t)m->suf_rank[sg ^ (3u << (a - 1))] + 1u;     // same sector: same popcount
          w[(a - 1) / 3] |= partner << (10 * ((a - 1) % 3));
        }
      m->suf_dg[k] = (uint8_t)(__builtin_popcount((sg ^ (sg >> 1)) & inner) | ((sg & 1u) << 4));
    }
    // the first general bond, if it joins a prefix and a suffix site (sd_dev_model::wrap_hop)
    m->wrap_hop = -1;
    const int nnh = count_nn_hops(m);
    if (nnh > 0 && nnh < (int)m->hop_i.size() && p >= 1) {
      const int bi = m->hop_i[nnh] - 1, bj = m->hop_j[nnh] - 1;
      if ((bi < p) != (bj < p)) {
        const int sb = (bi < p ? bj : bi) - p;
        m->wrap_hop = nnh; m->wrap_pb = bi < p ? bi : bj;
        for (int t = 0; t <= LS; ++t)
          for (int32_t i = m->suf_off[t]; i < m->suf_off[t + 1]; ++i) {
            const uint32_t sg = m->suf_states[i];
            const bool up = (sg >> sb) & 1u;
            const int tq = up ? t - 1 : t + 1;
            uint32_t field = 0;
            if (tq >= 0 && tq <= LS) field = (uint32_t)m->suf_rank[sg ^ (1u << sb)] + 1u;    // rank inside sector tq
            m->suf_part[4 * (size_t)i + 3] |= (field << 20) | ((up ? 1u : 0u) << 30);
          }
      }
    }
  }

  // SD_PLAN_TIMING=1: where the seconds of a plan go (stderr)
  const bool timing = getenv("SD_PLAN_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto tick = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[sd plan] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  // all feasible tiles in natural (row) order
  std::vector<TileRef> tiles;
  const uint32_t nP = 1u << p;
  for (uint32_t P = 0; P < nP; ++P) {
    int t2 = m->nup - __builtin_popcount(P);
    if (t2 < 0 || t2 > LS) continue;
    tiles.push_back({tile_base_global(m, P), P});
  }
  std::sort(tiles.begin(), tiles.end(), [](const TileRef &a, const TileRef &b) { return a.base < b.base; });
  const size_t T = tiles.size();
  if (T == 0) { err = "empty basis"; return SD_EINTERNAL; }
  tick("tiles enumerated + sorted");

  // ---- ownership of every tile ----
  // mode 0 ("range"): contiguous, tile-aligned basis-index ranges (what BASELINE.json's north star names).
  // mode 1 ("class"): a tile belongs to the cell (k1, k2) = (#up among sites 1..m1, #up among sites 1..m2), m2 < m1 < p,
  //   and the cells, taken in lexicographic order, are cut into nranks runs of equal weight.  Only the two bonds
  //   (m1,m1+1) and (m2,m2+1) can move a configuration to another cell, and only across a cut, so a rank imports 3-4x
  //   fewer rows than with index ranges (L=32: 0.15 / 0.50 / 0.74 imported rows per owned row at 2 / 4 / 8 ranks instead
  //   of 0.52 / 1.50 / 2.50).  Ownership is then a union of tiles, stored compactly in natural order.
  std::vector<int> owner(T, 0);
  std::vector<int64_t> tlen(T);
  for (size_t k = 0; k < T; ++k) tlen[k] = B(m, LS, m->nup - __builtin_popcount(tiles[k].P));
  int mode = m->shard_mode_req;
  if (mode < 0) {
    mode = 1;
    if (const char *e = getenv("SD_SHARD_MODE")) mode = (e[0] == 'r') ? 0 : 1;
  }
  if (nranks == 1 || count_nn_hops(m) == 0 || p < 6) mode = 0;
  if (mode == 1) {
    // choose (m1, m2) and the cuts: smallest worst-rank import ratio among balanced cuts (weights from binomials)
    double best = 1e300; int bm1 = -1, bm2 = -1; std::vector<int> bassign;
    for (int m1 = 4; m1 <= p - 1; ++m1)
      for (int m2 = 2; m2 <= m1 - 2; ++m2) {
        const int W2 = m2 + 1;
        std::vector<double> w((size_t)(m1 + 1) * W2, 0.0);
        double tot = 0;
        for (int k1 = 0; k1 <= m1; ++k1)
          for (int k2 = 0; k2 <= std::min(k1, m2); ++k2) {
            const double x = (double)B(m, m2, k2) * (double)B(m, m1 - m2, k1 - k2) * (double)B(m, L - m1, m->nup - k1);
            w[(size_t)k1 * W2 + k2] = x; tot += x;
          }
        std::vector<int> assign((size_t)(m1 + 1) * W2, -1);
        std::vector<double> size(nranks, 0.0), vol(nranks, 0.0);
        double acc = 0; int r = 0;
        for (size_t c = 0; c < w.size(); ++c) {
          if (w[c] == 0) continue;
          if (acc + w[c] / 2 > (r + 1) * tot / nranks && r < nranks - 1) ++r;
          assign[c] = r; acc += w[c]; size[r] += w[c];
        }
        bool ok = true;
        for (int q = 0; q < nranks; ++q) if (size[q] == 0 || size[q] > 1.15 * tot / nranks) ok = false;
        if (!ok) continue;
        auto A = [&](int k1, int k2) { return (k1 < 0 || k1 > m1 || k2 < 0 || k2 > m2 || k2 > k1) ? -1 : assign[(size_t)k1 * W2 + k2]; };
        for (int k1 = 0; k1 <= m1; ++k1)
          for (int k2 = 0; k2 <= std::min(k1, m2); ++k2) {
            const int rr = A(k1, k2);
            if (rr < 0) continue;
            const double rest = (double)B(m, L - m1, m->nup - k1);
            auto add = [&](double n, int other) { if (n > 0 && other >= 0 && other != rr) vol[rr] += n; };
            add((double)B(m, m2 - 1, k2 - 1) * (double)B(m, m1 - m2 - 1, k1 - k2) * rest, A(k1, k2 - 1));
            add((double)B(m, m2 - 1, k2) * (double)B(m, m1 - m2 - 1, k1 - k2 - 1) * rest, A(k1, k2 + 1));
            add((double)B(m, m2, k2) * (double)B(m, m1 - m2 - 1, k1 - k2 - 1) * (double)B(m, L - m1 - 1, m->nup - k1), A(k1 - 1, k2));
            add((double)B(m, m2, k2) * (double)B(m, m1 - m2 - 1, k1 - k2) * (double)B(m, L - m1 - 1, m->nup - k1 - 1), A(k1 + 1, k2));
          }
        double score = 0;
        for (int q = 0; q < nranks; ++q) score = std::max(score, vol[q] / size[q]);
        if (score < best) { best = score; bm1 = m1; bm2 = m2; bassign = assign; }
      }
    // nested bisections (nranks = 2, 4, 8) when they import less than the run-of-cells cut
    BisectPlan bis;
    const bool have_bis = !getenv("SD_SHARD_NO_BISECT") && best_bisect_plan(m, p, nranks, bis);
    if (getenv("SD_SHARD_DEBUG"))
      fprintf(stderr, "[sd shard] cells (m1=%d, m2=%d) worst import %.3f | bisections found=%d d=%d s=(%d,%d,%d) worst import %.3f busiest pair %.3g rows\n",
              bm1, bm2, best, (int)have_bis, bis.d, bis.s[1], bis.s[2], bis.s[3], bis.score, bis.busiest);
    if (have_bis && (bm1 < 0 || bis.score < best)) {
      for (size_t k = 0; k < T; ++k) {
        int cnt[3] = {0, 0, 0};
        for (int i = 0; i < bis.d; ++i) {
          const uint32_t lo = bis.s[i] == 0 ? 0u : ((1u << bis.s[i]) - 1u), hi = (1u << bis.s[i + 1]) - 1u;
          cnt[i] = __builtin_popcount(tiles[k].P & (hi & ~lo));
        }
        owner[k] = bisect_owner(bis, cnt);
      }
      // Balance.  A threshold on an integer count can only cut between two counts, and the cell at the median holds 15-20 % of a
      // branch: the ranks came out at 62-85 M rows where 75 M is even (L=32, 8 ranks).  Finer: order the configurations of a
      // branch by (count of the block, then the K sites that follow it, read as a binary number) and cut THAT order at its
      // weighted median.  The bonds inside those K sites now cross the cut too, but each with a fraction of the rows that sat at
      // the integer threshold: at L=32, 8 ranks the import is unchanged (333 M rows in all, busiest pair 17.8 M) and the ranks
      // own 71.2-77.4 M rows (K = 3).  Kept only if it does not cross more bonds (x 1.05) than the integer cut.
      int K = 3;
      if (const char *e = getenv("SD_SHARD_FINE")) K = std::max(0, std::min(6, atoi(e)));
      if (K > 0) {
        std::vector<int> fine(T, 0);
        std::vector<int32_t> idx_of((size_t)nP, -1);
        for (size_t k = 0; k < T; ++k) idx_of[tiles[k].P] = (int32_t)k;
        for (int lev = 0; lev < bis.d; ++lev) {
          const int sb = bis.s[bis.d - lev], kk = std::min(K, p - sb);
          const uint32_t bmask = (1u << sb) - 1u;
          auto zof = [&](uint32_t P) {
            uint32_t z = (uint32_t)__builtin_popcount(P & bmask);
            for (int j = 0; j < kk; ++j) z = 2 * z + ((P >> (sb + j)) & 1u);
            return z;
          };
          const size_t nz = ((size_t)sb + 1) << kk;
          std::vector<double> hist((size_t)(1 << lev) * nz, 0.0);
          for (size_t k = 0; k < T; ++k) hist[(size_t)fine[k] * nz + zof(tiles[k].P)] += (double)tlen[k];
          std::vector<uint32_t> thr((size_t)1 << lev, 0);
          for (int pre = 0; pre < (1 << lev); ++pre) {
            const double *h = hist.data() + (size_t)pre * nz;
            double tot = 0; for (size_t z = 0; z < nz; ++z) tot += h[z];
            double acc = 0, bestd = 1e300;
            for (size_t z = 0; z + 1 < nz; ++z) {
              acc += h[z];
              const double dd = std::fabs(acc - tot / 2);
              if (dd < bestd) { bestd = dd; thr[pre] = (uint32_t)z; }
            }
          }
          for (size_t k = 0; k < T; ++k) fine[k] = 2 * fine[k] + (zof(tiles[k].P) > thr[fine[k]] ? 1 : 0);
        }
        // rows whose hop partner lives on another rank (partner tiles counted once per reading tile), largest rank
        auto judge = [&](const std::vector<int> &own, double &crossing, double &largest) {
          crossing = 0;
          std::vector<double> size(nranks, 0.0);
          for (size_t k = 0; k < T; ++k) {
            const uint32_t P = tiles[k].P;
            size[own[k]] += (double)tlen[k];
            for (int b = 1; b <= p - 1; ++b)
              if (((P >> (b - 1)) ^ (P >> b)) & 1u) {
                const int32_t j = idx_of[P ^ (3u << (b - 1))];
                if (j >= 0 && own[j] != own[k]) crossing += (double)tlen[j];
              }
          }
          largest = 0;
          bool empty = false;
          for (double v : size) { largest = std::max(largest, v); if (v <= 0) empty = true; }
          return !empty;
        };
        double c0 = 0, l0 = 0, c1 = 0, l1 = 0;
        judge(owner, c0, l0);
        const bool ok1 = judge(fine, c1, l1);
        if (getenv("SD_SHARD_DEBUG"))
          fprintf(stderr, "[sd shard] fine thresholds (K=%d): largest rank %.3g -> %.3g rows, crossing rows %.4g -> %.4g: %s\n", K, l0, l1, c0,
                  c1, (ok1 && l1 < l0 && c1 <= 1.05 * c0) ? "kept" : "dropped");
        if (ok1 && l1 < l0 && c1 <= 1.05 * c0) owner.swap(fine);
      }
    } else if (bm1 < 0) mode = 0;
    else {
      const uint32_t mA = (1u << bm1) - 1, mB = (1u << bm2) - 1;
      for (size_t k = 0; k < T; ++k)
        owner[k] = bassign[(size_t)__builtin_popcount(tiles[k].P & mA) * (bm2 + 1) + __builtin_popcount(tiles[k].P & mB)];
    }
  }
  if (mode == 0) {
    for (int r = 1; r < nranks; ++r) {
      const int64_t target = (int64_t)((__int128)m->N * r / nranks);
      for (size_t k = 0; k < T; ++k) if (tiles[k].base >= target) owner[k] = std::max(owner[k], r);
    }
  }
  m->shard_mode = mode;
  tick("ownership");

  std::vector<int64_t> local_of(T, -1);
  m->addr.assign(nP, -1);
  m->row_lo = -1; m->row_hi = 0; m->n_local = 0;
  for (size_t k = 0; k < T; ++k) {
    if (owner[k] != rank) continue;
    if (m->row_lo < 0) m->row_lo = tiles[k].base;
    m->row_hi = tiles[k].base + tlen[k];
    local_of[k] = m->n_local;
    m->tile_prefix.push_back(tiles[k].P);
    m->tile_base.push_back(m->n_local);
    m->addr[tiles[k].P] = m->n_local;
    m->n_local += tlen[k];
    m->max_tile_len = std::max<int>(m->max_tile_len, (int)tlen[k]);
  }
  if (m->row_lo < 0) {
    // a rank without tiles (more ranks than tiles or cells): an empty range placed where the next owner's rows begin
    m->row_lo = m->row_hi = m->N;
    for (size_t k = 0; k < T; ++k)
      if (owner[k] > rank) { m->row_lo = m->row_hi = tiles[k].base; break; }
  }

  tick("local tiles, addr");
  std::vector<int64_t> xcd_scratch((size_t)1 << p, -1);
  xcd_order(m, p, m->tile_prefix, m->tile_base, xcd_scratch);
  tick("xcd_order");

  m->single_prefix = m->tile_prefix; m->single_base = m->tile_base;

  m->pack_src.clear(); m->pack_dst.clear(); m->pack_len.clear(); m->n_send = 0;
  m->packed = 0;
  if (nranks > 1) {
    const int nn = count_nn_hops(m);
    std::vector<uint8_t> need(nP);
    std::vector<int64_t> nloc(nranks, 0);
    // offset of every tile inside its OWNER's vector (owned tiles are stored compactly in natural order)
    std::vector<int64_t> loc_all(T, 0);
    for (size_t k = 0; k < T; ++k) { loc_all[k] = nloc[owner[k]]; nloc[owner[k]] += tlen[k]; }
    // cells: the tiles a peer needs come in runs that are contiguous in the owner's vector (at L=28, 8 ranks: ~50 runs of
    // ~80 000 rows per rank, 16x longer at L=32).  Such runs travel straight from psi -- one send per run, matched in order by
    // one receive per run -- and the pack kernel and its send buffer disappear (round 3: 0.16-0.39 ms per step and rank at
    // L=32).  Only plans whose runs are short on average (< 2048 rows) keep the packed form; SD_SHARD_PACK=1 / 0 forces / forbids it.
    std::vector<sd_slab> d_recv, d_send;          // the direct form, built beside the packed one; chosen after the loop
    int64_t n_runs_all = 0, n_rows_all = 0;
    for (int q = 0; q < nranks; ++q) {
      std::fill(need.begin(), need.end(), 0);
      collect_needs(m, tiles, owner, q, nn, need);
      int64_t halo_off = nloc[q];
      if (mode == 0) {
        // index ranges: owned rows are globally contiguous, so what travels is a few contiguous slabs of psi itself
        sd_slab cur{-1, 0, 0, 0};
        int64_t cur_end_global = -1;
        auto flush = [&]() { if (cur.count && q == rank) m->recv_slabs.push_back(cur); cur.count = 0; };
        for (size_t k = 0; k < T; ++k) {
          if (!need[tiles[k].P] || owner[k] == q) continue;
          const int own = owner[k];
          const int64_t len = tlen[k];
          if (q == rank) {
            m->addr[tiles[k].P] = halo_off;
            if (cur.count > 0 && cur.peer == own && cur_end_global == tiles[k].base) cur.count += len;
            else { flush(); cur = {own, halo_off, len, tiles[k].base}; }
            cur_end_global = tiles[k].base + len;
          } else if (own == rank) {
            const int64_t loc = local_of[k];
            if (!m->send_slabs.empty() && m->send_slabs.back().peer == q &&
                m->send_slabs.back().local_offset + m->send_slabs.back().count == loc)
              m->send_slabs.back().count += len;
            else m->send_slabs.push_back({q, loc, len, tiles[k].base});
          }
          halo_off += len;
        }
        flush();
      } else {
        // cells: the tiles a peer needs are scattered over the owner's rows, so the owner packs them (natural order) into
        // a send buffer and exactly one message travels per (owner, receiver) pair
        for (int o = 0; o < nranks; ++o) {
          if (o == q) continue;
          const int64_t start = halo_off, sstart = m->n_send;
          int64_t count = 0;
          int64_t run_loc = -1, run_halo = 0, run_cnt = 0;       // current run: offset at the owner, offset at the receiver, rows
          auto flush_run = [&]() {
            if (run_cnt <= 0) return;
            ++n_runs_all; n_rows_all += run_cnt;
            if (q == rank) d_recv.push_back({o, run_halo, run_cnt, -1});
            if (o == rank) d_send.push_back({q, run_loc, run_cnt, -1});
            run_cnt = 0;
          };
          for (size_t k = 0; k < T; ++k) {
            if (owner[k] != o || !need[tiles[k].P]) continue;
            if (q == rank) m->addr[tiles[k].P] = halo_off;
            if (o == rank) { m->pack_src.push_back(local_of[k]); m->pack_dst.push_back(m->n_send); m->pack_len.push_back((int32_t)tlen[k]); m->n_send += tlen[k]; }
            if (run_cnt > 0 && loc_all[k] == run_loc + run_cnt) run_cnt += tlen[k];
            else { flush_run(); run_loc = loc_all[k]; run_halo = halo_off; run_cnt = tlen[k]; }
            halo_off += tlen[k]; count += tlen[k];
          }
          flush_run();
          if (count > 0) {
            if (q == rank) m->recv_slabs.push_back({o, start, count, -1});
            if (o == rank) m->send_slabs.push_back({q, sstart, count, -1});
          }
        }
      }
      if (q == rank) m->n_halo = halo_off - m->n_local;
    }
    if (mode == 1) {
      const char *fp = getenv("SD_SHARD_PACK");         // unset: by run length; 1: always pack; 0: never (tests)
      const bool direct = fp ? atoi(fp) == 0 : (n_runs_all > 0 && n_rows_all / n_runs_all >= 2048);
      if (direct) {          // the same on every rank: the statistics cover all (owner, receiver) pairs
        m->recv_slabs.swap(d_recv); m->send_slabs.swap(d_send);
        m->pack_src.clear(); m->pack_dst.clear(); m->pack_len.clear(); m->n_send = 0;
      } else m->packed = 1;
    }
  }
  // Launch segments of the single tiles: (interior | boundary) x (length class).  Interior tiles (every hop partner owned)
  // come first: they can run while the halo exchange is still in flight.  Inside each part the tiles are split by the
  // smallest workgroup (64..1024 threads x 4 rows) that covers them, one launch per class: a 220-row tile in a 256-thread
  // workgroup leaves three waves idle and costs about as much as 300 extra rows; in a one-wave workgroup four times as
  // many such tiles are in flight per CU.  Each segment is XCD-ordered on its own (block index restarts per launch).
  // SD_LEN_CLASSES=0 keeps one class, =2 splits small plans too (tests).  Order only: results do not depend on it (except the order of partial sums).
  {
    const size_t ns = m->single_prefix.size();
    std::vector<uint8_t> boundary(ns, 0);
    if (nranks > 1 && m->n_halo > 0) {
      const int nn = count_nn_hops(m);
      auto remote = [&](uint32_t Q) { return Q < nP && m->addr[Q] >= m->n_local; };
      for (size_t k = 0; k < ns; ++k) {
        const uint32_t P = m->single_prefix[k];
        bool bd = false;
        if (nn > 0) {
          for (int b = 1; b <= p - 1 && !bd; ++b)
            if ((((P >> (b - 1)) ^ (P >> b)) & 1u) && remote(P ^ (3u << (b - 1)))) bd = true;
          if (!bd && p >= 1 && remote(P ^ (1u << (p - 1)))) bd = true;
        }
        for (size_t h = (size_t)nn; h < m->hop_i.size() && !bd; ++h) {
          const int i = m->hop_i[h], j = m->hop_j[h];
          uint32_t Q = P;
          if (i <= p) Q ^= 1u << (i - 1);
          if (j <= p) Q ^= 1u << (j - 1);
          if (Q != P && remote(Q)) bd = true;
        }
        boundary[k] = bd;
      }
    }
    bool split = ns >= 32768;   // measured: three launches lose at L=24 (4 096 tiles, +38 %) and L=26 (16 384, +6 %), win from L=28 (65 536, -10 %)
    if (const char *e = getenv("SD_LEN_CLASSES")) split = atoi(e) >= 2 || (split && atoi(e) != 0);   // 2: also for small plans
    int top = 0;                                     // class of the longest tile: the only class when not splitting
    while (top < SD_N_LEN_CLASS - 1 && (64 << top) * 4 < m->max_tile_len) ++top;
    // LS > 12 (SD_SUFFIX_BITS experiments): no packed partner table; the binomial form of the kernel is built for the two
    // largest workgroups only, one class for all tiles
    if (m->LS > 12) { split = false; top = std::max(top, 3); }
    auto cls_of = [&](uint32_t P) {
      if (!split) return top;
      const int64_t len = B(m, m->LS, m->nup - __builtin_popcount(P));
      int c = 0;
      while (c < top && (64 << c) * 4 < len) ++c;
      return c;
    };
    std::vector<std::vector<uint32_t>> sp(2 * SD_N_LEN_CLASS);
    std::vector<std::vector<int64_t>> sb(2 * SD_N_LEN_CLASS);
    // Short tiles (<= 16 rows) of an unsharded plan go to k_apply_short when they are many: a workgroup per one-row or twelve-row tile
    // is all set-up (L=36, nup=9: 2.0 M of 2.6 M tiles hold 11 % of the rows and took 70 % of the apply: 4.13 -> 2.36 ms).  Many = at least
    // 8192 and a fifth of all tiles (half filling at L=32: 40 000 of 1 M, not worth two more launches).  SD_SHORT_TILES=0 keeps them in their
    // length class; =1 moves them whatever their number.
    std::vector<uint32_t> short_p; std::vector<int64_t> short_b;
    bool use_short = false;
    if (nranks == 1 && m->LS <= 12 && p >= 1) {
      size_t n_sh = 0;
      for (size_t k = 0; k < ns; ++k) if (B(m, m->LS, m->nup - __builtin_popcount(m->single_prefix[k])) <= 16) ++n_sh;
      use_short = n_sh >= 8192 && 5 * n_sh >= ns;
      if (const char *e = getenv("SD_SHORT_TILES")) use_short = atoi(e) != 0 && n_sh > 0;
    }
    for (size_t k = 0; k < ns; ++k) {
      if (use_short && B(m, m->LS, m->nup - __builtin_popcount(m->single_prefix[k])) <= 16) {
        short_p.push_back(m->single_prefix[k]); short_b.push_back(m->single_base[k]);
        continue;
      }
      // longest class first: the short-tile launches fill the tail of the long one
      const int sgi = boundary[k] * SD_N_LEN_CLASS + (SD_N_LEN_CLASS - 1 - cls_of(m->single_prefix[k]));
      sp[sgi].push_back(m->single_prefix[k]); sb[sgi].push_back(m->single_base[k]);
    }
    m->single_prefix.clear(); m->single_base.clear();
    for (int sgi = 0; sgi < 2 * SD_N_LEN_CLASS; ++sgi) {
      m->seg_off[sgi] = (int)m->single_prefix.size();
      m->seg_cls[sgi] = SD_N_LEN_CLASS - 1 - sgi % SD_N_LEN_CLASS;
      if (split || (nranks > 1 && m->n_halo > 0)) xcd_order(m, p, sp[sgi], sb[sgi], xcd_scratch);   // else: already in XCD order
      m->single_prefix.insert(m->single_prefix.end(), sp[sgi].begin(), sp[sgi].end());
      m->single_base.insert(m->single_base.end(), sb[sgi].begin(), sb[sgi].end());
    }
    m->seg_off[2 * SD_N_LEN_CLASS] = (int)m->single_prefix.size();
    m->n_interior = m->seg_off[SD_N_LEN_CLASS];
    // the short tiles follow the class segments (in natural order: consecutive tiles, consecutive rows)
    // ... those of 2..16 rows first, then the one-row tiles
    m->short_off = (int)m->single_prefix.size(); m->n_short = (int)short_p.size(); m->n_short_multi = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (size_t k = 0; k < short_p.size(); ++k) {
        const bool one = B(m, m->LS, m->nup - __builtin_popcount(short_p[k])) == 1;
        if (one != (pass == 1)) continue;
        m->single_prefix.push_back(short_p[k]); m->single_base.push_back(short_b[k]);
        if (pass == 0) ++m->n_short_multi;
      }
  }
  tick("halo plan, length classes");
  m->tile_gbase.resize(m->tile_prefix.size());
  for (size_t k = 0; k < m->tile_prefix.size(); ++k) m->tile_gbase[k] = tile_base_global(m, m->tile_prefix[k]);
  tick("global bases");
  return SD_OK;
}

// uniform-exact diagonal: every partial sum k*q (k <= n) is representable
static bool exact_multiples(double q, int n) {
  for (int k = 1; k <= n; ++k)
    if (std::fma((double)k, q, -((double)k * q)) != 0.0) return false;
  return true;
}

template <class T>
static int up(sd_model *m, const std::vector<T> &v, const T **dst, std::string &err) {
  *dst = nullptr;
  size_t bytes = sizeof(T) * std::max<size_t>(v.size(), 1);
  void *d = nullptr;
  hipError_t e = hipMalloc(&d, bytes);
  if (e != hipSuccess) { err = std::string("hipMalloc: ") + hipGetErrorString(e); return SD_ENOMEM; }
  m->dev_allocs.push_back(d);
  if (!v.empty()) {
    e = hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { err = std::string("hipMemcpy: ") + hipGetErrorString(e); return SD_EHIP; }
  }
  *dst = (const T *)d;
  return SD_OK;
}

void sd_free_device_tables(sd_model *m) {
  for (void *d : m->dev_allocs) (void)hipFree(d);
  m->dev_allocs.clear();
  m->dev_ready = false;
}

int sd_upload_model(sd_model *m, std::string &err) {
  sd_free_device_tables(m);
  sd_dev_model &d = m->dm;
  memset(&d, 0, sizeof(d));
  d.L = m->L; d.nup = m->nup; d.p = m->p; d.LS = m->LS;
  d.n_hop = (int)m->hop_i.size(); d.n_zz = (int)m->zz_i.size();
  d.N = m->N; d.n_local = m->n_local; d.row_lo = m->row_lo;
  d.full_ls = m->full_ls;
  d.fs_dbits = m->fs_dbits;
  for (int q = 0; q < SD_FS_MAX_RANKS; ++q) { d.fs_halo_off[q] = m->fs_halo_off[q]; d.fs_peer_lo[q] = m->fs_peer_lo[q]; }
  d.nn_hops = (m->p >= 0 || m->full_ls > 0 || m->nup >= 0) ? count_nn_hops(m) : 0;     // (per-row path of a sector: closed-form partner index of the chain bonds)
  d.field_zero = 1;
  for (double h : m->field) if (h != 0.0) d.field_zero = 0;
  // closed-form diagonal when it is bit-identical to the reference's sequential sum
  d.diag_mode = 0; d.diag_q = 0.0; d.n_zz_nn = 0;
  if (d.field_zero) {
    bool uniform = true;
    for (double J : m->zz_J) if (J != m->zz_J[0]) uniform = false;
    if (m->zz_J.empty()) { d.diag_mode = 1; d.diag_q = 0.0; }
    else if (uniform) {
      double q = (m->zz_J[0] * 0.5) * 0.5;
      if (exact_multiples(q, (int)m->zz_J.size())) { d.diag_mode = 1; d.diag_q = q; }
    }
  }
  if (m->L >= 2 && (int)m->zz_i.size() >= m->L - 1) {     // leading zz bonds = the chain in order: anti-parallel bits in one xor
    bool nn = true;
    for (int k = 0; k < m->L - 1; ++k)
      if (m->zz_i[k] != k + 1 || m->zz_j[k] != k + 2) nn = false;
    if (nn) d.n_zz_nn = m->L - 1;
  }
  if (getenv("SD_EXACT_DIAG")) d.diag_mode = 0;
  if (getenv("SD_DIAG_LITERAL")) d.diag_mode = 2;
  // list-order diagonal: (J * (+-0.5)) * (+-0.5) = +-(J/4) and h * (+-0.5) = +-(h/2) exactly (binary scaling), so each term
  // of the reference's sequential sum is a sign applied to a constant: same bits, a select and an add per term
  m->zz_q.resize(m->zz_J.size());
  for (size_t k = 0; k < m->zz_J.size(); ++k) m->zz_q[k] = (m->zz_J[k] * 0.5) * 0.5;
  m->field_h.resize(m->field.size());
  for (size_t k = 0; k < m->field.size(); ++k) m->field_h[k] = m->field[k] * 0.5;
  m->hop_pow2 = true;
  for (int k = 0; k < d.nn_hops; ++k) {
    int ex; const double mant = std::frexp(m->hop_J[k], &ex);
    if (!(m->hop_J[k] == 0.0 || std::fabs(mant) == 0.5) || !std::isfinite(m->hop_J[k])) m->hop_pow2 = false;
  }
  if (getenv("SD_NO_FMA")) m->hop_pow2 = false;
  d.dbg = getenv("SD_DEBUG_SKIP") ? atoi(getenv("SD_DEBUG_SKIP")) : 0;
  int rc;
  if ((rc = up(m, m->hop_i, &d.hop_i, err))) return rc;
  if ((rc = up(m, m->hop_j, &d.hop_j, err))) return rc;
  if ((rc = up(m, m->hop_J, &d.hop_J, err))) return rc;
  if ((rc = up(m, m->zz_i, &d.zz_i, err))) return rc;
  if ((rc = up(m, m->zz_j, &d.zz_j, err))) return rc;
  if ((rc = up(m, m->zz_J, &d.zz_J, err))) return rc;
  if ((rc = up(m, m->field, &d.field, err))) return rc;
  if ((rc = up(m, m->zz_q, &d.zz_q, err))) return rc;
  if ((rc = up(m, m->field_h, &d.field_h, err))) return rc;
  if ((rc = up(m, m->binom, &d.binom, err))) return rc;
  d.n_tiles = (int)m->tile_prefix.size();
  if (m->p >= 0) {
    if ((rc = up(m, m->tile_prefix, &d.tile_prefix, err))) return rc;
    if ((rc = up(m, m->tile_base, &d.tile_base, err))) return rc;
    if ((rc = up(m, m->addr, &d.addr, err))) return rc;
    if ((rc = up(m, m->suf_states, &d.suf_states, err))) return rc;
    if ((rc = up(m, m->suf_off, &d.suf_off, err))) return rc;
    if ((rc = up(m, m->suf_rank, &d.suf_rank, err))) return rc;
    if (!m->suf_part.empty()) {
      if ((rc = up(m, m->suf_part, &d.suf_part, err))) return rc;
      if ((rc = up(m, m->suf_dg, &d.suf_dg, err))) return rc;
    } else { d.suf_part = nullptr; d.suf_dg = nullptr; }
    d.wrap_hop = m->suf_part.empty() ? -1 : m->wrap_hop; d.wrap_pb = m->wrap_pb;
    if (getenv("SD_NO_WRAP_IMAGE")) d.wrap_hop = -1;      // A/B: the gather form of the wrap bond
    if ((rc = up(m, m->tile_gbase, &d.tile_gbase, err))) return rc;
    d.n_pack = (int)m->pack_len.size();
    if ((rc = up(m, m->pack_src, &d.pack_src, err))) return rc;
    if ((rc = up(m, m->pack_dst, &d.pack_dst, err))) return rc;
    if ((rc = up(m, m->pack_len, &d.pack_len, err))) return rc;
    d.n_singles = (int)m->single_prefix.size();
    d.n_interior = m->n_interior; d.tile_off = 0;
    d.n_short = m->n_short; d.short_off = m->short_off; d.n_short_multi = m->n_short_multi;
    m->single_rec.resize(m->single_prefix.size());
    for (size_t k = 0; k < m->single_prefix.size(); ++k) {
      const uint32_t P = m->single_prefix[k];
      const int t2 = m->nup - __builtin_popcount(P);
      sd_tile_rec &r = m->single_rec[k];
      r.base = m->single_base[k]; r.prefix = P;
      r.len = (int32_t)B(m, m->LS, t2); r.nU = (int32_t)B(m, m->LS - 1, t2 - 1); r.suf_off = m->suf_off[t2];
      r.pad0 = r.pad1 = 0;
    }
    if ((rc = up(m, m->single_rec, &d.single_rec, err))) return rc;
    // far-bond list of every tile, resolved on the host: the kernel's waves load their p entries with one coalesced read that
    // depends on nothing but the block index, instead of gathering addr[P ^ bond] per lane after the tile record has arrived
    {
      const int p = m->p, LS = m->LS;
      // only where the kernel reads it (chain bonds present): with p near SD_MAX_PREFIX_BITS the table is tiles x p x 8 B on
      // host and device, far too much to spend on a model whose hops are all general bonds
      m->far_base.clear();
      m->far_base.shrink_to_fit();
      if (p >= 1 && d.nn_hops > 0) m->far_base.assign(m->single_prefix.size() * (size_t)p, -1);
      if (p >= 1 && d.nn_hops > 0)
        for (size_t k = 0; k < m->single_prefix.size(); ++k) {
          const uint32_t P = m->single_prefix[k];
          int64_t *fb = m->far_base.data() + k * (size_t)p;
          for (int b = 1; b <= p - 1; ++b)
            if (((P >> (b - 1)) ^ (P >> b)) & 1u) fb[b - 1] = m->addr[P ^ (3u << (b - 1))];
          const uint32_t bitp = (P >> (p - 1)) & 1u, Q = P ^ (1u << (p - 1));
          const int t2 = m->nup - __builtin_popcount(P), t2q = m->nup - __builtin_popcount(Q);
          if (t2q >= 0 && t2q <= LS) {
            const int64_t len = B(m, LS, t2), nU = B(m, LS - 1, t2 - 1), nUq = B(m, LS - 1, t2q - 1);
            const int64_t my_n = bitp ? len - nU : nU;
            if (my_n > 0) fb[p - 1] = m->addr[Q] + (bitp ? 0 : nUq);
          }
        }
      if ((rc = up(m, m->far_base, &d.far_base, err))) return rc;
    }
    if ((rc = up(m, m->single_prefix, &d.single_prefix, err))) return rc;
    if ((rc = up(m, m->single_base, &d.single_base, err))) return rc;
    // General-bond plan (sd_gbond): every hop beyond the leading chain bonds classified by where its sites lie, partner rows of
    // the suffix-suffix and the mixed bonds tabulated once -- so that second neighbours and long-range lists run as streams and LDS
    // reads like the chain bonds do, instead of a rank look-up and a gather per row and bond.  Not for the periodic chain's lone
    // wrap bond, which keeps its own form (suf_part word 3), nor without the packed tables.  SD_GEN_PLAN=0: the per-row form.
    m->gen.clear(); m->gen_ss_part.clear(); m->mix_part.clear();
    d.n_gen = 0; d.n_gen_mixed = 0; d.n_suf_rows = (int)m->suf_states.size();
    d.gen = nullptr; d.gen_ss_part = nullptr; d.mix_part = nullptr;
    {
      const int p = m->p, LS = m->LS, nn = d.nn_hops;
      const char *ge = getenv("SD_GEN_PLAN");
      const bool wrap_only = d.n_hop == nn + 1 && d.wrap_hop == nn && !(ge && atoi(ge) == 2);     // (2: A/B of the plan on the periodic chain)
      if (!m->suf_part.empty() && d.n_hop > nn && !wrap_only && !(ge && atoi(ge) == 0)) {
        const size_t nsr = m->suf_states.size();
        int n_ss = 0;
        std::vector<int> mix_site(LS, 0);
        for (int h = nn; h < d.n_hop; ++h) {
          const int bi = m->hop_i[h] - 1, bj = m->hop_j[h] - 1;
          sd_gbond g{-1, 0u, 0, 0, m->hop_J[h]};
          if (bi != bj) {
            const bool ip = bi < p, jp = bj < p;
            if (ip && jp) { g.kind = 0; g.pmask = (1u << bi) | (1u << bj); }
            else if (!ip && !jp) { g.kind = 1; g.pmask = (1u << (bi - p)) | (1u << (bj - p)); g.slot = n_ss++; }   // (pmask: the two SUFFIX bits, host use only)
            else { g.kind = 2; g.pb = ip ? bi : bj; g.pmask = 1u << g.pb; g.slot = (ip ? bj : bi) - p; mix_site[g.slot] = 1; ++d.n_gen_mixed; }
          }
          m->gen.push_back(g);
        }
        const int n_chunks = (n_ss + 11) / 12;
        m->gen_ss_part.assign((size_t)std::max(n_chunks, 1) * nsr * 4, 0u);
        for (const sd_gbond &g : m->gen)
          if (g.kind == 1) {
            const int chunk = g.slot / 12, f = g.slot % 12;
            for (size_t k = 0; k < nsr; ++k) {
              const uint32_t sg = m->suf_states[k];
              if (__builtin_popcount(sg & g.pmask) != 1) continue;
              const uint32_t partner = (uint32_t)m->suf_rank[sg ^ g.pmask] + 1u;              // same sector: same popcount
              m->gen_ss_part[((size_t)chunk * nsr + k) * 4 + (size_t)(f / 3)] |= partner << (10 * (f % 3));
            }
          }
        m->mix_part.assign((size_t)LS * nsr, 0);
        for (int sb = 0; sb < LS; ++sb)
          if (mix_site[sb])
            for (size_t k = 0; k < nsr; ++k) {
              const uint32_t sg = m->suf_states[k];
              m->mix_part[(size_t)sb * nsr + k] = (uint16_t)(((uint32_t)m->suf_rank[sg ^ (1u << sb)] + 1u) | (((sg >> sb) & 1u) << 15));
            }
        for (sd_gbond &g : m->gen) if (g.kind == 1) g.pmask = 0u;      // device side: unused
        if ((rc = up(m, m->gen, &d.gen, err))) return rc;
        if ((rc = up(m, m->gen_ss_part, &d.gen_ss_part, err))) return rc;
        if ((rc = up(m, m->mix_part, &d.mix_part, err))) return rc;
        d.n_gen = (int)m->gen.size();
        d.wrap_hop = -1;                                   // the wrap bond, if any, is one of the plan's mixed bonds
      }
    }
  }
  // General couplings (no exact shortcut for the diagonal): the reference's sequential sum of up to 2L-1 rounded terms per
  // row is evaluated once per model and read back as 8 B/row by every apply -- the same bits, as fast as the hand-tuned
  // in-kernel sum for the chain terms alone and 10 % faster with field terms (profiles/couplings_bench.py) -- and, above all,
  // the in-kernel evaluation for all rows of a thread at once cost the apply kernel 19 VGPRs, i.e. one wave per SIMD, on
  // EVERY path (KPM step at L=30: 3.03 -> 3.39 ms).  SD_DIAG_CACHE=0 evaluates inside the kernel (prefix part per thread,
  // the rest per row); a failed allocation (the cache is half a ComplexF64 vector) does the same.
  const char *dc_env = getenv("SD_DIAG_CACHE");
  if (m->p >= 0 && d.diag_mode == 0 && d.n_local > 0 && d.n_tiles > 0 && !(dc_env && atoi(dc_env) == 0)) {
    void *dc = nullptr;
    if (hipMalloc(&dc, sizeof(double) * (size_t)d.n_local) == hipSuccess) {
      m->dev_allocs.push_back(dc);
      if (sd_k_build_diag(d, (double *)dc) == SD_OK) d.diag_cache = (const double *)dc;
    } else {
      (void)hipGetLastError();
    }
  }
  // does k_apply_tiled need the rows' suffix configurations?  Not for the open chain with a cached diagonal or the uniform
  // chain form of it: partners, the suffix's anti-parallel count and its first site all come from the packed table.
  {
    const bool chain_diag = d.diag_mode == 1 && (d.n_zz == 0 || (d.n_zz_nn == m->L - 1 && d.n_zz == d.n_zz_nn));
    d.need_sig = (d.n_hop > d.nn_hops) || !(d.diag_cache || chain_diag) || !d.suf_part;
    d.need_sig_gen = !(d.diag_cache || chain_diag);
  }
  m->dev_ready = true;
  return SD_OK;
}
