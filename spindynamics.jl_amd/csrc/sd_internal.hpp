// Internal declarations of libspindyn (not part of the C ABI).
//
// Basis layout (what every kernel relies on).  The reference enumerates a
// fixed-nup sector with Combinatorics.combinations(1:L, nup) (src/Basis.jl:41),
// i.e. lexicographically over sorted site lists, site i <-> bit i-1.  That
// order is a binary tree over the sites 1,2,...,L with the "site up" child
// first, so
//     idx0(s) = sum over sites k with bit_k = 0 and r_k >= 1 of C(L-k, r_k-1),
//     r_k = nup - (#up among sites < k).
// We split the sites into a PREFIX (sites 1..p, low bits) and a SUFFIX (sites
// p+1..L, LS = L-p high bits).  All rows sharing a prefix configuration P
// (j = popcount(P) ups) are contiguous -- a TILE -- and inside a tile they are
// ordered exactly like the sector (LS, nup-j).  Hence
//     state(row) = P | suf_states[nup-j][row - base(P)] << p
//     idx0(s)    = base(P) + suf_rank[s >> p].
// A hop on a bond inside the prefix maps a whole tile onto another whole tile
// (same internal order); the bond straddling prefix/suffix maps a contiguous
// half of a tile onto a contiguous half of another tile; bonds inside the
// suffix stay inside the tile (served from LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/spindyn.h"

#define SD_MAX_L 63
#define SD_MAX_BONDS 4096  // bond lists are uploaded to device memory; this bounds host validation only
#define SD_MAX_PREFIX_BITS 26
#define SD_FS_MAX_RANKS 8   // full-basis sharding: ranks = top index bits, nranks in {2, 4, 8}

// everything a workgroup needs to start on a tile, fetched with one scalar load
struct sd_tile_rec {
  int64_t base;      // offset of the tile's first row in the local vector
  uint32_t prefix;   // prefix configuration P
  int32_t len;       // rows = C(LS, t'), t' = nup - popcount(P)
  int32_t nU;        // rows whose first suffix site is up = C(LS-1, t'-1)
  int32_t suf_off;   // offset of sector (LS, t') in suf_states
  int32_t pad0, pad1;
};

// A hop of the list beyond the leading chain bonds (the periodic chain's wrap, second neighbours, long-range lists:
// src/SpinModel.jl:44-46, 74-78), classified on the host by where its two sites lie (basis.cpp, build_general_plan):
//   kind 0, both in the prefix : tile P maps onto tile P ^ pmask at the same row offset -- a coalesced stream, like a chain bond;
//   kind 1, both in the suffix : the partner row is in this tile, its row + 1 is field `slot` of the packed table gen_ss_part;
//   kind 2, one in each        : every partner row lies in the ONE tile P ^ pmask (pmask = the prefix site), at the row mix_part
//                                names for suffix site `slot`; the kernel streams that tile into a second LDS image;
//   kind -1                    : i == j, never flips (the reference accepts such a bond and it does nothing).
struct sd_gbond {
  int32_t kind;
  uint32_t pmask;
  int32_t slot;
  int32_t pb;        // kind 2: the prefix site (0-based bit of P)
  double J;
};

struct sd_xfer_team;   // host threads of the staged host <-> device transfers (xfer.cpp)

struct sd_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  // scratch for block partial sums / reduced scalars
  double *d_partials = nullptr;
  size_t partials_cap = 0;  // doubles
  double *d_scalars = nullptr;  // 16 doubles, device
  double *h_scalars = nullptr;  // 16 doubles, pinned host
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // device work vectors of the recursion-level calls, kept between calls: a hipMalloc/hipFree pair costs 0.3-0.6 ms whatever
  // the size (profiles/alloc_cost.py), more than ten small-system steps.  sd_ctx_release_scratch / sd_ctx_destroy free them.
  std::vector<std::pair<void *, size_t>> pool_free;
  void *stage[2] = {nullptr, nullptr};   // device staging of the host-pointer operator calls (sd_apply, ...), kept between calls
  size_t stage_cap[2] = {0, 0};
  int kpm_doubling = 1;     // sd_ctx_set_kpm_doubling: two Chebyshev moments per apply (default) or the reference's one
  // pinned ring + copy team of the large host <-> device transfers (xfer.cpp), created on first use
  void *xfer_buf[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t xfer_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t xfer_chunk = 0;
  sd_xfer_team *xfer_team = nullptr;
  sd_apply_fn user_apply = nullptr;   // recursion-level operator supplied by the caller (sd_ctx_set_apply_callback), or null
  void *user_apply_data = nullptr;
  int64_t n_applies = 0;    // operator applications the recursion-level entries have queued on this context (sd_ctx_apply_count)
  int gs_blocked = 1;       // sd_ctx_set_gs_blocked: lanczos_groundstate re-orthogonalises in blocks of 8 columns (default) or column by column as the reference
  int q_batch = 1;          // sd_ctx_set_q_batch: the momenta of S(q,w) share the launches of their recursions at launch-bound sizes
  int kpm_pair_q = 1;       // sd_ctx_set_kpm_pair_q: for a real psi0 compute S(q,w) once per pair (q, 2pi - q) and copy the row
};

// Device-side view of a model, passed by value to kernels.
struct sd_dev_model {
  int L, nup;            // nup < 0: full basis
  int p, LS;             // prefix / suffix site counts (tiled path); p = -1 when untiled
  int n_hop, n_zz;
  int full_ls;           // full 2^L basis, tiled path: a tile is 2^full_ls consecutive rows (0: not in use)
  int fs_dbits;          // full basis sharded by its top fs_dbits index bits (0: unsharded); rank = those bits
  int64_t fs_halo_off[SD_FS_MAX_RANKS];   // per peer rank: element offset of its slab in the halo buffer, -1: nothing imported from it
  int64_t fs_peer_lo[SD_FS_MAX_RANKS];    // per peer rank: first row (of the peer's local vector) held by that slab
  int nn_hops;           // leading hops that are exactly (1,2),(2,3),...,(L-1,L) in order (L-1 or 0)
  int diag_mode;         // 0 list order with exact term magnitudes (select + add per term), 1 uniform closed form, 2 literal reference loop (SD_DIAG_LITERAL)
  double diag_q;         // Jz/4 for diag_mode 1
  int n_zz_nn;           // leading zz bonds that are the NN chain in order (L-1 or 0), for diag_mode 1
  int field_zero;
  int dbg;               // timing-only ablation bits from env SD_DEBUG_SKIP (1 prefix bonds, 2 straddle, 4 suffix bonds)
  int64_t N;             // global dimension
  int64_t n_local;       // rows owned by this shard (== N unsharded)
  int64_t row_lo;        // first owned global row
  const int *hop_i, *hop_j;      // 1-based sites
  const double *hop_J;
  const int *zz_i, *zz_j;
  const double *zz_J;
  const double *field;
  const double *zz_q;            // (zz_J * 0.5) * 0.5: the exact magnitude of every zz term (diag_mode 0)
  const double *field_h;         // field * 0.5: the exact magnitude of every field term
  const double *diag_cache;      // list-order diagonal of every local row, computed once (tiled plans without an exact shortcut), or null
  const int64_t *binom;  // (SD_MAX_L+1) x (SD_MAX_L+1) row-major: C(n,k)
  // tiled path tables
  int n_tiles;
  const uint32_t *tile_prefix;   // per local tile (processing order)
  const int64_t *tile_base;      // per local tile: offset of its first row in the local vector
  const int64_t *tile_gbase;     // per local tile: GLOBAL basis index of its first row
  int n_pack;                    // cell-sharded plans: tiles this rank packs into its send buffer
  const int64_t *pack_src, *pack_dst;
  const int32_t *pack_len;
  const int64_t *addr;           // 2^p entries: offset of tile P in [local | halo], -1 if unavailable
  const uint16_t *suf_states;    // concatenated sectors (LS, t'), t' = 0..LS
  const int32_t *suf_off;        // LS+2 offsets into suf_states
  const uint16_t *suf_rank;      // 2^LS entries: rank of sigma inside its sector
  // Packed partner table of the suffix bonds (LS <= 12), one 16-byte entry per row of every suffix sector, indexed like
  // suf_states: word w holds three 10-bit fields, field 3*w + j (bits 10j..10j+9) = 1 + (row of the hop partner on suffix bond
  // a = 3*w + j + 1 inside the tile), 0 when the bond cannot flip.  Word 3 also carries, above its two bond fields, the number of
  // anti-parallel neighbour pairs INSIDE the suffix (bits 20..23) and the first suffix site (bit 30): with them the uniform-zz
  // diagonal needs no configuration at all.  Null for LS > 12 (k_apply_tiled then computes partners from binomials).
  const uint32_t *suf_part;
  const uint8_t *suf_dg;         // per row of every suffix sector: anti-parallel pairs inside the suffix (bits 0..3), first suffix site (bit 4); with suf_part
  // First general bond of the hop list when it joins a prefix site and a suffix site (the periodic chain's (L, 1)): every partner
  // row lies in the ONE tile P ^ (1 << wrap_pb).  suf_part's word 3 then holds, in bits 20..29, 1 + the partner's row inside that
  // tile (suffix sector t' + 1 when the row's suffix site is down, t' - 1 when it is up) and in bit 30 the row's suffix site; the
  // kernel streams the partner tile into a second LDS image and reads it there.  wrap_hop = index of that hop, or -1.
  int wrap_hop, wrap_pb;
  int need_sig;                  // the kernel must load the rows' suffix configurations (general bonds, or a diagonal that is not the uniform chain form / cached)
  const int64_t *far_base;       // per processed tile (single_rec order) p entries: base of the partner tile for prefix bond b = lane+1 and, in entry p-1, of the straddling bond's partner rows; -1: no hop
  int n_singles;
  int n_interior;                // sharded plans: the first n_interior single tiles read no halo (their partners are all owned)
  int tile_off;                  // first tile of this launch (lets the interior / boundary parts run as separate launches)
  const uint32_t *single_prefix; // the local tiles in launch order (k_apply_tiled)
  const int64_t *single_base;
  const sd_tile_rec *single_rec;
  unsigned long long *stamps;    // diagnostic builds only (sd_debug_phase_profile): 8 s_memtime stamps per tile, else null
  // General bonds of a tiled plan with the packed tables (LS <= 12), resolved on the host: n_gen = n_hop - nn_hops entries in list
  // order, or 0 (no plan: k_apply_tiled then looks every general partner up per row).  gen_ss_part: per chunk of 12 suffix-suffix
  // bonds one 16-byte entry per row of every suffix sector (chunk-major), 10-bit fields like suf_part.  mix_part: per suffix
  // site s one uint16 per row of every suffix sector -- bits 0..9 = 1 + rank of (sigma ^ bit s) inside ITS sector, bit 15 = bit s
  // of sigma.  n_suf_rows = rows of all suffix sectors together (2^LS).
  int n_gen, n_gen_mixed;
  int n_suf_rows;
  int need_sig_gen;              // the GEN form of the kernel needs the suffix configurations only for a diagonal without shortcut or cache
  const sd_gbond *gen;
  const uint32_t *gen_ss_part;
  const uint16_t *mix_part;
  // Short tiles (at most 16 rows: suffix fillings 0, 1, LS-1, LS -- most of the tiles of a dilute sector) of an unsharded plan, taken
  // out of the length classes: single_rec[short_off .. short_off + n_short) run through k_apply_short, sixteen lanes per tile, one row
  // per lane, partner rows by the closed form of the combinadic order instead of a workgroup per tile.  n_short = 0: none.
  int n_short, short_off;
  int n_short_multi;             // the first n_short_multi of them hold 2..16 rows (sixteen lanes per tile), the rest one row (one lane per tile)
};

#define SD_N_LEN_CLASS 5   // tile length classes: workgroups of 64, 128, 256, 512, 1024 threads (x 4 rows)

struct sd_model {
  sd_ctx *ctx = nullptr;  // may be null (host-only model)
  int L = 0, nup = -1;
  int64_t N = 0;
  std::vector<int> hop_i, hop_j, zz_i, zz_j;
  std::vector<double> hop_J, zz_J, field;
  std::vector<double> zz_q, field_h;   // device tables of the list-order diagonal (see sd_dev_model)
  std::vector<int64_t> binom;  // host copy
  // plan
  int p = -1, LS = 0;
  int full_ls = 0;             // full-basis tiled path (nup < 0, L >= 12): log2 rows per tile
  int fs_dbits = 0;            // full basis sharded by its top fs_dbits index bits
  int64_t fs_halo_off[SD_FS_MAX_RANKS] = {-1, -1, -1, -1, -1, -1, -1, -1};
  int64_t fs_peer_lo[SD_FS_MAX_RANKS] = {0};
  int rank = 0, nranks = 1;
  int64_t row_lo = 0, row_hi = 0, n_local = 0, n_halo = 0;
  std::vector<uint32_t> tile_prefix;  // local tiles
  std::vector<int64_t> tile_base;
  std::vector<int64_t> addr;
  std::vector<uint16_t> suf_states, suf_rank;
  std::vector<uint32_t> suf_part;   // packed suffix-bond partner table (see sd_dev_model), empty for LS > 12
  std::vector<uint8_t> suf_dg;
  int wrap_hop = -1, wrap_pb = 0;   // see sd_dev_model
  std::vector<sd_gbond> gen;        // general-bond plan (see sd_dev_model), empty: none
  int n_short = 0, short_off = 0, n_short_multi = 0;   // short tiles at the end of the single-tile lists (see sd_dev_model)
  std::vector<uint32_t> gen_ss_part;
  std::vector<uint16_t> mix_part;
  std::vector<int64_t> far_base;
  std::vector<int32_t> suf_off;
  std::vector<sd_slab> recv_slabs, send_slabs;
  int shard_mode_req = -1;       // -1 auto (env SD_SHARD_MODE), 0 index ranges, 1 popcount cells
  int shard_mode = 0;           // mode in effect
  int packed = 0;               // cell mode: 1 = the peers' tiles are gathered into a send buffer by the pack kernel (short runs); 0 = the
                                // send slabs index psi itself (contiguous runs; also index-range mode and the full basis)
  int64_t n_send = 0;           // elements of the packed send buffer (cell mode)
  std::vector<int64_t> pack_src, pack_dst, tile_gbase;
  std::vector<int32_t> pack_len;
  std::vector<uint32_t> single_prefix;
  int n_interior = 0;
  int seg_off[2 * SD_N_LEN_CLASS + 1] = {0};   // launch segments of the single tiles: (interior | boundary) x length class
  int seg_cls[2 * SD_N_LEN_CLASS] = {0};       // workgroup size of a segment's kernel = 64 << seg_cls
  std::vector<int64_t> single_base;
  std::vector<sd_tile_rec> single_rec;
  int max_tile_len = 0;        // longest OWNED tile
  int max_tile_len_all = 0;    // longest tile of the whole basis (a partner tile imported from a peer may be longer than any owned one)
  bool hop_pow2 = false;  // every NN hop amplitude is +-2^k (or 0): J*psi is exact, fma == mul+add
  // device copies
  sd_dev_model dm{};
  std::vector<void *> dev_allocs;
  bool dev_ready = false;
};

// ---- host basis helpers (basis.cpp) ----
int64_t sd_binom(int n, int k);
void sd_fill_binom(std::vector<int64_t> &tab);
uint64_t sd_unrank_host(const sd_model *m, int64_t idx0);
int64_t sd_rank_host(const sd_model *m, uint64_t s);  // -1 if not in the basis
int sd_build_plan(sd_model *m, int rank, int nranks, std::string &err);
int sd_upload_model(sd_model *m, std::string &err);
void sd_free_device_tables(sd_model *m);

// ---- kernels (launch wrappers; *.hip) ----
enum sd_epilogue {
  SD_EPI_PLAIN = 0,     // out = H psi
  SD_EPI_RESCALE = 1,   // out = (H psi - b psi)/a
  SD_EPI_CHEB = 2,      // out = 2*(H psi - b psi)/a - prev ; acc_vec += c*out
  SD_EPI_KPM = 3,       // out = 2*(H psi - b psi)/a - prev ; partial sums Re<phi|out>, |out|^2
  SD_EPI_DOT = 4,       // out = H psi ; partial sums <psi|out> (re, im)
  SD_EPI_RESCALE_DOT = 5,  // out = (H psi - b psi)/a ; partial sums Re<phi|out>, |out|^2
  SD_EPI_RECUR = 6,     // out = 2*(H psi - b psi)/a - prev            (Chebyshev term whose accumulation is deferred)
  SD_EPI_CHEB2 = 7      // out = 2*(H psi - b psi)/a - prev ; acc_vec += c0*psi ; acc_vec += c*out   (two terms, one pass over acc_vec)
};
struct sd_epi_args {
  double a = 1.0, b = 0.0;
  double c_re = 0.0, c_im = 0.0;
  double c0_re = 0.0, c0_im = 0.0;   // CHEB2: coefficient of the deferred previous term (= the apply's input vector)
  const void *prev = nullptr;   // phi_prev / v_prev
  void *accv = nullptr;         // psi_t (CHEB)
  const void *phi = nullptr;    // KPM reference vector
  int negate = 0;               // PLAIN / DOT: out = -(H psi); batched launches: bit k = vector k
  int stream_hint = 0;          // set by sd_launch_apply: bit 0 non-temporal stores of out, bit 1 non-temporal side streams (prev, phi, psi_t); env SD_STREAM_HINT
  double *sums_dst = nullptr;   // where the two reduced sums of a DOT / KPM / RESCALE_DOT epilogue go (device; null: ctx->d_scalars[0..1])
  const void *halo = nullptr;   // sharded plans: imported partner tiles (offsets >= n_local); null = halo follows psi's owned rows
  // Batched launch (k_apply_tiled, unsharded tiled plans): grid.y = batch vectors stored `bstride` ELEMENTS apart -- psi, out,
  // prev, phi and accv alike -- run through the same operator in one launch (the reference threads over the momenta of S(q,w),
  // src/KPM_Sqw.jl:218, src/LanczosSqw.jl:65; here the momenta's vectors share a launch where one vector cannot fill the chip).
  // The two sums of vector k go to sums_dst + k * sums_bstride.  Every vector sees exactly the arithmetic of a launch of its own.
  int batch = 1;
  int64_t bstride = 0;
  int64_t sums_bstride = 2;
  int no_reduce = 0;            // sum epilogues: leave the per-tile pairs in ctx->d_partials (n_singles per vector), the consumer sums them
};
// Launches the apply with the chosen epilogue.  When the epilogue produces
// partial sums, the reduced values land in ctx->d_scalars[0..1] (device).
int sd_launch_apply(sd_ctx *ctx, const sd_model *m, int dtype, void *out, const void *psi, int epi,
                    const sd_epi_args &ea, int part = 0);
// out <- epilogue(hpsi) where hpsi = H psi came from a caller's operator: the fused steps of the recursions as one elementwise pass
int sd_launch_epilogue_only(sd_ctx *ctx, int dtype, int64_t n, void *out, const void *hpsi, const void *psi, int epi,
                            const sd_epi_args &ea);
int sd_launch_observable(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi, int mode, double *out_host);
int sd_launch_pack(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi, void *sendbuf);
int sd_launch_fill_randn_local(sd_ctx *ctx, const sd_model *m, int dtype, void *x, uint64_t seed);
int sd_launch_spin_op(sd_ctx *ctx, const sd_model *m, int dtype, int site, int op, const void *psi, void *out);
int sd_launch_szq(sd_ctx *ctx, const sd_model *m, int dtype_in, const void *psi0, double q, void *phi);

// BLAS-1 style kernels on device vectors of `n` doubles (n = nc * N).
// Reductions write their result to ctx->d_scalars[slot..] (device memory) in a
// fixed, deterministic order; sd_read_scalars copies them to the host.
int sd_k_dot(sd_ctx *ctx, int nc, const double *x, const double *y, int64_t N, int slot);  // conj(x).y -> [slot]=re,[slot+1]=im
int sd_k_dotu(sd_ctx *ctx, const double *x, const double *y, int64_t N, int slot);          // complex sum x_i*y_i, NO conjugation -> [slot]=re,[slot+1]=im
int sd_k_nrm2sq(sd_ctx *ctx, const double *x, int64_t n, int slot);
int sd_k_dot_to(sd_ctx *ctx, int nc, const double *x, const double *y, int64_t N, double *dst_dev);    // -> dst_dev[0..1] (any device address)
int sd_k_nrm2sq_to(sd_ctx *ctx, const double *x, int64_t n, double *dst_dev);                       // -> dst_dev[0..1]
int sd_k_imag_count(sd_ctx *ctx, const double *xc, int64_t N, int slot);   // ComplexF64 elements with Im != 0 -> d_scalars[slot] (d_scalars[slot+1] = 0)
// three-term update on un-normalised Lanczos vectors (kernels_blas1.hip, k_lanczos_fold): t <- w, |w|^2 -> n2_out[0..1]
int sd_k_lanczos_fold(sd_ctx *ctx, double *t, const double *uc, const double *up, int64_t N, int form, const double *dot_dev,
                      const double *n2c_dev, const double *n2p_dev, double *store_alpha, double *store_bc, double *n2_out);
int sd_k_lanczos_fold_p(sd_ctx *ctx, double *t, const double *uc, const double *up, int64_t N, int batch, int64_t bstride, int form,
                        const double *dotp, int ndot, const double *n2cp, const double *n2pp, int nn2, double *store_alpha,
                        double *store_bc, int64_t store_stride, double *n2out, int nb);
int sd_k_lanczos_fold_blocks(int64_t N);     // blocks (= |w|^2 partial pairs per vector) of sd_k_lanczos_fold_p for N elements
int sd_k_lanczos_fold_scalars_p(sd_ctx *ctx, int batch, int form, const double *dotp, int ndot, const double *n2cp, int nn2,
                                double *store_alpha, double *store_bc, int64_t store_stride);
int sd_k_build_diag(const sd_dev_model &dm, double *out);   // out[local row] = diag_of(row) through the tile tables; synchronous
int sd_k_lanczos_fold_scalars(sd_ctx *ctx, int form, const double *dot_dev, const double *n2c_dev, double *store_alpha,
                              double *store_bc);
int sd_k_mgs_chain(sd_ctx *ctx, double *w, const double *V, int64_t ld, int ncols, int64_t N, int slot);   // MGS against V[:,0..ncols-2], then dot with V[:,ncols-1] -> d_scalars[slot]
int sd_k_bgs_chain(sd_ctx *ctx, double *w, const double *V, int64_t ld, int ncols, int64_t N, int slot);   // the same in blocks of 8 columns (classical inside a block): ~half the passes over memory
// lanczos_groundstate as passes that sum their producer's partial lists themselves (kernels_blas1.hip, "without reduction launches")
int sd_k_gs_blocks(int64_t N);
int sd_k_gs_chain(sd_ctx *ctx, double *w, const double *V, int64_t ld, int ncols, int64_t N, const double *y, double *scratch,
                  double *alpha_part, double *chk_dev);
int sd_k_gs_update(sd_ctx *ctx, double *w, const double *vj, const double *vjm1, int64_t N, const double *alpha_part,
                   const double *n2_prev, double *store_alpha, double *store_beta, double *n2_out, double *upd_scratch);
int sd_k_gs_scale(sd_ctx *ctx, double *vnext, const double *w, int64_t N, const double *n2_part, double *store_beta);
int sd_k_mdot(sd_ctx *ctx, const double *V, int64_t ld, int ncols, const double *y, int64_t N, double *out_host);   // out[c] = V[:,c].y (real)
int sd_read_scalars(sd_ctx *ctx, int slot, int count, double *out);
int sd_k_scale_div(sd_ctx *ctx, double *y, const double *x, int64_t n, double d);      // y = x / d
// w = w - (a*v + b*u)   (lanczos_extremal form, src/Lanczos.jl:59-61); u may be null (b ignored)
// w = (w - a*v) - b*u   (two roundings: src/Lanczos.jl:222-224, :127-129); u may be null
int sd_k_sub2(sd_ctx *ctx, double *w, const double *v, const double *u, int64_t n, double a, double b);
// fused update + squared norm into d_scalars[slot] (one pass instead of update, norm)
// w -= (ar + i ai) * v  for complex vectors (src/TimeEvolution/Krylov.jl:156)
// y += (ar + i ai) * x  complex accumulate (Krylov reconstruction :186-188)
int sd_k_ccombine(sd_ctx *ctx, double *y, const double *const *cols, int64_t N, int ncols, const double *cr,
                  const double *ci);   // y = sum_k (cr+i ci)[k] * cols[k], column order, bit-identical to ncols k_cacc passes on y = 0
// y = c0*x0 (+ c1*x1) complex  (Chebyshev start, src/TimeEvolution/Chebyshev.jl:96-102)
int sd_k_cheb_init(sd_ctx *ctx, double *y, const double *x0, const double *x1, int64_t N, double c0r, double c0i,
                   double c1r, double c1i, int have1);
// complex <- real promotion / copy
int sd_k_promote(sd_ctx *ctx, double *yc, const double *x, int nc_in, int64_t N);
// y[i] = sum_k V[i + N*k] * coef[k]  (real, ground-state reconstruction src/Lanczos.jl:170); coef on host
int sd_k_gemv_cols(sd_ctx *ctx, double *y, const double *V, int64_t N, int ncols, const double *coef_host);
int sd_k_fill_randn(sd_ctx *ctx, double *x, int64_t n, uint64_t seed, uint64_t first);
double sd_randn_host(uint64_t seed, uint64_t k);

// partials[0 .. 2n) -> dst[0..1] (null: ctx->d_scalars[0..1]) in a fixed order; the caller reserved 2n + 2*SD_RED_STAGE_BLOCKS
// doubles of ctx->d_partials (long lists are summed in two stages)
#define SD_RED_STAGE_BLOCKS 512
int sd_reduce_pairs(sd_ctx *ctx, int64_t n, double *dst);
// the same for `batch` lists of n pairs stored back to back (n <= 16384): list k -> dst + k * dstride
int sd_reduce_pairs_batched(sd_ctx *ctx, int64_t n, int batch, double *dst, int64_t dstride);
// ---- communicators (comm.cpp): all three are no-ops for a null communicator or an unsharded model ----
int sd_comm_nranks(const sd_comm *c);
int sd_comm_exchange_start(sd_ctx *ctx, sd_comm *c, const sd_model *m, int dtype, const void *src, void *halo);
int sd_comm_exchange_wait(sd_ctx *ctx, sd_comm *c, const sd_model *m);
int sd_comm_allreduce_dev(sd_ctx *ctx, sd_comm *c, double *vals_dev, int count);   // in place, ordered on ctx->stream

// host <-> device copies of the host-pointer entry points, complete on return (xfer.cpp: pinned ring + host thread team for
// large transfers); ordered after the work already queued on ctx->stream
int sd_xfer_h2d(sd_ctx *ctx, void *dev, const void *host, size_t bytes);
int sd_xfer_d2h(sd_ctx *ctx, void *host, const void *dev, size_t bytes);
void sd_xfer_release(sd_ctx *ctx);

// grows ctx->d_partials to at least `doubles` entries (scratch for per-workgroup partial sums)
int sd_ensure_partials(sd_ctx *ctx, size_t doubles);

// error helpers
int sd_set_err(sd_ctx *ctx, int code, const std::string &msg);
// work-vector pool of a context (see sd_ctx::pool_free): take a cached block of at least `bytes` (and at most twice that), or hipMalloc
int sd_pool_take(sd_ctx *ctx, size_t bytes, void **out, size_t *got);
void sd_pool_give(sd_ctx *ctx, void *p, size_t bytes);
void sd_pool_release(sd_ctx *ctx);
#define SD_HIP(ctx, call)                                                                       \
  do {                                                                                          \
    hipError_t e__ = (call);                                                                    \
    if (e__ != hipSuccess)                                                                      \
      return sd_set_err((ctx), SD_EHIP, std::string(#call) + ": " + hipGetErrorString(e__));   \
  } while (0)
