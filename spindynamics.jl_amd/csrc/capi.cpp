// C ABI: contexts, models, operator-level entry points (include/spindyn.h).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "sd_internal.hpp"

int sd_set_err(sd_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  return code;
}

int sd_pool_take(sd_ctx *ctx, size_t bytes, void **out, size_t *got) {
  if (bytes == 0) bytes = 8;
  int best = -1;
  for (int i = 0; i < (int)ctx->pool_free.size(); ++i) {
    const size_t sz = ctx->pool_free[i].second;
    if (sz >= bytes && sz <= 2 * bytes && (best < 0 || sz < ctx->pool_free[best].second)) best = i;
  }
  if (best >= 0) {
    *out = ctx->pool_free[best].first; *got = ctx->pool_free[best].second;
    ctx->pool_free.erase(ctx->pool_free.begin() + best);
    return SD_OK;
  }
  static const bool dbg = getenv("SD_POOL_DEBUG") != nullptr;     // stderr trace of every allocation the pool could not serve
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(out, bytes);
  bool retried = false;
  if (e != hipSuccess) {                       // make room: drop what the pool still holds, then try once more
    (void)hipGetLastError();
    sd_pool_release(ctx);
    e = hipMalloc(out, bytes);
    retried = true;
  }
  if (dbg) {
    size_t held = 0, fr = 0, tot = 0;
    for (auto &b : ctx->pool_free) held += b.second;
    (void)hipMemGetInfo(&fr, &tot);
    fprintf(stderr, "[sd pool] hipMalloc %.3f GB took %.1f ms%s; pool holds %zu blocks / %.2f GB; device free %.1f of %.1f GB\n",
            bytes / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
            retried ? " (after dropping the pool: first attempt failed)" : "", ctx->pool_free.size(), held / 1e9, fr / 1e9, tot / 1e9);
  }
  if (e != hipSuccess) { *out = nullptr; return sd_set_err(ctx, SD_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  *got = bytes;
  return SD_OK;
}

void sd_pool_give(sd_ctx *ctx, void *p, size_t bytes) {
  if (!p) return;
  static const size_t cap = [] {               // bytes the pool may hold (SD_POOL_MAX_GB, default 96; 0 disables the pool)
    const char *e = getenv("SD_POOL_MAX_GB");
    return (size_t)((e ? atof(e) : 96.0) * 1e9);
  }();
  if (bytes > cap) { (void)hipFree(p); return; }
  // every block is kept (a hipMalloc/hipFree pair costs ~0.1 ms even for a few KB: hipFree waits for the device -- more than
  // three Lanczos steps of a small system); the oldest blocks make room when the pool is full
  ctx->pool_free.emplace_back(p, bytes);
  size_t held = 0;
  for (auto &b : ctx->pool_free) held += b.second;
  while (!ctx->pool_free.empty() && (held > cap || ctx->pool_free.size() > 256)) {
    held -= ctx->pool_free.front().second;
    (void)hipFree(ctx->pool_free.front().first);
    ctx->pool_free.erase(ctx->pool_free.begin());
  }
}

void sd_pool_release(sd_ctx *ctx) {
  for (auto &b : ctx->pool_free) (void)hipFree(b.first);
  ctx->pool_free.clear();
}

int sd_ensure_partials(sd_ctx *ctx, size_t doubles) {
  if (ctx->partials_cap >= doubles) return SD_OK;
  if (ctx->d_partials) (void)hipFree(ctx->d_partials);   // hipFree waits for the kernels still using it
  ctx->d_partials = nullptr; ctx->partials_cap = 0;
  SD_HIP(ctx, hipMalloc((void **)&ctx->d_partials, doubles * sizeof(double)));
  ctx->partials_cap = doubles;
  return SD_OK;
}

namespace {
thread_local std::string g_err;  // errors raised without a context

int fail(sd_ctx *ctx, int code, const std::string &msg) {
  g_err = msg;
  return sd_set_err(ctx, code, msg);
}

struct TmpDev {
  sd_ctx *ctx;
  void *p = nullptr;
  explicit TmpDev(sd_ctx *c) : ctx(c) {}
  ~TmpDev() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) {
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) return sd_set_err(ctx, SD_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return SD_OK;
  }
};

// grow-only device staging buffers of a context: a solver that calls sd_apply in a loop pays hipMalloc/hipFree once
int stage_buf(sd_ctx *ctx, int which, size_t bytes, void **out) {
  if (ctx->stage_cap[which] < bytes) {
    if (ctx->stage[which]) (void)hipFree(ctx->stage[which]);
    ctx->stage[which] = nullptr; ctx->stage_cap[which] = 0;
    hipError_t e = hipMalloc(&ctx->stage[which], bytes ? bytes : 1);
    if (e != hipSuccess) return sd_set_err(ctx, SD_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    ctx->stage_cap[which] = bytes;
  }
  *out = ctx->stage[which];
  return SD_OK;
}

int check_apply_args(sd_ctx *ctx, const sd_model *m, int dtype, const void *out, const void *psi, int64_t n) {
  if (!ctx) return SD_EARG;
  if (!m) return sd_set_err(ctx, SD_EARG, "null model");
  if (!m->dev_ready) return sd_set_err(ctx, SD_EARG, "model has no device tables (created without a context)");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "dtype must be SD_F64 or SD_C128");
  if (!out || !psi) return sd_set_err(ctx, SD_EARG, "null vector");
  if (out == psi) return sd_set_err(ctx, SD_EARG, "out must not alias psi");
  SD_HIP(ctx, hipSetDevice(ctx->device));   // the caller's thread may have another device current (several contexts per process)
  return SD_OK;
}
}  // namespace

extern "C" {

const char *sd_version(void) { return "spindyn-mi355x 0.1 (gfx950)"; }

int sd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *sd_status_string(int status) {
  switch (status) {
    case SD_OK: return "ok";
    case SD_EARG: return "invalid argument";
    case SD_EDIM: return "dimension mismatch";
    case SD_EZERO: return "starting vector has zero norm";
    case SD_ENOMEM: return "out of memory";
    case SD_EHIP: return "HIP error";
    case SD_ENODEV: return "no HIP device";
    case SD_EINTERNAL: return "internal error";
    case SD_ECOMM: return "communication error";
    default: return "unknown status";
  }
}

int sd_ctx_create(int device, sd_ctx **out) {
  if (!out) return SD_EARG;
  *out = nullptr;
  int n = sd_device_count();
  if (n <= 0) return fail(nullptr, SD_ENODEV, "no HIP device: libspindyn has no CPU fallback");
  if (device < 0 || device >= n) return fail(nullptr, SD_EARG, "device index out of range");
  sd_ctx *c = new (std::nothrow) sd_ctx();
  if (!c) return SD_ENOMEM;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->own_stream) != hipSuccess) {
    delete c;
    return fail(nullptr, SD_EHIP, "hipSetDevice/hipStreamCreate failed");
  }
  c->stream = c->own_stream;
  if (hipMalloc((void **)&c->d_scalars, 16 * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void **)&c->h_scalars, 16 * sizeof(double)) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    sd_ctx_destroy(c);
    return fail(nullptr, SD_EHIP, "context allocation failed");
  }
  (void)hipMemset(c->d_scalars, 0, 16 * sizeof(double));
  if (const char *e = getenv("SD_Q_BATCH")) c->q_batch = atoi(e) ? 1 : 0;
  *out = c;
  return SD_OK;
}

void sd_ctx_destroy(sd_ctx *c) {
  if (!c) return;
  if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
  if (c->d_partials) (void)hipFree(c->d_partials);
  for (void *b : c->stage) if (b) (void)hipFree(b);
  sd_xfer_release(c);
  sd_pool_release(c);
  if (c->d_scalars) (void)hipFree(c->d_scalars);
  if (c->h_scalars) (void)hipHostFree(c->h_scalars);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int sd_ctx_set_stream(sd_ctx *ctx, void *hip_stream) {
  if (!ctx) return SD_EARG;
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  SD_HIP(ctx, hipSetDevice(ctx->device));
  return SD_OK;
}

int sd_ctx_set_kpm_doubling(sd_ctx *ctx, int on) {
  if (!ctx) return SD_EARG;
  ctx->kpm_doubling = on ? 1 : 0;
  return SD_OK;
}

int64_t sd_ctx_apply_count(const sd_ctx *ctx) { return ctx ? ctx->n_applies : -1; }

int sd_ctx_set_gs_blocked(sd_ctx *ctx, int on) {
  if (!ctx) return SD_EARG;
  ctx->gs_blocked = on ? 1 : 0;
  return SD_OK;
}

int sd_ctx_set_kpm_pair_q(sd_ctx *ctx, int on) {
  if (!ctx) return SD_EARG;
  ctx->kpm_pair_q = on ? 1 : 0;
  return SD_OK;
}

int sd_ctx_set_q_batch(sd_ctx *ctx, int on) {
  if (!ctx) return SD_EARG;
  ctx->q_batch = on ? 1 : 0;
  return SD_OK;
}

int sd_ctx_release_scratch(sd_ctx *ctx) {
  if (!ctx) return SD_EARG;
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 2; ++k) {
    if (ctx->stage[k]) (void)hipFree(ctx->stage[k]);
    ctx->stage[k] = nullptr; ctx->stage_cap[k] = 0;
  }
  if (ctx->d_partials) (void)hipFree(ctx->d_partials);
  ctx->d_partials = nullptr; ctx->partials_cap = 0;
  sd_xfer_release(ctx);
  sd_pool_release(ctx);
  return SD_OK;
}

int sd_ctx_synchronize(sd_ctx *ctx) {
  if (!ctx) return SD_EARG;
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SD_OK;
}

const char *sd_last_error(const sd_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

// Plan + device tables of a model.  The plan's host tables grow with the number of tiles (up to 2^26 prefixes x p entries): a
// failed host allocation must come back as a status, never unwind through the C ABI.
static int plan_and_upload(sd_model *m, int rank, int nranks, std::string &err) {
  try {
    int rc = sd_build_plan(m, rank, nranks, err);
    if (!rc && m->ctx) {
      if (hipSetDevice(m->ctx->device) != hipSuccess) { rc = SD_EHIP; err = "hipSetDevice failed"; }
      else rc = sd_upload_model(m, err);
    }
    return rc;
  } catch (const std::bad_alloc &) {
    err = "out of host memory while building the plan tables of this model";
    return SD_ENOMEM;
  } catch (const std::exception &e) {
    err = std::string("plan construction failed: ") + e.what();
    return SD_EINTERNAL;
  }
}

int sd_model_create(sd_ctx *ctx, int L, int nup, int n_hop, const int *hop_i, const int *hop_j, const double *hop_J,
                    int n_zz, const int *zz_i, const int *zz_j, const double *zz_J, const double *field,
                    sd_model **out) {
  if (!out) return SD_EARG;
  *out = nullptr;
  // src/Basis.jl:9-20 _validate_basis_args
  if (L < 1) return fail(ctx, SD_EARG, "L must be at least 1");
  if (L > SD_MAX_L) return fail(ctx, SD_EARG, "L must be at most 63 when using UInt64 basis states");
  if (nup < -1 || nup > L) return fail(ctx, SD_EARG, "nup must satisfy 0 <= nup <= L");
  if (n_hop < 0 || n_zz < 0 || n_hop > SD_MAX_BONDS || n_zz > SD_MAX_BONDS) return fail(ctx, SD_EARG, "bad bond count");
  if ((n_hop && (!hop_i || !hop_j || !hop_J)) || (n_zz && (!zz_i || !zz_j || !zz_J))) return fail(ctx, SD_EARG, "null bond list");
  for (int k = 0; k < n_hop; ++k)
    if (hop_i[k] < 1 || hop_i[k] > L || hop_j[k] < 1 || hop_j[k] > L)
      return fail(ctx, SD_EARG, "hopping bond site out of range");
  for (int k = 0; k < n_zz; ++k)
    if (zz_i[k] < 1 || zz_i[k] > L || zz_j[k] < 1 || zz_j[k] > L) return fail(ctx, SD_EARG, "zz bond site out of range");
  sd_model *m = new (std::nothrow) sd_model();
  if (!m) return SD_ENOMEM;
  m->ctx = ctx; m->L = L; m->nup = nup;
  // a hop (i, i, J) is accepted as the reference accepts it (build_model takes any pair; in apply_H! bit_i != bit_j is
  // never true for i == j, src/Hamiltonian.jl:248-252): the term does nothing and is dropped here, the order of the rest kept
  for (int k = 0; k < n_hop; ++k) {
    if (hop_i[k] == hop_j[k]) continue;
    m->hop_i.push_back(hop_i[k]); m->hop_j.push_back(hop_j[k]); m->hop_J.push_back(hop_J[k]);
  }
  m->zz_i.assign(zz_i, zz_i + n_zz); m->zz_j.assign(zz_j, zz_j + n_zz); m->zz_J.assign(zz_J, zz_J + n_zz);
  m->field.assign(L, 0.0);
  if (field) m->field.assign(field, field + L);
  sd_fill_binom(m->binom);
  m->N = nup < 0 ? ((int64_t)1 << L) : sd_binom(L, nup);
  std::string err;
  int rc = plan_and_upload(m, 0, 1, err);
  if (rc) { sd_model_destroy(m); return fail(ctx, rc, err); }
  *out = m;
  return SD_OK;
}

int sd_xxz_chain(sd_ctx *ctx, int L, double Jxy, double Jz, double hz, int nup, int boundary, sd_model **out) {
  if (!out) return SD_EARG;
  *out = nullptr;
  if (boundary != 0 && boundary != 1) return fail(ctx, SD_EARG, "boundary must be :open or :periodic");
  if (L < 1 || L > SD_MAX_L) return fail(ctx, SD_EARG, "L must satisfy 1 <= L <= 63");
  int hi[SD_MAX_L + 1], hj[SD_MAX_L + 1];
  double hJ[SD_MAX_L + 1], zJ[SD_MAX_L + 1], f[SD_MAX_L + 1];
  int nb = 0;
  for (int i = 1; i <= L - 1; ++i) { hi[nb] = i; hj[nb] = i + 1; hJ[nb] = Jxy / 2; zJ[nb] = Jz; ++nb; }   // src/SpinModel.jl:71-72
  if (boundary == 1 && L > 2) { hi[nb] = L; hj[nb] = 1; hJ[nb] = Jxy / 2; zJ[nb] = Jz; ++nb; }           // :74-78
  for (int i = 0; i < L; ++i) f[i] = hz;                                                              // :87
  return sd_model_create(ctx, L, nup, nb, hi, hj, hJ, nb, hi, hj, zJ, f, out);
}

void sd_model_destroy(sd_model *m) {
  if (!m) return;
  sd_free_device_tables(m);
  delete m;
}

int64_t sd_model_dim(const sd_model *m) { return m ? m->N : -1; }
int sd_model_L(const sd_model *m) { return m ? m->L : -1; }
int sd_model_nup(const sd_model *m) { return m ? m->nup : -2; }
int sd_ctx_set_apply_callback(sd_ctx *ctx, sd_apply_fn fn, void *user) {
  if (!ctx) return SD_EARG;
  ctx->user_apply = fn;
  ctx->user_apply_data = fn ? user : nullptr;
  return SD_OK;
}
int sd_model_path(const sd_model *m) { return !m ? 0 : m->p >= 0 ? 1 : m->full_ls > 0 ? 2 : 0; }

int sd_model_states(const sd_model *m, int64_t start, int64_t count, uint64_t *out) {
  if (!m || !out || start < 0 || count < 0 || start + count > m->N) return SD_EARG;
  for (int64_t i = 0; i < count; ++i) out[i] = sd_unrank_host(m, start + i);
  return SD_OK;
}

int sd_model_rank(const sd_model *m, const uint64_t *states, int64_t n, int64_t *idx) {
  if (!m || !states || !idx || n < 0) return SD_EARG;
  for (int64_t i = 0; i < n; ++i) idx[i] = sd_rank_host(m, states[i]);
  return SD_OK;
}

int sd_model_set_shard_mode(sd_model *m, int rank, int nranks, int mode) {
  if (!m || mode < -1 || mode > 1) return SD_EARG;
  m->shard_mode_req = mode;
  return sd_model_set_shard(m, rank, nranks);
}

int sd_model_set_shard(sd_model *m, int rank, int nranks) {
  if (!m) return SD_EARG;
  std::string err;
  int rc = plan_and_upload(m, rank, nranks, err);
  if (rc) {
    // The host tables may be partly those of the new (rank, nranks) while the device tables still describe the old plan:
    // no kernel may run on that mixture.  The model is left without a plan -- every later call that needs one fails with
    // SD_EARG ("model has no device tables") -- until a set_shard succeeds; the caller is expected to destroy it.
    sd_free_device_tables(m);
    m->dev_ready = false;
    m->n_local = 0; m->n_halo = 0; m->n_send = 0; m->n_interior = 0;
    m->tile_prefix.clear(); m->tile_base.clear();
    m->recv_slabs.clear(); m->send_slabs.clear();
    m->pack_len.clear();
    return fail(m->ctx, rc, err);
  }
  return SD_OK;
}

int sd_model_shard_info(const sd_model *m, sd_shard_info *out) {
  if (!m || !out) return SD_EARG;
  out->rank = m->rank; out->nranks = m->nranks;
  out->row_lo = m->row_lo; out->row_hi = m->row_hi;
  out->n_local = m->n_local; out->n_halo = m->n_halo;
  out->n_recv_slabs = (int64_t)m->recv_slabs.size(); out->n_send_slabs = (int64_t)m->send_slabs.size();
  out->mode = m->shard_mode; out->n_send = m->n_send; out->n_local_tiles = (int64_t)m->tile_prefix.size();
  out->n_pack = (int64_t)m->pack_len.size();
  out->n_interior_tiles = m->n_interior;
  out->n_interior_rows = 0;
  if (m->p >= 0)
    for (int k = 0; k < m->n_interior && k < (int)m->single_prefix.size(); ++k)
      out->n_interior_rows += sd_binom(m->LS, m->nup - __builtin_popcount(m->single_prefix[k]));
  out->packed = m->packed;
  return SD_OK;
}

int sd_model_shard_slabs(const sd_model *m, sd_slab *recv_out, sd_slab *send_out) {
  if (!m) return SD_EARG;
  if (recv_out) for (size_t i = 0; i < m->recv_slabs.size(); ++i) recv_out[i] = m->recv_slabs[i];
  if (send_out) for (size_t i = 0; i < m->send_slabs.size(); ++i) send_out[i] = m->send_slabs[i];
  return SD_OK;
}

int sd_model_local_tiles(const sd_model *m, int64_t *local_base, int64_t *global_base, int32_t *len) {
  if (!m || m->p < 0) return SD_EARG;
  for (size_t k = 0; k < m->tile_prefix.size(); ++k) {
    if (local_base) local_base[k] = m->tile_base[k];
    if (global_base) global_base[k] = m->tile_gbase.size() == m->tile_prefix.size() ? m->tile_gbase[k] : -1;
    if (len) len[k] = (int32_t)sd_binom(m->LS, m->nup - __builtin_popcount(m->tile_prefix[k]));
  }
  return SD_OK;
}

int sd_model_shard_pack_list(const sd_model *m, int64_t *src, int64_t *dst, int32_t *len) {
  if (!m) return SD_EARG;
  for (size_t k = 0; k < m->pack_len.size(); ++k) {
    if (src) src[k] = m->pack_src[k];
    if (dst) dst[k] = m->pack_dst[k];
    if (len) len[k] = m->pack_len[k];
  }
  return SD_OK;
}

int sd_shard_pack_dev(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_dev, void *sendbuf_dev) {
  if (!ctx) return SD_EARG;
  if (!m || !psi_dev) return sd_set_err(ctx, SD_EARG, "null argument");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "bad dtype");
  if (m->n_send > 0 && !sendbuf_dev) return sd_set_err(ctx, SD_EARG, "this shard needs a send buffer");
  return sd_launch_pack(ctx, m, dtype, psi_dev, sendbuf_dev);
}

int sd_fill_randn_local_dev(sd_ctx *ctx, const sd_model *m, int dtype, void *x_dev, uint64_t seed) {
  if (!ctx) return SD_EARG;
  if (!m || !x_dev) return sd_set_err(ctx, SD_EARG, "null argument");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "bad dtype");
  return sd_launch_fill_randn_local(ctx, m, dtype, x_dev, seed);
}

// ---- operator level ----

int sd_apply_dev(sd_ctx *ctx, const sd_model *m, int dtype, void *out, const void *psi, int64_t n) {
  int rc = check_apply_args(ctx, m, dtype, out, psi, n);
  if (rc) return rc;
  if (n != m->n_local) return sd_set_err(ctx, SD_EDIM, "vector length does not match the (local) basis dimension");
  sd_epi_args ea;
  return sd_launch_apply(ctx, m, dtype, out, psi, SD_EPI_PLAIN, ea);
}

int sd_kpm_step_sharded_dev(sd_ctx *ctx, const sd_model *m, void *v_next, const void *v_curr, const void *halo,
                            const void *v_prev, const void *phi, int64_t n_local, double a, double b, int first,
                            double *sums_out) {
  // one KPM recursion step on this shard (src/KPM_Sqw.jl:106-117): first != 0: v_next = H~ v_curr; else
  // v_next = 2 H~ v_curr - v_prev.  sums_out = local { Re<phi|v_next>, |v_next|^2 } (all-reduce them over the ranks);
  // phi == NULL: Re<v_curr|v_next> instead (moment doubling, no extra vector read).
  int rc = check_apply_args(ctx, m, SD_C128, v_next, v_curr, n_local);
  if (rc) return rc;
  if (n_local != m->n_local) return sd_set_err(ctx, SD_EDIM, "vector length does not match the local basis dimension");
  if (!sums_out || (!first && !v_prev)) return sd_set_err(ctx, SD_EARG, "null argument");   // phi may be null: <v_curr|v_next>
  if (m->n_halo > 0 && !halo) return sd_set_err(ctx, SD_EARG, "this shard needs a halo buffer");
  sd_epi_args ea; ea.a = a; ea.b = b; ea.prev = v_prev; ea.phi = phi; ea.halo = halo;
  rc = sd_launch_apply(ctx, m, SD_C128, v_next, v_curr, first ? SD_EPI_RESCALE_DOT : SD_EPI_KPM, ea, 0);
  if (rc) return rc;
  return sd_read_scalars(ctx, 0, 2, sums_out);
}

int sd_apply_sharded_dev(sd_ctx *ctx, const sd_model *m, int dtype, void *out, const void *psi, const void *halo,
                         int64_t n_local, int epilogue, double a, double b, double c_re, double c_im,
                         const void *phi_prev, void *psi_t, int part) {
  int rc = check_apply_args(ctx, m, dtype, out, psi, n_local);
  if (rc) return rc;
  if (n_local != m->n_local) return sd_set_err(ctx, SD_EDIM, "vector length does not match the local basis dimension");
  if (part < 0 || part > 2) return sd_set_err(ctx, SD_EARG, "part must be 0 (all), 1 (interior) or 2 (boundary)");
  if (m->n_halo > 0 && !halo && part != 1) return sd_set_err(ctx, SD_EARG, "this shard needs a halo buffer");
  sd_epi_args ea; ea.a = a; ea.b = b; ea.c_re = c_re; ea.c_im = c_im; ea.prev = phi_prev; ea.accv = psi_t; ea.halo = halo;
  int epi;
  switch (epilogue) {
    case 0: epi = SD_EPI_PLAIN; break;
    case 1: epi = SD_EPI_RESCALE; break;
    case 2:
      if (dtype != SD_C128 || !phi_prev || !psi_t) return sd_set_err(ctx, SD_EARG, "Chebyshev epilogue needs ComplexF64 phi_prev and psi_t");
      epi = SD_EPI_CHEB; break;
    case 3:
      if (!phi_prev) return sd_set_err(ctx, SD_EARG, "recurrence epilogue needs phi_prev");
      epi = SD_EPI_RECUR; break;
    default: return sd_set_err(ctx, SD_EARG, "unknown epilogue");
  }
  return sd_launch_apply(ctx, m, dtype, out, psi, epi, ea, part);
}

int sd_apply_sharded_cheb2_dev(sd_ctx *ctx, const sd_model *m, void *out, const void *psi, const void *halo, int64_t n_local,
                               double a, double b, double c0_re, double c0_im, double c_re, double c_im,
                               const void *phi_prev, void *psi_t, int part) {
  int rc = check_apply_args(ctx, m, SD_C128, out, psi, n_local);
  if (rc) return rc;
  if (n_local != m->n_local) return sd_set_err(ctx, SD_EDIM, "vector length does not match the local basis dimension");
  if (part < 0 || part > 2) return sd_set_err(ctx, SD_EARG, "part must be 0 (all), 1 (interior) or 2 (boundary)");
  if (m->n_halo > 0 && !halo && part != 1) return sd_set_err(ctx, SD_EARG, "this shard needs a halo buffer");
  if (!phi_prev || !psi_t) return sd_set_err(ctx, SD_EARG, "null vector");
  sd_epi_args ea; ea.a = a; ea.b = b; ea.c0_re = c0_re; ea.c0_im = c0_im; ea.c_re = c_re; ea.c_im = c_im;
  ea.prev = phi_prev; ea.accv = psi_t; ea.halo = halo;
  return sd_launch_apply(ctx, m, SD_C128, out, psi, SD_EPI_CHEB2, ea, part);
}

int sd_apply_rescaled_dev(sd_ctx *ctx, const sd_model *m, int dtype, void *out, const void *psi, int64_t n, double a,
                          double b) {
  int rc = check_apply_args(ctx, m, dtype, out, psi, n);
  if (rc) return rc;
  if (n != m->n_local) return sd_set_err(ctx, SD_EDIM, "vector length does not match the (local) basis dimension");
  sd_epi_args ea; ea.a = a; ea.b = b;
  return sd_launch_apply(ctx, m, dtype, out, psi, SD_EPI_RESCALE, ea);
}

int sd_cheb_step_dev(sd_ctx *ctx, const sd_model *m, void *phi_next, const void *phi_curr, const void *phi_prev,
                     void *psi_t, int64_t n, double a, double b, double c_re, double c_im) {
  int rc = check_apply_args(ctx, m, SD_C128, phi_next, phi_curr, n);
  if (rc) return rc;
  if (!phi_prev || !psi_t) return sd_set_err(ctx, SD_EARG, "null vector");
  if (n != m->n_local) return sd_set_err(ctx, SD_EDIM, "vector length does not match the (local) basis dimension");
  sd_epi_args ea; ea.a = a; ea.b = b; ea.c_re = c_re; ea.c_im = c_im; ea.prev = phi_prev; ea.accv = psi_t;
  return sd_launch_apply(ctx, m, SD_C128, phi_next, phi_curr, SD_EPI_CHEB, ea);
}

static int apply_host(sd_ctx *ctx, const sd_model *m, int dtype, void *out, const void *psi, int64_t n, int epi,
                      double a, double b) {
  int rc = check_apply_args(ctx, m, dtype, out, psi, n);
  if (rc) return rc;
  if (m->nranks != 1) return sd_set_err(ctx, SD_EARG, "host-pointer applies need an unsharded model");
  if (n != m->N) return sd_set_err(ctx, SD_EDIM, "vector length does not match the basis dimension");
  const size_t bytes = (size_t)n * (dtype == SD_C128 ? 16 : 8);
  void *din, *dout;
  if ((rc = stage_buf(ctx, 0, bytes, &din)) || (rc = stage_buf(ctx, 1, bytes, &dout))) return rc;
  if ((rc = sd_xfer_h2d(ctx, din, psi, bytes))) return rc;
  sd_epi_args ea; ea.a = a; ea.b = b;
  rc = sd_launch_apply(ctx, m, dtype, dout, din, epi, ea);
  if (rc) return rc;
  return sd_xfer_d2h(ctx, out, dout, bytes);
}

int sd_apply(sd_ctx *ctx, const sd_model *m, int dtype, void *out, const void *psi, int64_t n) {
  return apply_host(ctx, m, dtype, out, psi, n, SD_EPI_PLAIN, 1.0, 0.0);
}

int sd_apply_rescaled(sd_ctx *ctx, const sd_model *m, int dtype, void *out, const void *psi, int64_t n, double a, double b) {
  return apply_host(ctx, m, dtype, out, psi, n, SD_EPI_RESCALE, a, b);
}

int sd_szq_dev(sd_ctx *ctx, const sd_model *m, int dtype_in, const void *psi0, int64_t n, double q, void *phi) {
  if (!ctx) return SD_EARG;
  if (!m || !psi0 || !phi) return sd_set_err(ctx, SD_EARG, "null argument");
  if (n != m->n_local) return sd_set_err(ctx, SD_EDIM, "vector length does not match the (local) basis dimension");
  return sd_launch_szq(ctx, m, dtype_in, psi0, q, phi);
}

int sd_szq(sd_ctx *ctx, const sd_model *m, int dtype_in, const void *psi0, int64_t n, double q, void *phi) {
  if (!ctx) return SD_EARG;
  if (!m || !psi0 || !phi) return sd_set_err(ctx, SD_EARG, "null argument");
  if (dtype_in != SD_F64 && dtype_in != SD_C128) return sd_set_err(ctx, SD_EARG, "bad dtype");
  if (m->nranks != 1) return sd_set_err(ctx, SD_EARG, "host-pointer entry points need an unsharded model");
  if (n != m->N) return sd_set_err(ctx, SD_EDIM, "vector length does not match the basis dimension");
  const size_t bin = (size_t)n * (dtype_in == SD_C128 ? 16 : 8), bout = (size_t)n * 16;
  TmpDev din(ctx), dout(ctx);
  int rc;
  if ((rc = din.alloc(bin)) || (rc = dout.alloc(bout))) return rc;
  if ((rc = sd_xfer_h2d(ctx, din.p, psi0, bin))) return rc;
  rc = sd_launch_szq(ctx, m, dtype_in, din.p, q, dout.p);
  if (rc) return rc;
  return sd_xfer_d2h(ctx, phi, dout.p, bout);
}

int sd_bench_apply_dev(sd_ctx *ctx, const sd_model *m, int dtype, void *buf_a, void *buf_b, int64_t n, int reps,
                       float *ms_per_apply) {
  int rc = check_apply_args(ctx, m, dtype, buf_a, buf_b, n);
  if (rc) return rc;
  if (n != m->n_local) return sd_set_err(ctx, SD_EDIM, "vector length does not match the (local) basis dimension");
  if (reps < 1 || !ms_per_apply) return sd_set_err(ctx, SD_EARG, "bad reps");
  sd_epi_args ea;
  void *src = buf_a, *dst = buf_b;
  SD_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  for (int r = 0; r < reps; ++r) {
    rc = sd_launch_apply(ctx, m, dtype, dst, src, SD_EPI_PLAIN, ea);
    if (rc) return rc;
    void *t = src; src = dst; dst = t;
  }
  SD_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  SD_HIP(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  SD_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *ms_per_apply = ms / reps;
  return SD_OK;
}

// Diagnostic only (not in include/spindyn.h): runs one apply with per-tile s_memtime stamps and returns the mean
// shader-cycle duration of each phase (prologue, list, diag+LDS, far bonds, barrier, suffix, epilogue) and the
// mean tile lifetime.  Never used by the product path.
int sd_debug_phase_profile(sd_ctx *ctx, sd_model *m, int dtype, void *out, const void *psi, double *phases /*8*/) {
  if (!ctx || !m || !m->dev_ready || m->p < 0) return SD_EARG;
  const size_t nt = m->single_prefix.size();
  unsigned long long *d = nullptr;
  SD_HIP(ctx, hipMalloc((void **)&d, nt * 8 * sizeof(unsigned long long)));
  SD_HIP(ctx, hipMemset(d, 0, nt * 8 * sizeof(unsigned long long)));
  m->dm.stamps = d;
  sd_epi_args ea;
  int rc = sd_launch_apply(ctx, m, dtype, out, psi, SD_EPI_PLAIN, ea);
  m->dm.stamps = nullptr;
  if (!rc) {
    std::vector<unsigned long long> h(nt * 8);
    hipError_t e = hipMemcpy(h.data(), d, nt * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = SD_EHIP;
    for (int k = 0; k < 8; ++k) phases[k] = 0.0;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (size_t t = 0; t < nt; ++t) {
      for (int k = 0; k < 6; ++k) phases[k] += (double)(h[8 * t + k + 1] - h[8 * t + k]);
      phases[6] += (double)(h[8 * t + 6] - h[8 * t]);
      tmin = std::min(tmin, h[8 * t + 7]); tmax = std::max(tmax, h[8 * t + 7]);
    }
    for (int k = 0; k < 7; ++k) phases[k] /= (double)nt;
    phases[7] = (double)(tmax - tmin) * 10.0;  // ns between first and last tile end (100 MHz realtime counter)
  }
  (void)hipFree(d);
  return rc;
}

// ---- observables / initial states ("next" rows f2, f3) ----

static int obs_dev(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi, int64_t n, int what, double *out, double *q_out) {
  if (!ctx) return SD_EARG;
  if (!m || !psi || !out) return sd_set_err(ctx, SD_EARG, "null argument");
  if (m->nranks != 1) return sd_set_err(ctx, SD_EARG, "observables need an unsharded model");
  if (n != m->N) return sd_set_err(ctx, SD_EDIM, "vector length does not match the basis dimension");
  const int L = m->L;
  if (what == 0) return sd_launch_observable(ctx, m, dtype, psi, 0, out);       // magnetization_per_site
  std::vector<double> S(L), R(L), Cr(L);
  int rc = sd_launch_observable(ctx, m, dtype, psi, 0, S.data());
  if (!rc) rc = sd_launch_observable(ctx, m, dtype, psi, 1, R.data());
  if (rc) return rc;
  for (int r = 0; r < L; ++r) {                                                   // src/Observables.jl:83-91
    double prod = 0.0;
    for (int i = 1; i <= L; ++i) { const int j = ((i + r - 1) % L) + 1; prod += S[i - 1] * S[j - 1]; }
    Cr[r] = (R[r] - prod) / L;
  }
  if (what == 1) { for (int r = 0; r < L; ++r) out[r] = Cr[r]; return SD_OK; }
  const double PI = 3.14159265358979323846;                                      // :100-109, plain DFT of C_r
  for (int k = 0; k < L; ++k) {
    double s = 0.0;
    for (int r = 0; r < L; ++r) s += Cr[r] * std::cos(2 * PI * k * r / L);
    out[k] = s;
    if (q_out) q_out[k] = 2 * PI * k / L;
  }
  return SD_OK;
}

static int obs_host(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi, int64_t n, int what, double *out, double *q_out) {
  if (!ctx) return SD_EARG;
  if (!m || !psi || !out) return sd_set_err(ctx, SD_EARG, "null argument");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "bad dtype");
  if (n != m->N) return sd_set_err(ctx, SD_EDIM, "vector length does not match the basis dimension");
  const size_t bytes = (size_t)n * (dtype == SD_C128 ? 16 : 8);
  TmpDev d(ctx);
  int rc = d.alloc(bytes);
  if (rc) return rc;
  if ((rc = sd_xfer_h2d(ctx, d.p, psi, bytes))) return rc;
  return obs_dev(ctx, m, dtype, d.p, n, what, out, q_out);
}

int sd_magnetization(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_host, int64_t n, double *mags_out) {
  return obs_host(ctx, m, dtype, psi_host, n, 0, mags_out, nullptr);
}
int sd_magnetization_dev(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_dev, int64_t n, double *mags_out) {
  return obs_dev(ctx, m, dtype, psi_dev, n, 0, mags_out, nullptr);
}
int sd_connected_correlations(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_host, int64_t n, double *C_out) {
  return obs_host(ctx, m, dtype, psi_host, n, 1, C_out, nullptr);
}
int sd_connected_correlations_dev(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_dev, int64_t n, double *C_out) {
  return obs_dev(ctx, m, dtype, psi_dev, n, 1, C_out, nullptr);
}
int sd_structure_factor(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_host, int64_t n, double *q_out, double *S_out) {
  return obs_host(ctx, m, dtype, psi_host, n, 2, S_out, q_out);
}
int sd_structure_factor_dev(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_dev, int64_t n, double *q_out, double *S_out) {
  return obs_dev(ctx, m, dtype, psi_dev, n, 2, S_out, q_out);
}

int sd_spin_operator(sd_ctx *ctx, const sd_model *m, int dtype, int site, int op, const void *psi_host, int64_t n,
                     void *out_host) {
  // create_spin_operator(site, op)(psi, model)  -- src/Hamiltonian.jl:49-136
  if (!ctx) return SD_EARG;
  if (!m || !psi_host || !out_host) return sd_set_err(ctx, SD_EARG, "null argument");
  if (site < 1) return sd_set_err(ctx, SD_EARG, "site must be at least 1");                                    // :50
  if (op < SD_SPIN_Z || op > SD_SPIN_Y) return sd_set_err(ctx, SD_EARG, "unsupported spin operator");        // :52-55
  if (site > m->L) return sd_set_err(ctx, SD_EARG, "site is outside the model");                             // :60-61
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "bad dtype");
  if (n != m->N) return sd_set_err(ctx, SD_EDIM, "state vector length does not match the basis");             // :63-66
  if (m->nup >= 0 && op != SD_SPIN_Z)                                                                          // :68-73
    return sd_set_err(ctx, SD_EARG, "operator changes total magnetization and cannot be applied within a fixed-nup sector");
  if (op == SD_SPIN_Y && dtype != SD_C128)   // result[...] += ±0.5im*psi into zeros(Float64) is an InexactError upstream (:118-127)
    return sd_set_err(ctx, SD_EARG, "S^y of a Float64 vector is complex: pass ComplexF64");
  if (m->nranks != 1) return sd_set_err(ctx, SD_EARG, "needs an unsharded model");
  const size_t bin = (size_t)n * (dtype == SD_C128 ? 16 : 8), bout = bin;
  TmpDev din(ctx), dout(ctx);
  int rc;
  if ((rc = din.alloc(bin)) || (rc = dout.alloc(bout))) return rc;
  if ((rc = sd_xfer_h2d(ctx, din.p, psi_host, bin))) return rc;
  rc = sd_launch_spin_op(ctx, m, dtype, site, op, din.p, dout.p);
  if (rc) return rc;
  return sd_xfer_d2h(ctx, out_host, dout.p, bout);
}

int sd_initial_state_index(const sd_model *m, int kind, const int *flips, int nflips, int64_t *idx0_out) {
  if (!m || !idx0_out || nflips < 0 || (nflips && !flips)) return SD_EARG;
  const int L = m->L;
  uint64_t s = 0;
  if (kind == SD_STATE_DOMAIN_WALL) {                          // src/InitialStates.jl:9-34
    const int nup = m->nup >= 0 ? m->nup : (L + 1) / 2;
    for (int i = 0; i < nup; ++i) s |= (uint64_t)1 << i;
  } else if (kind == SD_STATE_NEEL) {                          // :40-63
    for (int i = 0; i < L; i += 2) s |= (uint64_t)1 << i;
  } else if (kind == SD_STATE_POLARIZED_UP) {                  // :70-89
    s = ((uint64_t)1 << L) - 1;
  } else if (kind == SD_STATE_POLARIZED_DOWN) {
    s = 0;
  } else if (kind == SD_STATE_POLARIZED_FLIPS) {               // :97-130
    for (int k = 0; k < nflips; ++k) if (flips[k] < 1 || flips[k] > L) return fail(m->ctx, SD_EARG, "flip site is outside the model");
    s = ((uint64_t)1 << L) - 1;
    for (int k = 0; k < nflips; ++k) s ^= (uint64_t)1 << (flips[k] - 1);
  } else return fail(m->ctx, SD_EARG, "unknown initial-state kind");
  const int64_t idx = sd_rank_host(m, s);
  if (idx < 0) return fail(m->ctx, SD_EARG, "requested state is not contained in the model basis");
  *idx0_out = idx;
  return SD_OK;
}

int sd_fill_randn_dev(sd_ctx *ctx, void *x, int64_t n, uint64_t seed, uint64_t first) {
  if (!ctx) return SD_EARG;
  if (!x || n < 0) return sd_set_err(ctx, SD_EARG, "bad argument");
  return sd_k_fill_randn(ctx, (double *)x, n, seed, first);
}

int sd_dot_dev(sd_ctx *ctx, int dtype, const void *x, const void *y, int64_t n, double *out) {
  if (!ctx) return SD_EARG;
  if (!out || n < 0 || (n > 0 && (!x || !y))) return sd_set_err(ctx, SD_EARG, "bad argument");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "bad dtype");
  out[0] = out[1] = 0.0;
  if (n == 0) return SD_OK;
  int rc = sd_k_dot(ctx, dtype == SD_C128 ? 2 : 1, (const double *)x, (const double *)y, n, 4);
  if (rc) return rc;
  return sd_read_scalars(ctx, 4, dtype == SD_C128 ? 2 : 1, out);
}

int sd_nrm2sq_dev(sd_ctx *ctx, int dtype, const void *x, int64_t n, double *out) {
  if (!ctx) return SD_EARG;
  if (!out || n < 0 || (n > 0 && !x)) return sd_set_err(ctx, SD_EARG, "bad argument");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "bad dtype");
  *out = 0.0;
  if (n == 0) return SD_OK;
  int rc = sd_k_nrm2sq(ctx, (const double *)x, n * (dtype == SD_C128 ? 2 : 1), 4);
  if (rc) return rc;
  return sd_read_scalars(ctx, 4, 1, out);
}

int sd_fill_randn_host(double *x, int64_t n, uint64_t seed, uint64_t first) {
  if (!x || n < 0) return SD_EARG;
  for (int64_t i = 0; i < n; ++i) x[i] = sd_randn_host(seed, first + (uint64_t)i);
  return SD_OK;
}

}  // extern "C"
