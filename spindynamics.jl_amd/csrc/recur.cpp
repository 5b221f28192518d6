// On-device recursions: the host logic of the reference's solvers with every
// N-vector living in HBM.  Each iteration is one apply kernel (with a fused
// epilogue where the recursion allows it) plus a few BLAS-1 kernels; only the
// scalars the host-side tridiagonal / Chebyshev algebra needs cross PCIe.
//
// Reference functions followed (file:line under the reference repository):
//   lanczos_extremal        src/Lanczos.jl:27-84
//   estimate_energy_bounds  src/Lanczos.jl:255-271
//   lanczos_groundstate     src/Lanczos.jl:87-181
//   lanczos_tridiag         src/Lanczos.jl:196-246
//   krylov_time_evolve      src/TimeEvolution/Krylov.jl:136-192
//   chebyshev_time_evolve   src/TimeEvolution/Chebyshev.jl:61-124
//   compute_chebyshev_moments / kpm_sw / kpm_sqw   src/KPM_Sqw.jl:34-128,191-256
//   spectral_from_tridiagonal / lanczos_sqw        src/LanczosSqw.jl:18-80
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "sd_internal.hpp"

namespace {

#define RC(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)
// No C++ exception may cross the C ABI: a failed host allocation (std::vector of the m x m work, std::string of a message) inside a
// recursion comes back as a status.
#define SD_ABI_GUARD(ctx, call)                                                                              \
  try {                                                                                                       \
    return (call);                                                                                            \
  } catch (const std::bad_alloc &) {                                                                          \
    return (ctx) ? sd_set_err((ctx), SD_ENOMEM, "out of host memory inside the call") : SD_ENOMEM;           \
  } catch (const std::exception &e__) {                                                                       \
    return (ctx) ? sd_set_err((ctx), SD_EINTERNAL, std::string("unexpected exception: ") + e__.what()) : SD_EINTERNAL; \
  } catch (...) {                                                                                             \
    return (ctx) ? sd_set_err((ctx), SD_EINTERNAL, "unexpected exception") : SD_EINTERNAL;                   \
  }

struct DBuf {   // device work vector, taken from / returned to the context's pool
  double *p = nullptr;
  sd_ctx *owner = nullptr;
  size_t bytes = 0;
  DBuf() = default;
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
  ~DBuf() { release(); }
  void release() {
    if (p) sd_pool_give(owner, p, bytes);
    p = nullptr; bytes = 0;
  }
  int alloc(sd_ctx *ctx, int64_t doubles) {
    release();
    owner = ctx;
    void *q = nullptr;
    int rc = sd_pool_take(ctx, sizeof(double) * (size_t)std::max<int64_t>(doubles, 1), &q, &bytes);
    p = (double *)q;
    return rc;
  }
};

// the caller's host vectors (src/PublicAPI.jl:50-88): staged through the context's pinned ring when large (xfer.cpp)
int h2d(sd_ctx *ctx, double *d, const void *h, int64_t doubles) { return sd_xfer_h2d(ctx, d, h, sizeof(double) * (size_t)doubles); }
int d2h(sd_ctx *ctx, void *h, const double *d, int64_t doubles) { return sd_xfer_d2h(ctx, h, d, sizeof(double) * (size_t)doubles); }
int d2d(sd_ctx *ctx, double *dst, const double *src, int64_t doubles) {
  SD_HIP(ctx, hipMemcpyAsync(dst, src, sizeof(double) * (size_t)doubles, hipMemcpyDeviceToDevice, ctx->stream));
  return SD_OK;
}

// The caller's operator (sd_ctx_set_apply_callback; the reference's `applyH!` argument): out <- H psi by the callback on the
// context's stream, then the recursion's fused step as one elementwise pass over out.
int user_op(sd_ctx *ctx, const sd_model *m, int dtype, double *out, const double *psi, int64_t n, int epi, const sd_epi_args &ea) {
  if (ctx->user_apply(ctx->user_apply_data, dtype, out, psi, n, (void *)ctx->stream))
    return sd_set_err(ctx, SD_ECOMM, "the apply callback returned an error");
  if (epi == SD_EPI_PLAIN && !ea.negate) return SD_OK;
  return sd_launch_epilogue_only(ctx, dtype, n, out, out, psi, epi, ea);
}
// out <- H psi for the entry points that run their own loop on an unsharded model
int plain_op(sd_ctx *ctx, const sd_model *m, int dtype, double *out, const double *psi) {
  sd_epi_args ea;
  ++ctx->n_applies;
  if (ctx->user_apply) return user_op(ctx, m, dtype, out, psi, m->n_local, SD_EPI_PLAIN, ea);
  return sd_launch_apply(ctx, m, dtype, out, psi, SD_EPI_PLAIN, ea);
}

// What a recursion needs from "the operator": the apply on this rank's rows (halo exchange included) and the sum of device
// scalars over the ranks.  Unsharded (comm == nullptr, nranks == 1) both reduce to the plain launch / nothing, so the same
// loops serve the single-GPU and the sharded entry points.
struct Op {
  sd_ctx *ctx = nullptr;
  const sd_model *m = nullptr;
  sd_comm *comm = nullptr;
  int64_t n = 0;             // rows of this rank (== N unsharded)
  DBuf halo, send;           // imported partner tiles / tiles packed for the peers (sharded plans; ComplexF64-sized)
  bool overlap = true;       // interior tiles run while the exchange is in flight
  bool use_callback = true;  // false: always the built-in operator (the operator-level entry sd_apply_sharded, which a caller's
                             // operator may itself call on a sharded model: include/spindyn.h, sd_ctx_set_apply_callback)

  int init(sd_ctx *c, const sd_model *mm, sd_comm *cm) {
    if (!c) return SD_EARG;
    ctx = c; m = mm; comm = cm;
    if (!m || !m->dev_ready) return sd_set_err(ctx, SD_EARG, "model has no device tables");
    if (m->nranks != 1 && !comm) return sd_set_err(ctx, SD_EARG, "a sharded model needs a communicator (the unsharded entry points take an unsharded model)");
    if (comm && sd_comm_nranks(comm) != m->nranks) return sd_set_err(ctx, SD_EARG, "communicator size does not match the model's shard count");
    SD_HIP(ctx, hipSetDevice(ctx->device));
    n = m->n_local;
    if (m->nranks > 1) {
      RC(halo.alloc(ctx, 2 * std::max<int64_t>(m->n_halo, 1)));
      if (m->packed) RC(send.alloc(ctx, 2 * std::max<int64_t>(m->n_send, 1)));
    }
    return SD_OK;
  }
  // out = epilogue(H psi) on the owned rows.  Sharded: pack (cell mode), post the exchange, interior tiles, wait, boundary tiles.
  int apply(int dtype, double *out, const double *psi, int epi, sd_epi_args ea) {
    ++ctx->n_applies;
    if (use_callback && ctx->user_apply) return user_op(ctx, m, dtype, out, psi, n, epi, ea);
    if (m->nranks == 1) return sd_launch_apply(ctx, m, dtype, out, psi, epi, ea, 0);
    ea.halo = halo.p;
    const void *src = psi;
    if (m->packed) { RC(sd_launch_pack(ctx, m, dtype, psi, send.p)); src = send.p; }
    RC(sd_comm_exchange_start(ctx, comm, m, dtype, src, halo.p));
    if (overlap && m->n_interior > 0 && n > 0) {
      RC(sd_launch_apply(ctx, m, dtype, out, psi, epi, ea, 1));
      RC(sd_comm_exchange_wait(ctx, comm, m));
      return sd_launch_apply(ctx, m, dtype, out, psi, epi, ea, 2);
    }
    RC(sd_comm_exchange_wait(ctx, comm, m));
    return sd_launch_apply(ctx, m, dtype, out, psi, epi, ea, 0);
  }
  // device scalars <- their sum over the ranks (in place, ordered on the context's stream)
  int reduce(double *dev, int count) { return sd_comm_allreduce_dev(ctx, comm, dev, count); }
};

double norm_dev(Op &op, const double *x, int64_t n, int *rc) {
  double v = 0.0;
  sd_ctx *ctx = op.ctx;
  *rc = sd_k_nrm2sq(ctx, x, n, 2);
  if (!*rc) *rc = op.reduce(ctx->d_scalars + 2, 1);
  if (!*rc) *rc = sd_read_scalars(ctx, 2, 1, &v);
  return std::sqrt(v);
}

// Every SD_BREAK_PEEK steps the queued recursions read back the betas filed so far (one small copy + one synchronisation)
// and stop queueing when one is below tol or not a number: the reference's break (src/Lanczos.jl:66-70, 228-231) is then
// at most SD_BREAK_PEEK - 1 discarded steps late instead of lanc_m - j.
constexpr int SD_BREAK_PEEK = 32;
int peek_breakdown(sd_ctx *ctx, const double *d_be, int count, double tol, std::vector<double> &buf, bool *broke) {
  buf.resize((size_t)count);
  SD_HIP(ctx, hipMemcpyAsync(buf.data(), d_be, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *broke = false;
  for (int k = 0; k < count; ++k)
    if (!(std::fabs(buf[k]) >= tol)) { *broke = true; break; }
  return SD_OK;
}

// ---- Lanczos at launch-bound sizes: two launches per step, any number of start vectors per launch ----
// The un-reorthogonalised recursion of lanczos_extremal (form 0, src/Lanczos.jl:27-84) and lanczos_tridiag (form 1, :196-246)
// for Qb NORMALISED start vectors stored back to back in u0 (consumed): step j is one batched apply with the <u|Hu> epilogue,
// its per-tile pairs left unreduced, and one batched update pass that sums those pairs and the previous passes' |w|^2 pairs
// itself (k_lanczos_fold_p).  Nothing touches the host until the end.  alpha, beta: Qb rows of mm (beta[., mm-1] unused).
// Same arithmetic per element as the four-launch form; the reductions are summed in another (fixed) order.
bool lanczos_fused_ok(const Op &op, int Qb) {
  const sd_model *m = op.m;
  static const int on = getenv("SD_LANCZOS_FUSED") ? atoi(getenv("SD_LANCZOS_FUSED")) : 1;
  return on && m->nranks == 1 && !op.ctx->user_apply && m->p >= 0 && m->dm.n_singles <= 4096 && op.n <= ((int64_t)1 << 22) &&
         (int64_t)Qb * op.n * 16 * 3 <= ((int64_t)4 << 30);
}
int lanczos_fused(Op &op, int Qb, double *u0, int mm, int form, int negate, double tol, std::vector<double> &alpha,
                  std::vector<double> &beta) {
  sd_ctx *ctx = op.ctx;
  const int64_t N = op.n;
  const int nt = op.m->dm.n_singles, nbf = sd_k_lanczos_fold_blocks(N);
  DBuf wb, vp, ab, n2;
  RC(wb.alloc(ctx, 2 * N * Qb)); RC(vp.alloc(ctx, 2 * N * Qb));
  const int64_t srow = 2 * (int64_t)mm;                                  // per vector: alpha[mm] | beta[mm]
  RC(ab.alloc(ctx, srow * Qb)); RC(n2.alloc(ctx, 3 * 2 * (int64_t)nbf * Qb));
  SD_HIP(ctx, hipMemsetAsync(ab.p, 0, sizeof(double) * (size_t)(srow * Qb), ctx->stream));
  double *d_al = ab.p, *d_be = ab.p + mm;
  auto n2buf = [&](int j) { return n2.p + (size_t)(j % 3) * 2 * (size_t)nbf * Qb; };
  sd_epi_args ea; ea.negate = negate; ea.batch = Qb; ea.bstride = N; ea.no_reduce = 1;
  double *ucur = u0, *uprev = vp.p, *t = wb.p;
  std::vector<double> peek;
  for (int j = 1; j <= mm; ++j) {
    ctx->n_applies += Qb - 1;
    RC(op.apply(SD_C128, t, ucur, SD_EPI_DOT, ea));                      // per-tile pairs of <u|Hu> -> ctx->d_partials
    const double *n2c = j > 1 ? n2buf(j - 1) : nullptr, *n2p = j > 2 ? n2buf(j - 2) : nullptr;
    if (j == mm) {                                                       // alpha_m; no vector behind it
      RC(sd_k_lanczos_fold_scalars_p(ctx, Qb, form, ctx->d_partials, nt, n2c, nbf, d_al + (j - 1), j > 1 ? d_be + (j - 2) : nullptr, srow));
      break;
    }
    RC(sd_k_lanczos_fold_p(ctx, t, ucur, j > 1 ? uprev : nullptr, N, Qb, N, form, ctx->d_partials, nt, n2c, n2p, nbf, d_al + (j - 1),
                           j > 1 ? d_be + (j - 2) : nullptr, srow, n2buf(j), nbf));
    { double *old = uprev; uprev = ucur; ucur = t; t = old; }
    if (j % SD_BREAK_PEEK == 0 && j < mm - 1) {                          // stop queueing once EVERY vector has broken down
      peek.resize((size_t)(srow * Qb));
      SD_HIP(ctx, hipMemcpyAsync(peek.data(), ab.p, sizeof(double) * peek.size(), hipMemcpyDeviceToHost, ctx->stream));
      SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
      bool all_broke = true;
      for (int q = 0; q < Qb && all_broke; ++q) {
        bool broke = false;
        for (int k = 0; k < j - 1; ++k) if (!(std::fabs(peek[(size_t)(srow * q) + mm + k]) >= tol)) { broke = true; break; }
        all_broke = broke;
      }
      if (all_broke) break;
    }
  }
  std::vector<double> host((size_t)(srow * Qb));
  SD_HIP(ctx, hipMemcpyAsync(host.data(), ab.p, sizeof(double) * host.size(), hipMemcpyDeviceToHost, ctx->stream));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  alpha.assign((size_t)Qb * mm, 0.0); beta.assign((size_t)Qb * mm, 0.0);
  for (int q = 0; q < Qb; ++q)
    for (int k = 0; k < mm; ++k) { alpha[(size_t)q * mm + k] = host[(size_t)(srow * q) + k]; beta[(size_t)q * mm + k] = host[(size_t)(srow * q) + mm + k]; }
  return SD_OK;
}

// lanczos_extremal on device vectors; v_prev (2n doubles, un-normalised start) is consumed
int extremal_dev(Op &op, int lanc_m, double tol, double *v_prev, int negate, double *emin, double *emax) {
  sd_ctx *ctx = op.ctx;
  const int64_t N = op.n;
  const int mm = (int)std::min<int64_t>(lanc_m, op.m->N);
  if (mm < 1) return sd_set_err(ctx, SD_EARG, "lanc_m must be >= 1");
  DBuf w, vc;
  RC(w.alloc(ctx, 2 * N)); RC(vc.alloc(ctx, 2 * N));
  double *v_curr = vc.p;
  int rc = 0;
  double nrm = norm_dev(op, v_prev, 2 * N, &rc); RC(rc);
  RC(sd_k_scale_div(ctx, v_prev, v_prev, 2 * N, nrm));                     // :40
  if (lanczos_fused_ok(op, 1)) {                                           // launch-bound sizes: two launches per step
    std::vector<double> al, be;
    RC(lanczos_fused(op, 1, v_prev, mm, 0, negate, tol, al, be));
    int actual = mm;
    for (int j = 1; j < mm; ++j)
      if (!(be[j - 1] >= tol)) { actual = j; break; }                      // :66-70
    std::vector<double> ev(actual);
    if (sd_symtridiag_eig(actual, al.data(), be.data(), ev.data(), nullptr))   // :80-83
      return sd_set_err(ctx, SD_EINTERNAL, "tridiagonal eigen-solver did not converge");
    *emin = ev[0]; *emax = ev[actual - 1];
    return SD_OK;
  }
  // the loop is queued without host round trips (alpha_j, beta_j stay on the device, see tridiag_dev); the break on
  // beta_j < tol (:66-70) is applied to the values read back at the end.  Vectors stay un-normalised (k_lanczos_fold).
  DBuf ab; RC(ab.alloc(ctx, 4 * (int64_t)mm + 2));
  double *d_al = ab.p, *d_be = ab.p + mm, *d_n2 = ab.p + 2 * (int64_t)mm;      // d_n2[2j]: |w_j|^2
  SD_HIP(ctx, hipMemsetAsync(ab.p, 0, sizeof(double) * (4 * (size_t)mm + 2), ctx->stream));
  sd_epi_args ea; ea.negate = negate;
  std::vector<double> peek;
  double *ucur = v_prev, *uprev = v_curr, *t = w.p;
  const double *n2c = nullptr, *n2p = nullptr;       // |ucur|^2, |uprev|^2 on the device; null: normalised
  for (int j = 1; j <= mm; ++j) {
    RC(op.apply(SD_C128, t, ucur, SD_EPI_DOT, ea));                        // :51 + :55 fused -> d_scalars[0]
    RC(op.reduce(ctx->d_scalars + 0, 2));
    if (j == mm) {                                                         // alpha_m; no vector behind it
      RC(sd_k_lanczos_fold_scalars(ctx, 0, ctx->d_scalars + 0, n2c, d_al + (j - 1), j > 1 ? d_be + (j - 2) : nullptr));
      break;
    }
    double *n2o = d_n2 + 2 * (int64_t)j;
    RC(sd_k_lanczos_fold(ctx, t, ucur, j == 1 ? nullptr : uprev, N, 0, ctx->d_scalars + 0, n2c, n2p, d_al + (j - 1),
                         j > 1 ? d_be + (j - 2) : nullptr, n2o));          // :58-65 (beta_{j-1} is filed by this pass)
    RC(op.reduce(n2o, 1));
    { double *old = uprev; uprev = ucur; ucur = t; t = old; }
    n2p = n2c; n2c = n2o;
    if (j % SD_BREAK_PEEK == 0 && j < mm) {
      bool broke = false;
      RC(peek_breakdown(ctx, d_be, j - 1, tol, peek, &broke));
      if (broke) break;
    }
  }
  std::vector<double> host(2 * (size_t)mm);
  SD_HIP(ctx, hipMemcpyAsync(host.data(), ab.p, sizeof(double) * 2 * (size_t)mm, hipMemcpyDeviceToHost, ctx->stream));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int actual = mm;
  for (int j = 1; j < mm; ++j)
    if (!(host[mm + j - 1] >= tol)) { actual = j; break; }                 // :66-70
  std::vector<double> alpha(host.begin(), host.begin() + mm), beta(host.begin() + mm, host.end());
  std::vector<double> ev(actual);
  int rce = sd_symtridiag_eig(actual, alpha.data(), beta.data(), ev.data(), nullptr);   // :80-83
  if (rce) return sd_set_err(ctx, SD_EINTERNAL, "tridiagonal eigen-solver did not converge");
  *emin = ev[0]; *emax = ev[actual - 1];
  // note: v_prev / v_curr may have been swapped; the caller's buffer is scratch from here on
  (void)hipStreamSynchronize(ctx->stream);
  return SD_OK;
}

// start vector of this rank's rows: the caller's (host pointer, unsharded calls; device pointer, sharded calls) or the
// counter-based N(0,1) vector keyed by the GLOBAL element index -- the same state for every sharding
int start_vector_op(Op &op, double *d, const void *given, bool given_on_dev, int64_t doubles, uint64_t seed) {
  if (given) return given_on_dev ? d2d(op.ctx, d, (const double *)given, doubles) : h2d(op.ctx, d, given, doubles);
  if (op.m->nranks > 1) return sd_launch_fill_randn_local(op.ctx, op.m, SD_C128, d, seed);
  return sd_k_fill_randn(op.ctx, d, doubles, seed, 0);
}

int moments_dev(Op &op, const double *phi, int M, double a, double b, double *mu) {
  // compute_chebyshev_moments  src/KPM_Sqw.jl:95-128  (phi: device, c128, this rank's rows, normalised by the caller)
  sd_ctx *ctx = op.ctx;
  const int64_t N = op.n;
  if (M < 2) return sd_set_err(ctx, SD_EARG, "kpm_m must be >= 2");
  DBuf b0, b1, b2;
  RC(b0.alloc(ctx, 2 * N)); RC(b1.alloc(ctx, 2 * N)); RC(b2.alloc(ctx, 2 * N));
  double s[2];
  for (int doubling = ctx->kpm_doubling ? 1 : 0; doubling >= 0; --doubling) {
    double *v_prev = b0.p, *v_curr = b1.p, *v_next = b2.p;
    RC(d2d(ctx, v_prev, phi, 2 * N));
    RC(sd_k_dot(ctx, 2, phi, v_prev, N, 4)); RC(op.reduce(ctx->d_scalars + 4, 2));
    RC(sd_read_scalars(ctx, 4, 2, s)); mu[0] = s[0];                                                // :103
    sd_epi_args ea; ea.a = a; ea.b = b;
    if (!doubling) {
      // the reference's recursion: one moment <phi|T_k phi> per apply
      ea.phi = phi;
      RC(op.apply(SD_C128, v_curr, v_prev, SD_EPI_RESCALE_DOT, ea));                                // :106-107
      RC(op.reduce(ctx->d_scalars + 0, 2));
      RC(sd_read_scalars(ctx, 0, 2, s)); mu[1] = s[0];
      for (int k = 2; k <= M - 1; ++k) {
        ea.prev = v_prev;
        RC(op.apply(SD_C128, v_next, v_curr, SD_EPI_KPM, ea));                                      // :111-117 fused
        RC(op.reduce(ctx->d_scalars + 0, 2));
        RC(sd_read_scalars(ctx, 0, 2, s));
        mu[k] = s[0];
        const double nv = std::sqrt(s[1]);
        if (nv > 1e3) RC(sd_k_scale_div(ctx, v_next, v_next, 2 * N, nv));                           // :118-121
        double *t = v_prev; v_prev = v_curr; v_curr = v_next; v_next = t;                           // :124
      }
      return SD_OK;
    }
    // Two moments per apply from the same vectors v_n = T_n(H~) phi (T_m T_n = (T_{m+n} + T_{|m-n|})/2, H~ Hermitian):
    //   mu_{2n}   = 2 <v_n|v_n>       - mu_0,      mu_{2n+1} = 2 Re<v_n|v_{n+1}> - mu_1.
    // Same moments as the reference's loop up to rounding (<= 1e-13 here), in half the applies and without re-reading phi.
    // If the reference's overflow guard (:118-121, |v| > 1e3: the bounds do not contain the spectrum) would fire, the
    // identity no longer mirrors what the reference computes, so the reference recursion is run instead.
    ea.phi = nullptr;                                           // epilogue: s0 = Re<v_curr|v_next>, s1 = |v_next|^2
    // every step is queued without a host round trip: step n files its two sums at d_sum[2n..2n+1]; one read-back at the end
    const int nsteps = M / 2;                                   // applies: v_1 .. v_nsteps
    DBuf sm; RC(sm.alloc(ctx, 2 * (int64_t)nsteps + 2));
    ea.sums_dst = sm.p;
    RC(op.apply(SD_C128, v_curr, v_prev, SD_EPI_RESCALE_DOT, ea));
    RC(op.reduce(sm.p, 2));
    for (int n = 1; n < nsteps; ++n) {
      ea.prev = v_prev; ea.sums_dst = sm.p + 2 * n;
      RC(op.apply(SD_C128, v_next, v_curr, SD_EPI_KPM, ea));                                        // v_{n+1}
      RC(op.reduce(sm.p + 2 * n, 2));
      double *t = v_prev; v_prev = v_curr; v_curr = v_next; v_next = t;
    }
    std::vector<double> hs(2 * (size_t)nsteps);
    SD_HIP(ctx, hipMemcpyAsync(hs.data(), sm.p, sizeof(double) * 2 * (size_t)nsteps, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    bool guard = false;
    for (int k = 1; k <= nsteps; ++k) {                         // sums of step k: [Re<v_{k-1}|v_k>, |v_k|^2]
      const double s0 = hs[2 * (size_t)(k - 1)], s1 = hs[2 * (size_t)(k - 1) + 1];
      if (k == 1) mu[1] = s0;
      else if (2 * k - 1 <= M - 1) mu[2 * k - 1] = 2.0 * s0 - mu[1];
      if (2 * k <= M - 1) mu[2 * k] = 2.0 * s1 - mu[0];
      if (!(std::sqrt(s1) <= 1e3)) guard = true;
    }
    if (!guard) return SD_OK;
  }
  return SD_OK;
}

// compute_chebyshev_moments (src/KPM_Sqw.jl:95-128) for Qb normalised vectors AT ONCE: phi = Qb vectors of n elements stored
// back to back.  The reference threads over the momenta (src/KPM_Sqw.jl:218); where one vector cannot fill the chip (its own
// documented sizes: L = 16..20) a recursion step per momentum is a launch-bound 5-10 us whatever it computes, so the momenta's
// vectors share the launches instead: one batched apply + one batched reduction per step for all of them (sd_epi_args::batch).
// Every vector sees exactly the arithmetic of moments_dev -- same kernels per tile, same summation order -- so mu is
// bit-identical to the one-momentum-at-a-time loop.  ok[k] = 0: the reference's overflow guard (:118-121) would have fired for
// vector k; the caller reruns it through moments_dev.  Unsharded tiled plans only (the caller checks).
int moments_dev_batched(Op &op, const double *phi, int Qb, int M, double a, double b, double *mu /* Qb x M */, std::vector<char> &ok) {
  sd_ctx *ctx = op.ctx;
  const int64_t N = op.n;
  if (M < 2) return sd_set_err(ctx, SD_EARG, "kpm_m must be >= 2");
  ok.assign((size_t)Qb, 1);
  const bool doubling = ctx->kpm_doubling != 0;
  const int nsteps = doubling ? M / 2 : M - 1;                  // applies: v_1 .. v_nsteps
  DBuf b0, b1, b2, sm;
  RC(b0.alloc(ctx, 2 * N * Qb)); RC(b1.alloc(ctx, 2 * N * Qb)); RC(b2.alloc(ctx, 2 * N * Qb));
  RC(sm.alloc(ctx, (int64_t)Qb * (2 * (int64_t)nsteps + 2)));
  const int64_t srow = 2 * (int64_t)nsteps + 2;                  // per vector: [mu0, 0, (s0, s1) of step 1, 2, ...]
  double *v_prev = b0.p, *v_curr = b1.p, *v_next = b2.p;
  RC(d2d(ctx, v_prev, phi, 2 * N * Qb));
  for (int k = 0; k < Qb; ++k) RC(sd_k_dot_to(ctx, 2, phi + 2 * N * k, v_prev + 2 * N * k, N, sm.p + srow * k));     // :103
  sd_epi_args ea; ea.a = a; ea.b = b;
  ea.batch = Qb; ea.bstride = N; ea.sums_bstride = srow;
  ea.phi = doubling ? nullptr : phi;                            // doubling: s0 = Re<v_curr|v_next>; reference loop: Re<phi|v_next>
  ea.sums_dst = sm.p + 2;
  ctx->n_applies += Qb - 1;                                     // (Op::apply counts one)
  RC(op.apply(SD_C128, v_curr, v_prev, SD_EPI_RESCALE_DOT, ea));                                    // :106-107
  for (int n = 1; n < nsteps; ++n) {
    ea.prev = v_prev; ea.sums_dst = sm.p + 2 + 2 * n;
    ctx->n_applies += Qb - 1;
    RC(op.apply(SD_C128, v_next, v_curr, SD_EPI_KPM, ea));                                          // :111-117 fused
    double *t = v_prev; v_prev = v_curr; v_curr = v_next; v_next = t;                               // :124
  }
  std::vector<double> hs((size_t)Qb * (size_t)srow);
  SD_HIP(ctx, hipMemcpyAsync(hs.data(), sm.p, sizeof(double) * hs.size(), hipMemcpyDeviceToHost, ctx->stream));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int q = 0; q < Qb; ++q) {
    const double *h = hs.data() + (size_t)q * (size_t)srow;
    double *m = mu + (size_t)q * (size_t)M;
    m[0] = h[0];
    for (int k = 1; k <= nsteps; ++k) {                         // sums of step k: [s0, |v_k|^2]
      const double s0 = h[2 * (size_t)k], s1 = h[2 * (size_t)k + 1];
      if (doubling) {
        if (k == 1) m[1] = s0;
        else if (2 * k - 1 <= M - 1) m[2 * k - 1] = 2.0 * s0 - m[1];
        if (2 * k <= M - 1) m[2 * k] = 2.0 * s1 - m[0];
      } else {
        m[k] = s0;
      }
      if (!(std::sqrt(s1) <= 1e3)) ok[q] = 0;                   // the reference would have renormalised v_next here
    }
  }
  return SD_OK;
}

// the reference's break on beta_j < tol (src/Lanczos.jl:228-231; NaN counts as a break) applied to the read-back coefficients
void tridiag_trim(int mm, double tol, const double *al, const double *be, double *alpha, double *beta, int *m_eff_out) {
  int m_eff = mm;
  for (int j = 1; j <= mm - 1; ++j)
    if (!(be[j - 1] >= tol)) { m_eff = j; break; }
  for (int k = 0; k < mm; ++k) alpha[k] = k < m_eff ? al[k] : 0.0;
  for (int k = 0; k + 1 < mm; ++k) beta[k] = (m_eff < mm ? k < m_eff : k < mm - 1) ? be[k] : 0.0;
  *m_eff_out = m_eff;
}

int tridiag_dev(Op &op, double *vcur /* normalised start, consumed */, int lanc_m, double tol,
                double *alpha, double *beta, int *m_eff_out) {
  // lanczos_tridiag  src/Lanczos.jl:196-246 with two live vectors (the reference keeps all m).
  // No host round trip inside the loop: alpha_j and beta_j stay on the device (the update and normalisation passes read
  // them there and file them into d_al / d_be), so the whole recursion is queued at once and read back once.  The
  // reference's break on beta_j < tol (:228-231) is applied afterwards: the steps before it are unaffected by what was
  // queued behind them, the rest is discarded.
  sd_ctx *ctx = op.ctx;
  const int64_t n = op.n;
  const int mm = (int)std::min<int64_t>(lanc_m, op.m->N);
  if (lanczos_fused_ok(op, 1)) {                                           // launch-bound sizes: two launches per step
    std::vector<double> al, be;
    RC(lanczos_fused(op, 1, vcur, mm, 1, 0, tol, al, be));
    tridiag_trim(mm, tol, al.data(), be.data(), alpha, beta, m_eff_out);
    return SD_OK;
  }
  DBuf wb, vp, ab;
  RC(wb.alloc(ctx, 2 * n)); RC(vp.alloc(ctx, 2 * n)); RC(ab.alloc(ctx, 4 * (int64_t)mm + 2));
  double *d_al = ab.p, *d_be = ab.p + mm, *d_n2 = ab.p + 2 * (int64_t)mm;      // d_n2[2j]: |w_j|^2
  SD_HIP(ctx, hipMemsetAsync(ab.p, 0, sizeof(double) * (4 * (size_t)mm + 2), ctx->stream));
  sd_epi_args ea;
  std::vector<double> peek;
  // the vectors stay un-normalised (u_{j+1} = w_j, |w_j|^2 on the device): the update pass divides on the fly and the
  // normalising pass of :227/:233 disappears (k_lanczos_fold)
  double *ucur = vcur, *uprev = vp.p, *t = wb.p;
  const double *n2c = nullptr, *n2p = nullptr;
  for (int j = 1; j <= mm - 1; ++j) {
    RC(op.apply(SD_C128, t, ucur, SD_EPI_DOT, ea));                                                // :218-219 -> d_scalars[0]
    RC(op.reduce(ctx->d_scalars + 0, 2));
    double *n2o = d_n2 + 2 * (int64_t)j;
    RC(sd_k_lanczos_fold(ctx, t, ucur, j > 1 ? uprev : nullptr, n, 1, ctx->d_scalars + 0, n2c, n2p, d_al + (j - 1),
                         j > 1 ? d_be + (j - 2) : nullptr, n2o));                                  // :222-227
    RC(op.reduce(n2o, 1));
    { double *old = uprev; uprev = ucur; ucur = t; t = old; }
    n2p = n2c; n2c = n2o;
    if (j % SD_BREAK_PEEK == 0 && j < mm - 1) {       // bound the work queued behind a breakdown: look at the betas so far
      bool broke = false;
      RC(peek_breakdown(ctx, d_be, j - 1, tol, peek, &broke));
      if (broke) break;
    }
  }
  RC(op.apply(SD_C128, t, ucur, SD_EPI_DOT, ea));                                                  // :237-239
  RC(op.reduce(ctx->d_scalars + 0, 2));
  RC(sd_k_lanczos_fold_scalars(ctx, 1, ctx->d_scalars + 0, n2c, d_al + (mm - 1), mm > 1 ? d_be + (mm - 2) : nullptr));
  std::vector<double> host(2 * (size_t)mm);
  SD_HIP(ctx, hipMemcpyAsync(host.data(), ab.p, sizeof(double) * 2 * (size_t)mm, hipMemcpyDeviceToHost, ctx->stream));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  tridiag_trim(mm, tol, host.data(), host.data() + mm, alpha, beta, m_eff_out);
  return SD_OK;
}

}  // namespace

// --------------------------------------------------------------------------
// host numerics shared with the C ABI
// --------------------------------------------------------------------------

// Implicit-QL symmetric tridiagonal eigen-solver (stands in for LAPACK's
// eigvals/eigen(SymTridiagonal) at src/Lanczos.jl:80-83,164-165,
// src/TimeEvolution/Krylov.jl:175-176, src/LanczosSqw.jl:23-24).
extern "C" int sd_symtridiag_eig(int n, const double *d_in, const double *e_in, double *w, double *z) {
  if (n <= 0 || !d_in || !w) return SD_EARG;
  std::vector<double> d(d_in, d_in + n), e(n, 0.0);
  for (int i = 0; i + 1 < n; ++i) e[i] = e_in[i];
  if (z) { std::fill(z, z + (size_t)n * n, 0.0); for (int i = 0; i < n; ++i) z[i + (size_t)n * i] = 1.0; }
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; ++l) {
    int iter = 0, mm;
    do {
      for (mm = l; mm < n - 1; ++mm)
        if (std::fabs(e[mm]) <= eps * (std::fabs(d[mm]) + std::fabs(d[mm + 1]))) break;
      if (mm != l) {
        if (iter++ == 300) return SD_EINTERNAL;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[mm] - d[l] + e[l] / (g + std::copysign(r, g));
        double s = 1.0, c = 1.0, pp = 0.0;
        int i;
        bool underflow = false;
        for (i = mm - 1; i >= l; --i) {
          double f = s * e[i], b = c * e[i];
          r = std::hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) { d[i + 1] -= pp; e[mm] = 0.0; underflow = true; break; }
          s = f / r; c = g / r;
          g = d[i + 1] - pp;
          r = (d[i] - g) * s + 2.0 * c * b;
          pp = s * r;
          d[i + 1] = g + pp;
          g = c * r - b;
          if (z)
            for (int k = 0; k < n; ++k) {
              double *zi = z + (size_t)n * i, *zi1 = z + (size_t)n * (i + 1);
              const double f2 = zi1[k];
              zi1[k] = s * zi[k] + c * f2;
              zi[k] = c * zi[k] - s * f2;
            }
        }
        if (underflow) continue;
        d[l] -= pp; e[l] = g; e[mm] = 0.0;
      }
    } while (mm != l);
  }
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a] < d[b]; });
  std::vector<double> zc;
  if (z) zc.assign(z, z + (size_t)n * n);
  for (int k = 0; k < n; ++k) {
    w[k] = d[order[k]];
    if (z) std::memcpy(z + (size_t)n * k, zc.data() + (size_t)n * order[k], sizeof(double) * n);
  }
  return SD_OK;
}

// c_k = (2 - delta_k0) * (-i)^k * J_k(a dt) * exp(-i b dt)   (src/TimeEvolution/Chebyshev.jl:74-79)
extern "C" int sd_chebyshev_coeffs(int cheb_n, double a, double b, double dt, double *c) {
  if (cheb_n < 1 || !c) return SD_EARG;
  const double ph = b * dt, pr = std::cos(ph), pi = -std::sin(ph);
  for (int k = 0; k < cheb_n; ++k) {
    const double f = (k == 0) ? 1.0 : 2.0;
    const double J = std::cyl_bessel_j((double)k, a * dt);   // besselj(k, x), integer order
    double xr, xi;
    switch (k & 3) {
      case 0: xr = f; xi = 0; break;
      case 1: xr = 0; xi = -f; break;
      case 2: xr = -f; xi = 0; break;
      default: xr = 0; xi = f; break;
    }
    xr *= J; xi *= J;
    c[2 * k] = xr * pr - xi * pi;
    c[2 * k + 1] = xr * pi + xi * pr;
  }
  return SD_OK;
}

extern "C" int sd_kpm_kernel(int M, int kernel, double *g) {
  if (M < 1 || !g) return SD_EARG;
  const double PI = 3.14159265358979323846;
  for (int n = 0; n < M; ++n) g[n] = 1.0;
  if (kernel == SD_KERNEL_JACKSON) {
    for (int n = 0; n < M; ++n)
      g[n] = ((M - n + 1) * std::cos(PI * n / (M + 1)) + std::sin(PI * n / (M + 1)) * (1.0 / std::tan(PI / (M + 1)))) / (M + 1);
  } else if (kernel == SD_KERNEL_LORENTZ) {
    const double lam = 3.0;
    for (int n = 0; n < M; ++n) g[n] = std::sinh(lam * (1 - (double)n / M)) / std::sinh(lam);
  }
  return SD_OK;
}

extern "C" int sd_kpm_rescaling_from_bounds(double Emin, double Emax, double *a, double *b) {
  if (!a || !b) return SD_EARG;
  *a = (Emax - Emin) / (2 * 0.99);
  *b = (Emax + Emin) / 2;
  return SD_OK;
}

extern "C" int sd_kpm_reconstruct(const double *mu, int kpm_m, const double *omega, int W, double a, double b,
                                  double E0, double *S) {
  if (kpm_m < 1 || W < 0 || !mu || !S) return SD_EARG;
  const double PI = 3.14159265358979323846;
  std::vector<double> T(std::max(kpm_m, 2));
  for (int iw = 0; iw < W; ++iw) {
    const double x = (omega[iw] + E0 - b) / a;                       // src/KPM_Sqw.jl:61
    if (std::fabs(x) >= 1.0) { S[iw] = 0.0; continue; }
    T[0] = 1.0;
    if (kpm_m >= 2) T[1] = x;
    for (int n = 2; n < kpm_m; ++n) T[n] = 2.0 * x * T[n - 1] - T[n - 2];
    double sum_val = mu[0] * T[0];
    for (int n = 1; n < kpm_m; ++n) sum_val += 2.0 * mu[n] * T[n];
    const double denom = PI * std::sqrt(1.0 - x * x);
    const double v = sum_val / (a * denom);
    S[iw] = v > 0.0 ? v : 0.0;
  }
  return SD_OK;
}

extern "C" int sd_spectral_from_tridiagonal(const double *alpha, const double *beta, int mt, double norm_phi, double E0,
                                            const double *omega, int W, double eta, int broaden, double *S) {
  if (mt < 1 || !alpha || !S) return SD_EARG;
  if (broaden != SD_BROADEN_LORENTZ && broaden != SD_BROADEN_GAUSS) return SD_EARG;
  const double PI = 3.14159265358979323846;
  std::vector<double> th(mt), Q((size_t)mt * mt);
  int rc = sd_symtridiag_eig(mt, alpha, beta, th.data(), Q.data());
  if (rc) return rc;
  for (int iw = 0; iw < W; ++iw) {
    double s = 0.0;
    for (int k = 0; k < mt; ++k) {
      const double q1 = Q[(size_t)mt * k];
      const double wgt = q1 * q1 * (norm_phi * norm_phi);
      const double sh = omega[iw] - (th[k] - E0);
      const double f = broaden == SD_BROADEN_LORENTZ ? (1 / PI) * (eta / (sh * sh + eta * eta))
                                                     : (1 / (std::sqrt(2 * PI) * eta)) * std::exp(-(sh * sh) / (2 * eta * eta));
      s += f * wgt;
    }
    S[iw] = s;
  }
  return SD_OK;
}

// --------------------------------------------------------------------------
// recursion-level C ABI
// --------------------------------------------------------------------------

// Every entry point exists in two forms sharing one core: the unsharded form (host vectors in / out, comm == nullptr) and
// the sharded form (this rank's rows as device vectors, a communicator).
static int energy_bounds_core(Op &op, int lanc_m, const void *psi0_a, const void *psi0_b, bool on_dev, uint64_t seed,
                              double *Emin, double *Emax) {
  sd_ctx *ctx = op.ctx;
  double lo, hi;
  if (lanczos_fused_ok(op, 2)) {
    // launch-bound sizes: the run on H (:258) and the run on -H (:261-267) are independent recursions -- one batch of two vectors,
    // the second with the negated operator (bit 1 of the epilogue's negate mask); each sees the arithmetic of a run of its own
    const int64_t N = op.n;
    const int mm = (int)std::min<int64_t>(lanc_m, op.m->N);
    if (mm < 1) return sd_set_err(ctx, SD_EARG, "lanc_m must be >= 1");
    DBuf v; RC(v.alloc(ctx, 4 * N));
    RC(start_vector_op(op, v.p, psi0_a, on_dev, 2 * N, seed));
    RC(start_vector_op(op, v.p + 2 * N, psi0_b, on_dev, 2 * N, seed + 0x9E3779B97F4A7C15ULL));
    int rc = 0;
    for (int k = 0; k < 2; ++k) {
      const double nrm = norm_dev(op, v.p + 2 * N * k, 2 * N, &rc); RC(rc);
      RC(sd_k_scale_div(ctx, v.p + 2 * N * k, v.p + 2 * N * k, 2 * N, nrm));            // :40
    }
    std::vector<double> al, be;
    RC(lanczos_fused(op, 2, v.p, mm, 0, /*negate mask: vector 1*/ 2, 1e-12, al, be));
    for (int k = 0; k < 2; ++k) {
      const double *a = al.data() + (size_t)k * mm, *b = be.data() + (size_t)k * mm;
      int actual = mm;
      for (int j = 1; j < mm; ++j)
        if (!(b[j - 1] >= 1e-12)) { actual = j; break; }                                 // :66-70
      std::vector<double> ev(actual);
      if (sd_symtridiag_eig(actual, a, b, ev.data(), nullptr)) return sd_set_err(ctx, SD_EINTERNAL, "tridiagonal eigen-solver did not converge");
      if (k == 0) *Emax = ev[actual - 1]; else *Emin = -ev[actual - 1];
    }
    return SD_OK;
  }
  {
    DBuf v; RC(v.alloc(ctx, 2 * op.n));
    RC(start_vector_op(op, v.p, psi0_a, on_dev, 2 * op.n, seed));
    RC(extremal_dev(op, lanc_m, 1e-12, v.p, 0, &lo, &hi));              // src/Lanczos.jl:258
    *Emax = hi;
  }
  {
    DBuf v; RC(v.alloc(ctx, 2 * op.n));
    RC(start_vector_op(op, v.p, psi0_b, on_dev, 2 * op.n, seed + 0x9E3779B97F4A7C15ULL));
    RC(extremal_dev(op, lanc_m, 1e-12, v.p, 1, &lo, &hi));              // :261-267
    *Emin = -hi;
  }
  return SD_OK;
}

static int sd_lanczos_extremal_impl(sd_ctx *ctx, const sd_model *m, int lanc_m, double tol, const void *psi0,
                                   uint64_t seed, int negate, double *emin, double *emax) {
  Op op; RC(op.init(ctx, m, nullptr));
  if (!emin || !emax) return sd_set_err(ctx, SD_EARG, "null output");
  DBuf v; RC(v.alloc(ctx, 2 * op.n));
  RC(start_vector_op(op, v.p, psi0, false, 2 * op.n, seed));
  return extremal_dev(op, lanc_m, tol, v.p, negate, emin, emax);
}
extern "C" int sd_lanczos_extremal(sd_ctx *ctx, const sd_model *m, int lanc_m, double tol, const void *psi0,
                                   uint64_t seed, int negate, double *emin, double *emax) {
  SD_ABI_GUARD(ctx, sd_lanczos_extremal_impl(ctx, m, lanc_m, tol, psi0, seed, negate, emin, emax));
}

static int sd_lanczos_extremal_sharded_impl(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int lanc_m, double tol,
                                           const void *psi0_dev, uint64_t seed, int negate, double *emin, double *emax) {
  Op op; RC(op.init(ctx, m, comm));
  if (!emin || !emax) return sd_set_err(ctx, SD_EARG, "null output");
  DBuf v; RC(v.alloc(ctx, 2 * op.n));
  RC(start_vector_op(op, v.p, psi0_dev, true, 2 * op.n, seed));
  return extremal_dev(op, lanc_m, tol, v.p, negate, emin, emax);
}
extern "C" int sd_lanczos_extremal_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int lanc_m, double tol,
                                           const void *psi0_dev, uint64_t seed, int negate, double *emin, double *emax) {
  SD_ABI_GUARD(ctx, sd_lanczos_extremal_sharded_impl(ctx, m, comm, lanc_m, tol, psi0_dev, seed, negate, emin, emax));
}

static int sd_energy_bounds_impl(sd_ctx *ctx, const sd_model *m, int lanc_m, const void *psi0_a, const void *psi0_b,
                                uint64_t seed, double *Emin, double *Emax) {
  Op op; RC(op.init(ctx, m, nullptr));
  if (!Emin || !Emax) return sd_set_err(ctx, SD_EARG, "null output");
  return energy_bounds_core(op, lanc_m, psi0_a, psi0_b, false, seed, Emin, Emax);
}
extern "C" int sd_energy_bounds(sd_ctx *ctx, const sd_model *m, int lanc_m, const void *psi0_a, const void *psi0_b,
                                uint64_t seed, double *Emin, double *Emax) {
  SD_ABI_GUARD(ctx, sd_energy_bounds_impl(ctx, m, lanc_m, psi0_a, psi0_b, seed, Emin, Emax));
}

static int sd_energy_bounds_sharded_impl(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int lanc_m, uint64_t seed,
                                        double *Emin, double *Emax) {
  Op op; RC(op.init(ctx, m, comm));
  if (!Emin || !Emax) return sd_set_err(ctx, SD_EARG, "null output");
  return energy_bounds_core(op, lanc_m, nullptr, nullptr, true, seed, Emin, Emax);
}
extern "C" int sd_energy_bounds_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int lanc_m, uint64_t seed,
                                        double *Emin, double *Emax) {
  SD_ABI_GUARD(ctx, sd_energy_bounds_sharded_impl(ctx, m, comm, lanc_m, seed, Emin, Emax));
}

static int sd_apply_sharded_impl(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int dtype, void *out_dev, const void *psi_dev,
                                int64_t n_local, int overlap) {
  if (!ctx) return SD_EARG;
  if (!m || !out_dev || !psi_dev) return sd_set_err(ctx, SD_EARG, "null argument");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "dtype must be SD_F64 or SD_C128");
  if (out_dev == psi_dev) return sd_set_err(ctx, SD_EARG, "out must not alias psi");
  Op op; RC(op.init(ctx, m, comm));
  if (n_local != op.n) return sd_set_err(ctx, SD_EDIM, "vector length does not match the local basis dimension");
  op.overlap = overlap != 0;
  op.use_callback = false;      // operator level: always the built-in H
  sd_epi_args ea;
  RC(op.apply(dtype, (double *)out_dev, (const double *)psi_dev, SD_EPI_PLAIN, ea));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the halo / send buffers go back to the pool
  return SD_OK;
}
extern "C" int sd_apply_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int dtype, void *out_dev, const void *psi_dev,
                                int64_t n_local, int overlap) {
  SD_ABI_GUARD(ctx, sd_apply_sharded_impl(ctx, m, comm, dtype, out_dev, psi_dev, n_local, overlap));
}

static int sd_dot_sharded_impl(sd_ctx *ctx, sd_comm *comm, int dtype, const void *x, const void *y, int64_t n_local,
                              double *out2) {
  if (!ctx) return SD_EARG;
  if (!x || !y || !out2 || n_local < 0) return sd_set_err(ctx, SD_EARG, "bad argument");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "bad dtype");
  SD_HIP(ctx, hipSetDevice(ctx->device));
  RC(sd_k_dot(ctx, dtype == SD_C128 ? 2 : 1, (const double *)x, (const double *)y, n_local, 4));
  RC(sd_comm_allreduce_dev(ctx, comm, ctx->d_scalars + 4, 2));
  return sd_read_scalars(ctx, 4, 2, out2);
}
extern "C" int sd_dot_sharded(sd_ctx *ctx, sd_comm *comm, int dtype, const void *x, const void *y, int64_t n_local,
                              double *out2) {
  SD_ABI_GUARD(ctx, sd_dot_sharded_impl(ctx, comm, dtype, x, y, n_local, out2));
}

static int sd_lanczos_groundstate_impl(sd_ctx *ctx, const sd_model *m, int lanc_m, double tol, double orth_tol,
                                      const double *psi0, uint64_t seed, double *E0, double *psi_gs, int *m_actual_out) {
  Op op; RC(op.init(ctx, m, nullptr));
  if (!E0 || !psi_gs) return sd_set_err(ctx, SD_EARG, "null output");
  const int64_t N = m->N;
  const int mm = (int)std::min<int64_t>(lanc_m, N);
  if (mm < 1) return sd_set_err(ctx, SD_EARG, "lanc_m must be >= 1");
  DBuf V, w, tmp;
  RC(V.alloc(ctx, N * (int64_t)mm)); RC(w.alloc(ctx, N)); RC(tmp.alloc(ctx, N));
  if (psi0) RC(h2d(ctx, V.p, psi0, N));
  else RC(sd_k_fill_randn(ctx, V.p, N, seed, 0));
  int rc = 0;
  double nrm = norm_dev(op, V.p, N, &rc); RC(rc);
  RC(sd_k_scale_div(ctx, V.p, V.p, N, nrm));                                             // :100,105
  std::vector<double> alpha(mm, 0.0), beta(mm, 0.0);
  int m_actual = mm;
  // The reference's orthogonality check of step j (:142-153: dot(V[:,k], w/beta) for every k <= j) as the sequential loop
  // would run it, from column k0 on: corrects w and beta[j-1] when a test fires.  Returns 1 on breakdown (beta < tol).
  auto check_and_correct = [&](int j, double *wj, double *scratch_vec, int *broke) -> int {
    double s[2];
    std::vector<double> chk(j);
    *broke = 0;
    for (int k = 1; k <= j;) {
      RC(sd_k_scale_div(ctx, scratch_vec, wj, N, beta[j - 1]));
      RC(sd_k_mdot(ctx, V.p + N * (int64_t)(k - 1), N, j - k + 1, scratch_vec, N, chk.data()));
      int hit = -1;
      for (int q = 0; q < j - k + 1 && hit < 0; ++q)
        if (std::fabs(chk[q]) > orth_tol) hit = k + q;
      if (hit < 0) break;
      double *vk = V.p + N * (int64_t)(hit - 1);
      RC(sd_k_dot(ctx, 1, vk, wj, N, 4)); RC(sd_read_scalars(ctx, 4, 1, s));
      RC(sd_k_sub2(ctx, wj, vk, nullptr, N, s[0], 0.0));
      beta[j - 1] = norm_dev(op, wj, N, &rc); RC(rc);
      if (beta[j - 1] < tol) { *broke = 1; break; }                                       // inner break only (:150)
      k = hit + 1;
    }
    return SD_OK;
  };
  static const int gs_fused_env = getenv("SD_GS_FUSED") ? atoi(getenv("SD_GS_FUSED")) : 1;
  if (ctx->gs_blocked && !ctx->user_apply && gs_fused_env && mm >= 2) {
    // Passes that sum their producer's partial lists themselves (kernels_blas1.hip, k_gs_*): a step is the apply, one launch per
    // block of 8 columns, the update and the normalising pass -- no reduction launches, alpha_j and beta_j stay on the device --
    // and the orthogonality check of step j rides along with the Gram-Schmidt passes of step j + 1 (they read the same columns;
    // w_j / beta_j is V[:,j+1] itself), which removes a sweep over all columns per step.  ONE host synchronisation per step, to
    // look at the check (and beta) of the step before.  The one hit the reference's check loop finds on EVERY step -- the v_{j-1}
    // component that :129 puts back after the re-orthogonalisation had removed it -- is repaired inside the
    // update (k_gs_correct: w -= dot(v_{j-1}, w) v_{j-1}, then beta_j = norm(w), as :147-148).  If a check still fires -- orthogonality
    // lost beyond orth_tol, which the full re-orthogonalisation does not let happen in practice -- the step before is run through
    // the reference's loop itself and the current step is redone.
    const int nb = sd_k_gs_blocks(N);
    const size_t lst = (size_t)nb * 8;
    DBuf w2, scr, lists, ab, chkd;
    RC(w2.alloc(ctx, N));
    RC(scr.alloc(ctx, (int64_t)((size_t)(mm / 8 + 4) * lst)));
    RC(lists.alloc(ctx, (int64_t)(4 * lst)));                      // alpha partials | |w|^2 partials of the two latest steps | the update's pair
    RC(ab.alloc(ctx, 2 * (int64_t)mm + 2)); RC(chkd.alloc(ctx, (int64_t)mm + 16));
    SD_HIP(ctx, hipMemsetAsync(ab.p, 0, sizeof(double) * (2 * (size_t)mm + 2), ctx->stream));
    double *d_al = ab.p, *d_be = ab.p + mm, *apart = lists.p;
    auto n2list = [&](int j) { return lists.p + lst * (size_t)(1 + (j & 1)); };
    std::vector<double> hchk(mm + 16), hb(2);
    bool deferred = true;                 // false for the redo of a step whose predecessor was just checked the sequential way
    int j = 1;
    while (j <= mm) {
      double *vj = V.p + N * (int64_t)(j - 1);
      double *t = (j & 1) ? w.p : w2.p, *wprev = (j & 1) ? w2.p : w.p;     // w_{j-1} stays intact while step j runs
      RC(plain_op(ctx, m, SD_F64, t, vj));                                                // :113
      const bool chk_now = deferred && j >= 2;
      RC(sd_k_gs_chain(ctx, t, V.p, N, j, N, chk_now ? vj : nullptr, scr.p, apart, chkd.p));   // :116-124 (+ the check of step j-1)
      RC(sd_k_gs_update(ctx, t, vj, j == 1 ? nullptr : V.p + N * (int64_t)(j - 2), N, apart, n2list(j - 1), d_al + (j - 1),
                        j == 1 ? nullptr : d_be + (j - 2), n2list(j), lists.p + 3 * lst));  // :127-129, the k = j-1 repair (:142-148), |w|^2
      if (j < mm) RC(sd_k_gs_scale(ctx, V.p + N * (int64_t)j, t, N, n2list(j), d_be + (j - 1)));   // :133, :155
      if (j >= 2) {
        // the step before: beta_{j-1} (breakdown, :136-139) and its orthogonality check (:142-153)
        SD_HIP(ctx, hipMemcpyAsync(hb.data(), d_be + (j - 2), sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (chk_now) SD_HIP(ctx, hipMemcpyAsync(hchk.data(), chkd.p, sizeof(double) * (size_t)(j - 1), hipMemcpyDeviceToHost, ctx->stream));
        SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (deferred) beta[j - 2] = hb[0];
        if (!(beta[j - 2] >= tol)) { m_actual = j - 1; break; }                            // (NaN counts as a breakdown)
        bool fired = false;
        if (chk_now)
          for (int k = 0; k < j - 1 && !fired; ++k) fired = std::fabs(hchk[k]) > orth_tol;
        if (fired) {
          int broke = 0;
          RC(check_and_correct(j - 1, wprev, t, &broke));
          if (broke) { m_actual = j - 1; break; }
          RC(sd_k_scale_div(ctx, vj, wprev, N, beta[j - 2]));                              // :155 with the corrected w, beta
          // the device copy of |w_{j-1}|^2 (a one-entry list) and of beta_{j-1} follow the correction
          std::vector<double> one(lst, 0.0); one[0] = beta[j - 2] * beta[j - 2];
          SD_HIP(ctx, hipMemcpyAsync(n2list(j - 1), one.data(), sizeof(double) * lst, hipMemcpyHostToDevice, ctx->stream));
          SD_HIP(ctx, hipMemcpyAsync(d_be + (j - 2), &beta[j - 2], sizeof(double), hipMemcpyHostToDevice, ctx->stream));
          SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
          deferred = false;
          continue;                                                                        // redo step j on the corrected v_j
        }
      }
      deferred = true;
      ++j;
    }
    std::vector<double> host(2 * (size_t)mm);
    SD_HIP(ctx, hipMemcpyAsync(host.data(), ab.p, sizeof(double) * host.size(), hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < m_actual; ++k) alpha[k] = host[k];
    for (int k = 0; k + 1 < m_actual; ++k) beta[k] = host[mm + k];
  } else
  for (int j = 1; j <= mm; ++j) {
    double *vj = V.p + N * (int64_t)(j - 1);
    RC(plain_op(ctx, m, SD_F64, w.p, vj));                                                // :113
    double s[2];
    // :116-124: w -= dot(V[:,k], w) V[:,k] for k = 1..j-1, then alpha_j = dot(V[:,j], w) -- one chain of fused
    // subtract-and-dot kernels whose scalars stay on the device; only alpha_j comes back to the host
    // (default: in blocks of 8 columns, classical Gram-Schmidt inside a block -- the coefficients differ from the sequential
    // chain's by O(eps * |coeff|), half the passes over memory; sd_ctx_set_gs_blocked(ctx, 0): the reference's column order)
    if (ctx->gs_blocked) RC(sd_k_bgs_chain(ctx, w.p, V.p, N, j, N, 4));
    else RC(sd_k_mgs_chain(ctx, w.p, V.p, N, j, N, 4));
    RC(sd_read_scalars(ctx, 4, 1, s));
    alpha[j - 1] = s[0];                                                                 // :124
    RC(sd_k_sub2(ctx, w.p, vj, j == 1 ? nullptr : V.p + N * (int64_t)(j - 2), N, alpha[j - 1],
                 j == 1 ? 0.0 : beta[j - 2]));                                           // :127-129
    if (j < mm) {
      beta[j - 1] = norm_dev(op, w.p, N, &rc); RC(rc);                                   // :133
      if (beta[j - 1] < tol) { m_actual = j; break; }                                    // :136-139
      // :142-153: for k = 1..j test |dot(V[:,k], w/beta)| > orth_tol and correct w when it fires (then w/beta changes).
      // The tests between two corrections all use the same w/beta, so they are taken in one pass (sd_k_mdot) and the
      // first one that fires is handled exactly as the reference's sequential loop would; the rest is tested again.
      std::vector<double> chk(j);
      for (int k = 1; k <= j;) {
        RC(sd_k_scale_div(ctx, tmp.p, w.p, N, beta[j - 1]));
        RC(sd_k_mdot(ctx, V.p + N * (int64_t)(k - 1), N, j - k + 1, tmp.p, N, chk.data()));
        int hit = -1;
        for (int q = 0; q < j - k + 1 && hit < 0; ++q)
          if (std::fabs(chk[q]) > orth_tol) hit = k + q;
        if (hit < 0) break;
        double *vk = V.p + N * (int64_t)(hit - 1);
        RC(sd_k_dot(ctx, 1, vk, w.p, N, 4)); RC(sd_read_scalars(ctx, 4, 1, s));
        RC(sd_k_sub2(ctx, w.p, vk, nullptr, N, s[0], 0.0));
        beta[j - 1] = norm_dev(op, w.p, N, &rc); RC(rc);
        if (beta[j - 1] < tol) { m_actual = j; break; }                                 // inner break only (:150)
        k = hit + 1;
      }
      RC(sd_k_scale_div(ctx, V.p + N * (int64_t)j, w.p, N, beta[j - 1]));                 // :155
    }
  }
  std::vector<double> ev(m_actual), Z((size_t)m_actual * m_actual);
  if (sd_symtridiag_eig(m_actual, alpha.data(), beta.data(), ev.data(), Z.data()))       // :164-165
    return sd_set_err(ctx, SD_EINTERNAL, "tridiagonal eigen-solver did not converge");
  *E0 = ev[0];                                                                           // :167
  RC(sd_k_gemv_cols(ctx, w.p, V.p, N, m_actual, Z.data()));                              // :170 (first eigenvector = column 0)
  nrm = norm_dev(op, w.p, N, &rc); RC(rc);
  RC(sd_k_scale_div(ctx, w.p, w.p, N, nrm));                                             // :171
  RC(d2h(ctx, psi_gs, w.p, N));
  if (m_actual_out) *m_actual_out = m_actual;
  return SD_OK;
}
extern "C" int sd_lanczos_groundstate(sd_ctx *ctx, const sd_model *m, int lanc_m, double tol, double orth_tol,
                                      const double *psi0, uint64_t seed, double *E0, double *psi_gs, int *m_actual_out) {
  SD_ABI_GUARD(ctx, sd_lanczos_groundstate_impl(ctx, m, lanc_m, tol, orth_tol, psi0, seed, E0, psi_gs, m_actual_out));
}

static int sd_lanczos_tridiag_impl(sd_ctx *ctx, const sd_model *m, const void *v, int64_t n, int lanc_m, double tol,
                                  double *alpha, double *beta, int *m_eff, double *norm_v) {
  Op op; RC(op.init(ctx, m, nullptr));
  if (n != m->N) return sd_set_err(ctx, SD_EDIM, "vector length does not match the basis dimension");
  if (!v || !alpha || !beta || !m_eff || !norm_v) return sd_set_err(ctx, SD_EARG, "null argument");
  if (lanc_m < 1) return sd_set_err(ctx, SD_EARG, "lanc_m must be >= 1");
  DBuf vc; RC(vc.alloc(ctx, 2 * n));
  RC(h2d(ctx, vc.p, v, 2 * n));
  int rc = 0;
  const double normv = norm_dev(op, vc.p, 2 * n, &rc); RC(rc);
  if (normv == 0) return sd_set_err(ctx, SD_EZERO, "starting vector has zero norm");     // :210-212
  RC(sd_k_scale_div(ctx, vc.p, vc.p, 2 * n, normv));
  *norm_v = normv;
  return tridiag_dev(op, vc.p, lanc_m, tol, alpha, beta, m_eff);
}
extern "C" int sd_lanczos_tridiag(sd_ctx *ctx, const sd_model *m, const void *v, int64_t n, int lanc_m, double tol,
                                  double *alpha, double *beta, int *m_eff, double *norm_v) {
  SD_ABI_GUARD(ctx, sd_lanczos_tridiag_impl(ctx, m, v, n, lanc_m, tol, alpha, beta, m_eff, norm_v));
}

// krylov_time_evolve; on_dev: psi0 / psit are device vectors (psit ComplexF64; may alias a ComplexF64 psi0)
static int krylov_evolve_core(Op &op, int dtype, const void *psi0, int64_t n, double dt, int kry_m, void *psit, bool on_dev) {
  sd_ctx *ctx = op.ctx;
  if (!psi0 || !psit) return sd_set_err(ctx, SD_EARG, "null vector");
  auto emit = [&](const double *src) -> int {          // result to the caller
    if (!on_dev) return d2h(ctx, psit, src, 2 * n);
    RC(d2d(ctx, (double *)psit, src, 2 * n));
    SD_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the work vectors go back to the pool: nothing may still use them
    return SD_OK;
  };
  if (n != op.n) return sd_set_err(ctx, SD_EDIM, "vector length does not match the (local) basis dimension");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "bad dtype");
  if (kry_m < 1) return sd_set_err(ctx, SD_EARG, "kry_m must be >= 1");
  const int nc = dtype == SD_C128 ? 2 : 1;
  // all Krylov vectors are kept complex on the device; a real psi0 keeps exactly-zero imaginary parts
  std::vector<DBuf> V(kry_m);
  DBuf in, w;
  RC(w.alloc(ctx, 2 * n));
  const double *inp = (const double *)psi0;
  if (!on_dev) { RC(in.alloc(ctx, nc * n)); RC(h2d(ctx, in.p, psi0, nc * n)); inp = in.p; }
  int rc = 0;
  const double norm0 = norm_dev(op, inp, nc * n, &rc); RC(rc);
  RC(V[0].alloc(ctx, 2 * n));
  RC(sd_k_promote(ctx, V[0].p, inp, nc, n));
  if (norm0 == 0) return emit(V[0].p);                                                    // :145-147
  RC(sd_k_scale_div(ctx, V[0].p, V[0].p, 2 * n, norm0));                                  // :148
  // the Lanczos part is queued without host round trips (see tridiag_dev): alpha_j (complex) and beta_j stay on the device
  // and are read back once; the break on |beta_j| < 1e-14 (:162-168) is applied to the values afterwards
  // V[j] holds the un-normalised u_{j+1} = w_j (k_lanczos_fold): no normalising pass; the final combination divides its
  // coefficients by beta_j instead
  DBuf ab; RC(ab.alloc(ctx, 5 * (int64_t)kry_m + 2));
  double *d_al = ab.p, *d_be = ab.p + 2 * (int64_t)kry_m, *d_n2 = ab.p + 3 * (int64_t)kry_m;   // alpha as (re, im) pairs, beta, |w_j|^2 at [2j]
  SD_HIP(ctx, hipMemsetAsync(ab.p, 0, sizeof(double) * (5 * (size_t)kry_m + 2), ctx->stream));
  sd_epi_args ea;
  const double *n2c = nullptr, *n2p = nullptr;
  const bool fused = lanczos_fused_ok(op, 1);          // launch-bound sizes: two launches per step (see lanczos_fused)
  DBuf n2l;
  const int nt = op.m->dm.n_singles, nbf = sd_k_lanczos_fold_blocks(n);
  if (fused) { RC(n2l.alloc(ctx, 3 * 2 * (int64_t)nbf)); ea.no_reduce = 1; }
  auto n2buf = [&](int j) { return n2l.p + (size_t)(j % 3) * 2 * (size_t)nbf; };
  for (int j = 1; fused && j <= kry_m; ++j) {
    double *t = w.p;
    if (j < kry_m) { RC(V[j].alloc(ctx, 2 * n)); t = V[j].p; }
    RC(op.apply(SD_C128, t, V[j - 1].p, SD_EPI_DOT, ea));                                 // per-tile pairs of <u|Hu> (re, im) -> ctx->d_partials
    const double *pc = j > 1 ? n2buf(j - 1) : nullptr, *pp = j > 2 ? n2buf(j - 2) : nullptr;
    if (j == kry_m) {
      RC(sd_k_lanczos_fold_scalars_p(ctx, 1, 2, ctx->d_partials, nt, pc, nbf, d_al + 2 * (j - 1), j > 1 ? d_be + (j - 2) : nullptr, 0));
      break;
    }
    RC(sd_k_lanczos_fold_p(ctx, t, V[j - 1].p, j > 1 ? V[j - 2].p : nullptr, n, 1, n, 2, ctx->d_partials, nt, pc, pp, nbf,
                           d_al + 2 * (j - 1), j > 1 ? d_be + (j - 2) : nullptr, 0, n2buf(j), nbf));   // :156-161
  }
  for (int j = 1; !fused && j <= kry_m; ++j) {
    double *t = w.p;
    if (j < kry_m) { RC(V[j].alloc(ctx, 2 * n)); t = V[j].p; }
    RC(op.apply(SD_C128, t, V[j - 1].p, SD_EPI_DOT, ea));                                 // :153,155 -> d_scalars[0..1]
    RC(op.reduce(ctx->d_scalars + 0, 2));
    if (j == kry_m) {
      RC(sd_k_lanczos_fold_scalars(ctx, 2, ctx->d_scalars + 0, n2c, d_al + 2 * (j - 1), j > 1 ? d_be + (j - 2) : nullptr));
      break;
    }
    double *n2o = d_n2 + 2 * (int64_t)j;
    RC(sd_k_lanczos_fold(ctx, t, V[j - 1].p, j > 1 ? V[j - 2].p : nullptr, n, 2, ctx->d_scalars + 0, n2c, n2p,
                         d_al + 2 * (j - 1), j > 1 ? d_be + (j - 2) : nullptr, n2o));      // :156-161
    RC(op.reduce(n2o, 1));
    n2p = n2c; n2c = n2o;
  }
  std::vector<double> hostab(3 * (size_t)kry_m);
  SD_HIP(ctx, hipMemcpyAsync(hostab.data(), ab.p, sizeof(double) * 3 * (size_t)kry_m, hipMemcpyDeviceToHost, ctx->stream));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<double> alr(kry_m, 0.0), beta(kry_m, 0.0);
  int m_eff = kry_m;
  for (int j = 1; j <= kry_m; ++j) {
    alr[j - 1] = hostab[2 * (size_t)(j - 1)];
    if (j < kry_m) {
      beta[j - 1] = hostab[2 * (size_t)kry_m + (j - 1)];
      if (!(std::fabs(beta[j - 1]) >= 1e-14)) { m_eff = j; break; }                       // :162-168
    }
  }
  // reduced problem on the host (:175-182).  Deviation (documented in DESIGN.md): Re(alpha) enters a
  // symmetric tridiagonal solve instead of the reference's general complex eigen of the same matrix.
  std::vector<double> ev(m_eff), Q((size_t)m_eff * m_eff), yr(m_eff, 0.0), yi(m_eff, 0.0);
  if (sd_symtridiag_eig(m_eff, alr.data(), beta.data(), ev.data(), Q.data()))
    return sd_set_err(ctx, SD_EINTERNAL, "tridiagonal eigen-solver did not converge");
  for (int l = 0; l < m_eff; ++l) {
    const double ph = -ev[l] * dt, cr = std::cos(ph), ci = std::sin(ph);
    const double q0 = Q[(size_t)m_eff * l] * norm0;
    for (int k = 0; k < m_eff; ++k) {
      const double qk = Q[k + (size_t)m_eff * l];
      yr[k] += qk * cr * q0; yi[k] += qk * ci * q0;
    }
  }
  {                                                                                       // :185-188, one fused pass
    std::vector<const double *> cols(m_eff);
    for (int k = 0; k < m_eff; ++k) cols[k] = V[k].p;
    for (int k = 1; k < m_eff; ++k) { yr[k] /= beta[k - 1]; yi[k] /= beta[k - 1]; }     // V[k] = beta_k v_{k+1}
    RC(sd_k_ccombine(ctx, w.p, cols.data(), n, m_eff, yr.data(), yi.data()));
  }
  const double nn = norm_dev(op, w.p, 2 * n, &rc); RC(rc);
  RC(sd_k_scale_div(ctx, w.p, w.p, 2 * n, nn));                                           // :190
  return emit(w.p);
}

static int sd_krylov_evolve_impl(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi0, int64_t n, double dt,
                                int kry_m, void *psit) {
  Op op; RC(op.init(ctx, m, nullptr));
  return krylov_evolve_core(op, dtype, psi0, n, dt, kry_m, psit, false);
}
extern "C" int sd_krylov_evolve(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi0, int64_t n, double dt,
                                int kry_m, void *psit) {
  SD_ABI_GUARD(ctx, sd_krylov_evolve_impl(ctx, m, dtype, psi0, n, dt, kry_m, psit));
}

static int sd_krylov_evolve_dev_impl(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi0_dev, int64_t n, double dt,
                                    int kry_m, void *psit_dev) {
  Op op; RC(op.init(ctx, m, nullptr));
  return krylov_evolve_core(op, dtype, psi0_dev, n, dt, kry_m, psit_dev, true);
}
extern "C" int sd_krylov_evolve_dev(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi0_dev, int64_t n, double dt,
                                    int kry_m, void *psit_dev) {
  SD_ABI_GUARD(ctx, sd_krylov_evolve_dev_impl(ctx, m, dtype, psi0_dev, n, dt, kry_m, psit_dev));
}

static int sd_krylov_evolve_sharded_impl(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int dtype, const void *psi0_dev,
                                        int64_t n_local, double dt, int kry_m, void *psit_dev) {
  Op op; RC(op.init(ctx, m, comm));
  return krylov_evolve_core(op, dtype, psi0_dev, n_local, dt, kry_m, psit_dev, true);
}
extern "C" int sd_krylov_evolve_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int dtype, const void *psi0_dev,
                                        int64_t n_local, double dt, int kry_m, void *psit_dev) {
  SD_ABI_GUARD(ctx, sd_krylov_evolve_sharded_impl(ctx, m, comm, dtype, psi0_dev, n_local, dt, kry_m, psit_dev));
}

// chebyshev_time_evolve on device vectors: psi0_dev (c128, n elements) is read, psit_dev receives psi(t); they may be the
// same buffer (psi0 is copied into the recursion's own vectors first).  host_in / host_out select the host-pointer form.
static int chebyshev_evolve_core(Op &op, const void *psi0, bool host_in, int64_t n, double dt, int cheb_n,
                                 double Emin, double Emax, void *psit, bool host_out) {
  sd_ctx *ctx = op.ctx;
  if (!psi0 || !psit) return sd_set_err(ctx, SD_EARG, "null vector");
  if (n != op.n) return sd_set_err(ctx, SD_EDIM, "vector length does not match the (local) basis dimension");
  if (cheb_n < 1) return sd_set_err(ctx, SD_EARG, "cheb_n must be >= 1");               // :65
  const double a = (Emax - Emin) / (2 * 0.9999), b = (Emax + Emin) / 2;                   // :70-71
  std::vector<double> c(2 * (size_t)cheb_n);
  sd_chebyshev_coeffs(cheb_n, a, b, dt, c.data());
  DBuf b0, b1, b2, ptb;
  RC(b0.alloc(ctx, 2 * n)); RC(b1.alloc(ctx, 2 * n)); RC(b2.alloc(ctx, 2 * n));
  double *pprev = b0.p, *pcur = b1.p, *pnext = b2.p;
  if (host_in) RC(h2d(ctx, pprev, psi0, 2 * n));                                          // :90
  else RC(d2d(ctx, pprev, (const double *)psi0, 2 * n));
  struct { double *p; } pt;                       // psi_t accumulates in the caller's device buffer when there is one
  if (host_out) { RC(ptb.alloc(ctx, 2 * n)); pt.p = ptb.p; }
  else pt.p = (double *)psit;
  sd_epi_args ea; ea.a = a; ea.b = b;
  RC(op.apply(SD_C128, pcur, pprev, SD_EPI_RESCALE, ea));                                 // :93
  RC(sd_k_cheb_init(ctx, pt.p, pprev, pcur, n, c[0], c[1], cheb_n >= 2 ? c[2] : 0.0, cheb_n >= 2 ? c[3] : 0.0,
                    cheb_n >= 2));                                                        // :96-102
  // :110-121, one fused pass per term.  Terms are taken in pairs: the first of a pair only advances the recurrence, the
  // second adds both terms to psi_t in order (c_k phi_k is exact in the apply's input vector), so psi_t is read and
  // written once per two terms -- same bits as one accumulation per term, 72 instead of 80 B/row per term.
  int k = 2;
  ea.accv = pt.p;
  if ((cheb_n - 2) % 2 == 1) {
    ea.prev = pprev; ea.c_re = c[2 * k]; ea.c_im = c[2 * k + 1];
    RC(op.apply(SD_C128, pnext, pcur, SD_EPI_CHEB, ea));
    double *t = pprev; pprev = pcur; pcur = pnext; pnext = t;
    ++k;
  }
  for (; k + 1 <= cheb_n - 1; k += 2) {
    ea.prev = pprev;
    RC(op.apply(SD_C128, pnext, pcur, SD_EPI_RECUR, ea));                                 // phi_k
    { double *t = pprev; pprev = pcur; pcur = pnext; pnext = t; }
    ea.prev = pprev; ea.c0_re = c[2 * k]; ea.c0_im = c[2 * k + 1]; ea.c_re = c[2 * k + 2]; ea.c_im = c[2 * k + 3];
    RC(op.apply(SD_C128, pnext, pcur, SD_EPI_CHEB2, ea));                                 // phi_{k+1}; psi_t += c_k phi_k + c_{k+1} phi_{k+1}
    { double *t = pprev; pprev = pcur; pcur = pnext; pnext = t; }
  }
  if (host_out) RC(d2h(ctx, psit, pt.p, 2 * n));
  else SD_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the work vectors go back to the pool: nothing may still use them
  return SD_OK;
}

static int sd_chebyshev_evolve_impl(sd_ctx *ctx, const sd_model *m, const void *psi0, int64_t n, double dt, int cheb_n,
                                   double Emin, double Emax, void *psit) {
  Op op; RC(op.init(ctx, m, nullptr));
  return chebyshev_evolve_core(op, psi0, true, n, dt, cheb_n, Emin, Emax, psit, true);
}
extern "C" int sd_chebyshev_evolve(sd_ctx *ctx, const sd_model *m, const void *psi0, int64_t n, double dt, int cheb_n,
                                   double Emin, double Emax, void *psit) {
  SD_ABI_GUARD(ctx, sd_chebyshev_evolve_impl(ctx, m, psi0, n, dt, cheb_n, Emin, Emax, psit));
}

static int sd_chebyshev_evolve_dev_impl(sd_ctx *ctx, const sd_model *m, const void *psi0_dev, int64_t n, double dt, int cheb_n,
                                       double Emin, double Emax, void *psit_dev) {
  Op op; RC(op.init(ctx, m, nullptr));
  return chebyshev_evolve_core(op, psi0_dev, false, n, dt, cheb_n, Emin, Emax, psit_dev, false);
}
extern "C" int sd_chebyshev_evolve_dev(sd_ctx *ctx, const sd_model *m, const void *psi0_dev, int64_t n, double dt, int cheb_n,
                                       double Emin, double Emax, void *psit_dev) {
  SD_ABI_GUARD(ctx, sd_chebyshev_evolve_dev_impl(ctx, m, psi0_dev, n, dt, cheb_n, Emin, Emax, psit_dev));
}

static int sd_chebyshev_evolve_sharded_impl(sd_ctx *ctx, const sd_model *m, sd_comm *comm, const void *psi0_dev,
                                           int64_t n_local, double dt, int cheb_n, double Emin, double Emax, void *psit_dev) {
  Op op; RC(op.init(ctx, m, comm));
  return chebyshev_evolve_core(op, psi0_dev, false, n_local, dt, cheb_n, Emin, Emax, psit_dev, false);
}
extern "C" int sd_chebyshev_evolve_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, const void *psi0_dev,
                                           int64_t n_local, double dt, int cheb_n, double Emin, double Emax, void *psit_dev) {
  SD_ABI_GUARD(ctx, sd_chebyshev_evolve_sharded_impl(ctx, m, comm, psi0_dev, n_local, dt, cheb_n, Emin, Emax, psit_dev));
}

static int sd_kpm_moments_impl(sd_ctx *ctx, const sd_model *m, const void *phi, int64_t n, int M, double a, double b,
                              double *mu) {
  Op op; RC(op.init(ctx, m, nullptr));
  if (n != m->N) return sd_set_err(ctx, SD_EDIM, "vector length does not match the basis dimension");
  if (!phi || !mu) return sd_set_err(ctx, SD_EARG, "null argument");
  DBuf ph; RC(ph.alloc(ctx, 2 * n));
  RC(h2d(ctx, ph.p, phi, 2 * n));
  return moments_dev(op, ph.p, M, a, b, mu);
}
extern "C" int sd_kpm_moments(sd_ctx *ctx, const sd_model *m, const void *phi, int64_t n, int M, double a, double b,
                              double *mu) {
  SD_ABI_GUARD(ctx, sd_kpm_moments_impl(ctx, m, phi, n, M, a, b, mu));
}

static int sd_kpm_moments_sharded_impl(sd_ctx *ctx, const sd_model *m, sd_comm *comm, const void *phi_dev, int64_t n_local,
                                      int M, double a, double b, double *mu) {
  Op op; RC(op.init(ctx, m, comm));
  if (n_local != op.n) return sd_set_err(ctx, SD_EDIM, "vector length does not match the local basis dimension");
  if (!phi_dev || !mu) return sd_set_err(ctx, SD_EARG, "null argument");
  RC(moments_dev(op, (const double *)phi_dev, M, a, b, mu));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SD_OK;
}
extern "C" int sd_kpm_moments_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, const void *phi_dev, int64_t n_local,
                                      int M, double a, double b, double *mu) {
  SD_ABI_GUARD(ctx, sd_kpm_moments_sharded_impl(ctx, m, comm, phi_dev, n_local, M, a, b, mu));
}

// H is real in the S^z basis, so for a real psi0 phi_{2pi-q} = conj(phi_q): the moments mu_n and the Lanczos coefficients
// alpha_j, beta_j of q and of 2pi - q agree, and so do the rows S(q, w).  same_as[j] = i < j when q[j] = 2 pi k - q[i] (to 8 ulp)
// and psi0 is real -- Float64, or ComplexF64 whose imaginary parts are all zero (counted on the device, summed over the ranks) --,
// else -1.  momenta(model) always holds both members of a pair; the reference recomputes the second row (src/KPM_Sqw.jl:218-252,
// src/LanczosSqw.jl:63-77), which differs from the copy by the rounding of exp(iqr) only.  Not with a caller's
// operator (it need not be real); sd_ctx_set_kpm_pair_q(ctx, 0) switches it off.
int pair_momenta(Op &op, int dtype, const double *psic, int64_t n, const double *q, int Qn, std::vector<int> &same_as) {
  sd_ctx *ctx = op.ctx;
  same_as.assign((size_t)std::max(Qn, 0), -1);
  if (!ctx->kpm_pair_q || ctx->user_apply || Qn < 2) return SD_OK;
  bool real_psi = dtype == SD_F64;
  if (!real_psi) {
    RC(sd_k_imag_count(ctx, psic, n, 6));
    RC(op.reduce(ctx->d_scalars + 6, 1));
    double cnt[1]; RC(sd_read_scalars(ctx, 6, 1, cnt));
    real_psi = cnt[0] == 0.0;
  }
  if (!real_psi) return SD_OK;
  const double two_pi = 6.283185307179586476925286766559;
  for (int j = 1; j < Qn; ++j)
    for (int i = 0; i < j; ++i) {
      if (same_as[i] >= 0) continue;
      const double sum = q[i] + q[j], k = std::nearbyint(sum / two_pi);
      const double tol = 8 * 2.220446049250313e-16 * std::max(two_pi, std::max(std::fabs(q[i]), std::fabs(q[j])));
      if (std::fabs(sum - k * two_pi) <= tol) { same_as[j] = i; break; }
    }
  return SD_OK;
}

// kpm_sqw (src/KPM_Sqw.jl:191-256); psi0: host vector (unsharded form) or this rank's rows on the device
static int kpm_sqw_core(Op &op, int dtype, const void *psi0, bool on_dev, int64_t n, const double *q, int Qn,
                        const double *omega, int W, int have_ab, double a, double b, int kpm_m, int kernel, uint64_t seed,
                        double *Smat) {
  sd_ctx *ctx = op.ctx;
  const sd_model *m = op.m;
  if (n != op.n) return sd_set_err(ctx, SD_EDIM, "vector length does not match the (local) basis dimension");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "bad dtype");
  if (kpm_m < 2) return sd_set_err(ctx, SD_EARG, "kpm_m must be >= 2");
  if (!psi0 || !Smat || (Qn > 0 && !q) || (W > 0 && !omega)) return sd_set_err(ctx, SD_EARG, "null argument");
  const int nc = dtype == SD_C128 ? 2 : 1;
  DBuf in, psic, phi;
  RC(psic.alloc(ctx, 2 * n)); RC(phi.alloc(ctx, 2 * n));
  const double *inp = (const double *)psi0;
  if (!on_dev) { RC(in.alloc(ctx, nc * n)); RC(h2d(ctx, in.p, psi0, nc * n)); inp = in.p; }
  RC(sd_k_promote(ctx, psic.p, inp, nc, n));                                              // :202
  in.release();
  sd_epi_args ea;
  double s[2];
  // phi doubles as the scratch for H psi0 (only <psi0|H psi0> is kept): the recursion then holds psi0, phi and its three
  // work vectors -- five vectors of n elements plus halo and send buffer on a shard
  RC(op.apply(SD_C128, phi.p, psic.p, SD_EPI_DOT, ea));                                   // :208-209
  RC(op.reduce(ctx->d_scalars + 0, 2));
  RC(sd_read_scalars(ctx, 0, 2, s));
  const double E0 = s[0];
  if (!have_ab) {                                                                         // :212-214
    double Emin, Emax;
    RC(energy_bounds_core(op, 80, nullptr, nullptr, true, seed, &Emin, &Emax));
    sd_kpm_rescaling_from_bounds(Emin, Emax, &a, &b);
  }
  std::vector<double> mu(kpm_m), g(kpm_m);
  sd_kpm_kernel(kpm_m, kernel, g.data());
  int rc = 0;
  // every pair (q, 2 pi - q) of the list is computed once for a real psi0 (pair_momenta)
  std::vector<int> same_as;
  RC(pair_momenta(op, dtype, psic.p, n, q, Qn, same_as));
  // Launch-bound sizes: the momenta's vectors share their launches (moments_dev_batched).  sd_ctx_set_q_batch(ctx, 0): one momentum at a time.
  {
    std::vector<int> live;
    for (int iq = 0; iq < Qn; ++iq) if (same_as[iq] < 0) live.push_back(iq);
    const int q_batch = ctx->q_batch;
    const int Qb = (int)live.size();
    // five batches of vectors (phi + the recursion's three + nothing else) within 4 GiB, vectors of at most 2^22 rows
    const bool batched = q_batch && Qb >= 2 && m->nranks == 1 && !ctx->user_apply && m->p >= 0 && m->dm.n_singles <= 16384 &&
                         n <= ((int64_t)1 << 22) && (int64_t)Qb * n * 16 * 4 <= ((int64_t)4 << 30);
    if (batched) {
      DBuf phib, nrm;
      RC(phib.alloc(ctx, 2 * n * Qb)); RC(nrm.alloc(ctx, 2 * (int64_t)Qb));
      for (int k = 0; k < Qb; ++k) {
        RC(sd_launch_szq(ctx, m, SD_C128, psic.p, q[live[k]], phib.p + 2 * n * k));       // :223
        RC(sd_k_nrm2sq_to(ctx, phib.p + 2 * n * k, 2 * n, nrm.p + 2 * k));
      }
      std::vector<double> hn(2 * (size_t)Qb);
      SD_HIP(ctx, hipMemcpyAsync(hn.data(), nrm.p, sizeof(double) * hn.size(), hipMemcpyDeviceToHost, ctx->stream));
      SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
      // zero vectors (:226-229) drop out of the batch; the rest are normalised in place (:231)
      std::vector<int> kept; std::vector<double> norms;
      for (int k = 0; k < Qb; ++k) {
        const double nphi = std::sqrt(hn[2 * (size_t)k]);
        double *Srow = Smat + (size_t)live[k] * W;
        if (nphi == 0) { for (int iw = 0; iw < W; ++iw) Srow[iw] = 0.0; continue; }
        const int dst = (int)kept.size();
        if (dst != k) RC(d2d(ctx, phib.p + 2 * n * dst, phib.p + 2 * n * k, 2 * n));
        RC(sd_k_scale_div(ctx, phib.p + 2 * n * dst, phib.p + 2 * n * dst, 2 * n, nphi));
        kept.push_back(live[k]); norms.push_back(nphi);
      }
      const int Qk = (int)kept.size();
      if (Qk > 0) {
        std::vector<double> mub((size_t)Qk * (size_t)kpm_m);
        std::vector<char> okv;
        RC(moments_dev_batched(op, phib.p, Qk, kpm_m, a, b, mub.data(), okv));
        for (int k = 0; k < Qk; ++k) {
          double *Srow = Smat + (size_t)kept[k] * W;
          double *muk = mub.data() + (size_t)k * (size_t)kpm_m;
          if (!okv[k]) RC(moments_dev(op, phib.p + 2 * n * k, kpm_m, a, b, muk));          // the guard fired: the reference's loop
          for (int j = 0; j < kpm_m; ++j) muk[j] *= g[j];                                 // :53
          sd_kpm_reconstruct(muk, kpm_m, omega, W, a, b, E0, Srow);
          const double n2 = norms[k] * norms[k];
          for (int iw = 0; iw < W; ++iw) Srow[iw] *= n2;                                  // :252
        }
      }
      for (int iq = 0; iq < Qn; ++iq)
        if (same_as[iq] >= 0) std::memcpy(Smat + (size_t)iq * W, Smat + (size_t)same_as[iq] * W, sizeof(double) * (size_t)W);
      SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return SD_OK;
    }
  }
  for (int iq = 0; iq < Qn; ++iq) {                                                       // :218 (serial over q)
    double *Srow = Smat + (size_t)iq * W;
    if (same_as[iq] >= 0) {
      std::memcpy(Srow, Smat + (size_t)same_as[iq] * W, sizeof(double) * (size_t)W);
      continue;
    }
    RC(sd_launch_szq(ctx, m, SD_C128, psic.p, q[iq], phi.p));                             // :223
    const double norm_phi = norm_dev(op, phi.p, 2 * n, &rc); RC(rc);
    if (norm_phi == 0) { for (int iw = 0; iw < W; ++iw) Srow[iw] = 0.0; continue; }       // :226-229
    RC(sd_k_scale_div(ctx, phi.p, phi.p, 2 * n, norm_phi));                               // :231
    RC(moments_dev(op, phi.p, kpm_m, a, b, mu.data()));
    for (int k = 0; k < kpm_m; ++k) mu[k] *= g[k];                                        // :53
    sd_kpm_reconstruct(mu.data(), kpm_m, omega, W, a, b, E0, Srow);
    const double n2 = norm_phi * norm_phi;
    for (int iw = 0; iw < W; ++iw) Srow[iw] *= n2;                                        // :252
  }
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SD_OK;
}

static int sd_kpm_sqw_impl(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi0, int64_t n, const double *q, int Qn,
                          const double *omega, int W, int have_ab, double a, double b, int kpm_m, int kernel,
                          uint64_t seed, double *Smat) {
  Op op; RC(op.init(ctx, m, nullptr));
  return kpm_sqw_core(op, dtype, psi0, false, n, q, Qn, omega, W, have_ab, a, b, kpm_m, kernel, seed, Smat);
}
extern "C" int sd_kpm_sqw(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi0, int64_t n, const double *q, int Qn,
                          const double *omega, int W, int have_ab, double a, double b, int kpm_m, int kernel,
                          uint64_t seed, double *Smat) {
  SD_ABI_GUARD(ctx, sd_kpm_sqw_impl(ctx, m, dtype, psi0, n, q, Qn, omega, W, have_ab, a, b, kpm_m, kernel, seed, Smat));
}

static int sd_kpm_sqw_sharded_impl(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int dtype, const void *psi0_dev,
                                  int64_t n_local, const double *q, int Qn, const double *omega, int W, int have_ab, double a,
                                  double b, int kpm_m, int kernel, uint64_t seed, double *Smat) {
  Op op; RC(op.init(ctx, m, comm));
  return kpm_sqw_core(op, dtype, psi0_dev, true, n_local, q, Qn, omega, W, have_ab, a, b, kpm_m, kernel, seed, Smat);
}
extern "C" int sd_kpm_sqw_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int dtype, const void *psi0_dev,
                                  int64_t n_local, const double *q, int Qn, const double *omega, int W, int have_ab, double a,
                                  double b, int kpm_m, int kernel, uint64_t seed, double *Smat) {
  SD_ABI_GUARD(ctx, sd_kpm_sqw_sharded_impl(ctx, m, comm, dtype, psi0_dev, n_local, q, Qn, omega, W, have_ab, a, b, kpm_m, kernel, seed, Smat));
}

static int sd_lanczos_sqw_impl(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi0, int64_t n, const double *q,
                              int Qn, const double *omega, int W, int lanc_m, double eta, int broaden, double *Smat) {
  Op op; RC(op.init(ctx, m, nullptr));
  if (n != m->N) return sd_set_err(ctx, SD_EDIM, "vector length does not match the basis dimension");
  if (dtype != SD_F64 && dtype != SD_C128) return sd_set_err(ctx, SD_EARG, "bad dtype");
  if (broaden != SD_BROADEN_LORENTZ && broaden != SD_BROADEN_GAUSS) return sd_set_err(ctx, SD_EARG, "unknown broadening");
  if (lanc_m < 1) return sd_set_err(ctx, SD_EARG, "lanc_m must be >= 1");
  const int nc = dtype == SD_C128 ? 2 : 1;
  DBuf in, psic, tmp, phi;
  RC(in.alloc(ctx, nc * n)); RC(psic.alloc(ctx, 2 * n)); RC(tmp.alloc(ctx, 2 * n)); RC(phi.alloc(ctx, 2 * n));
  RC(h2d(ctx, in.p, psi0, nc * n));
  RC(sd_k_promote(ctx, psic.p, in.p, nc, n));
  RC(plain_op(ctx, m, SD_C128, tmp.p, psic.p));                                           // src/LanczosSqw.jl:58
  // E0 = real(dot(conj(psi0c), tmp)) (sic, :59): dot conjugates its first argument again, so this is Re sum psi_i*tmp_i,
  // the product sum WITHOUT conjugation (equal to <psi|H|psi> for a real psi0).  Reduced on the device.
  RC(sd_k_dotu(ctx, psic.p, tmp.p, n, 4));
  double e0s[2]; RC(sd_read_scalars(ctx, 4, 2, e0s));
  const double E0 = e0s[0];
  const int mm = (int)std::min<int64_t>(lanc_m, n);
  std::vector<double> alpha(mm), beta(std::max(mm, 1));
  int rc = 0;
  std::vector<int> same_as;
  RC(pair_momenta(op, dtype, psic.p, n, q, Qn, same_as));      // real psi0: the Lanczos coefficients of q and 2 pi - q agree
  {   // launch-bound sizes: all momenta in one recursion (src/LanczosSqw.jl:65 threads over them)
    std::vector<int> live;
    for (int iq = 0; iq < Qn; ++iq) if (same_as[iq] < 0) live.push_back(iq);
    const int Qb = (int)live.size();
    if (ctx->q_batch && Qb >= 2 && lanczos_fused_ok(op, Qb)) {
      DBuf phib, nrm;
      RC(phib.alloc(ctx, 2 * n * Qb)); RC(nrm.alloc(ctx, 2 * (int64_t)Qb));
      for (int k = 0; k < Qb; ++k) {
        RC(sd_launch_szq(ctx, m, SD_C128, psic.p, q[live[k]], phib.p + 2 * n * k));
        RC(sd_k_nrm2sq_to(ctx, phib.p + 2 * n * k, 2 * n, nrm.p + 2 * k));
      }
      std::vector<double> hn(2 * (size_t)Qb);
      SD_HIP(ctx, hipMemcpyAsync(hn.data(), nrm.p, sizeof(double) * hn.size(), hipMemcpyDeviceToHost, ctx->stream));
      SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
      std::vector<int> kept; std::vector<double> norms;
      for (int k = 0; k < Qb; ++k) {
        const double nphi = std::sqrt(hn[2 * (size_t)k]);
        double *Srow = Smat + (size_t)live[k] * W;
        if (nphi == 0) { for (int iw = 0; iw < W; ++iw) Srow[iw] = 0.0; continue; }       // :67-70
        const int dst = (int)kept.size();
        if (dst != k) RC(d2d(ctx, phib.p + 2 * n * dst, phib.p + 2 * n * k, 2 * n));
        RC(sd_k_scale_div(ctx, phib.p + 2 * n * dst, phib.p + 2 * n * dst, 2 * n, nphi));
        kept.push_back(live[k]); norms.push_back(nphi);
      }
      const int Qk = (int)kept.size();
      if (Qk > 0) {
        std::vector<double> al, be;
        RC(lanczos_fused(op, Qk, phib.p, mm, 1, 0, 1e-12, al, be));                       // :73
        for (int k = 0; k < Qk; ++k) {
          int m_eff = 0;
          tridiag_trim(mm, 1e-12, al.data() + (size_t)k * mm, be.data() + (size_t)k * mm, alpha.data(), beta.data(), &m_eff);
          int rs = sd_spectral_from_tridiagonal(alpha.data(), beta.data(), m_eff, norms[k], E0, omega, W, eta, broaden,
                                                Smat + (size_t)kept[k] * W);
          if (rs) return sd_set_err(ctx, rs, "spectral_from_tridiagonal failed");
        }
      }
      for (int iq = 0; iq < Qn; ++iq)
        if (same_as[iq] >= 0) std::memcpy(Smat + (size_t)iq * W, Smat + (size_t)same_as[iq] * W, sizeof(double) * (size_t)W);
      return SD_OK;
    }
  }
  for (int iq = 0; iq < Qn; ++iq) {
    double *Srow = Smat + (size_t)iq * W;
    if (same_as[iq] >= 0) { std::memcpy(Srow, Smat + (size_t)same_as[iq] * W, sizeof(double) * (size_t)W); continue; }
    RC(sd_launch_szq(ctx, m, SD_C128, psic.p, q[iq], phi.p));
    const double normv = norm_dev(op, phi.p, 2 * n, &rc); RC(rc);
    if (normv == 0) { for (int iw = 0; iw < W; ++iw) Srow[iw] = 0.0; continue; }          // :67-70
    RC(sd_k_scale_div(ctx, phi.p, phi.p, 2 * n, normv));
    int m_eff = 0;
    RC(tridiag_dev(op, phi.p, lanc_m, 1e-12, alpha.data(), beta.data(), &m_eff));     // :73
    int rs = sd_spectral_from_tridiagonal(alpha.data(), beta.data(), m_eff, normv, E0, omega, W, eta, broaden, Srow);
    if (rs) return sd_set_err(ctx, rs, "spectral_from_tridiagonal failed");
  }
  return SD_OK;
}
extern "C" int sd_lanczos_sqw(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi0, int64_t n, const double *q,
                              int Qn, const double *omega, int W, int lanc_m, double eta, int broaden, double *Smat) {
  SD_ABI_GUARD(ctx, sd_lanczos_sqw_impl(ctx, m, dtype, psi0, n, q, Qn, omega, W, lanc_m, eta, broaden, Smat));
}
