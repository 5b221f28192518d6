/*
 * spindyn.h -- C ABI of libspindyn.so, the MI355X-native (gfx950 HIP) engine for
 * the matrix-free spin-1/2 Hamiltonian apply H|psi> and the recursions built on
 * it (Lanczos / Krylov / Chebyshev / KPM).
 *
 * The reference (javahedi/SpinDynamics.jl) has no FFI layer; its de-facto
 * operator seam is the Julia callable  applyH!(out, psi, model)  handed to every
 * solver (src/Lanczos.jl:27-29,87-89,196-198,255; src/TimeEvolution/Krylov.jl:
 * 136-137; src/TimeEvolution/Chebyshev.jl:61-62; src/Hamiltonian.jl:286-288) and
 * supplied by PublicAPI as Hamiltonian.apply_H! (src/PublicAPI.jl:28,62,70,79).
 * Each entry point below names the reference function it replaces; the Julia
 * `ccall` stubs a maintainer would add are in INTEGRATION.md and julia/.
 *
 * Conventions
 *  - plain C types only; every function returns an int status (SD_OK == 0).
 *  - dtype: SD_F64 (Float64) or SD_C128 (ComplexF64, interleaved re,im).
 *  - sites are 1-based as in the reference's bond tuples (i, j, J).
 *  - basis indices are 0-based at this ABI (reference index = idx0 + 1).
 *  - "host" pointers are caller-owned host memory, only touched during the call.
 *    "dev" pointers are device (HIP) pointers on the context's device.
 *  - a sd_model is immutable once set up -- creation, then at most one sd_model_set_shard[_mode] call before any other
 *    thread sees it -- and may then be shared between threads; everything a call may change (stream, scratch, the
 *    kpm switches, a caller's operator) lives in the sd_ctx, which must not be used concurrently: one context per thread.
 */
#ifndef SPINDYN_H
#define SPINDYN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes.  Mapping to the reference's Julia exceptions:
 *   SD_EARG   -> ArgumentError      (src/Basis.jl:10-16, src/SpinModel.jl:80, src/PublicAPI.jl:34,87,152)
 *   SD_EDIM   -> DimensionMismatch / AssertionError (src/Hamiltonian.jl:63-66,220,289)
 *   SD_EZERO  -> error("starting vector has zero norm") (src/Lanczos.jl:210-212)
 */
#define SD_OK 0
#define SD_EARG 1
#define SD_EDIM 2
#define SD_EZERO 3
#define SD_ENOMEM 4
#define SD_EHIP 5
#define SD_ENODEV 6
#define SD_EINTERNAL 7
#define SD_ECOMM 8   /* RCCL / exchange callback failure */

#define SD_F64 1
#define SD_C128 2

/* KPM damping kernels (src/KPM_Sqw.jl:131-145); anything else = no damping */
#define SD_KERNEL_JACKSON 0
#define SD_KERNEL_LORENTZ 1
#define SD_KERNEL_NONE 2
/* broadening for the Lanczos spectral function (src/LanczosSqw.jl:29-40) */
#define SD_BROADEN_LORENTZ 0
#define SD_BROADEN_GAUSS 1

typedef struct sd_ctx sd_ctx;
typedef struct sd_model sd_model;

/* ---- library / context ------------------------------------------------ */
const char *sd_version(void);
/* number of HIP devices visible (0 when there is no GPU; never fails) */
int sd_device_count(void);
/* Creates a context on HIP device `device` with its own stream.  Fails with
 * SD_ENODEV when no GPU is present: there is no CPU fallback in this library. */
int sd_ctx_create(int device, sd_ctx **out);
void sd_ctx_destroy(sd_ctx *ctx);
/* Use an externally owned hipStream_t (e.g. torch's current stream) for all
 * launches of this context; NULL restores the context's own stream. */
int sd_ctx_set_stream(sd_ctx *ctx, void *hip_stream);
/* Chebyshev moments (sd_kpm_moments, sd_kpm_sqw): on != 0 (default) computes two moments per apply from
 * mu_2n = 2<v_n|v_n> - mu_0, mu_2n+1 = 2Re<v_n|v_n+1> - mu_1 (v_n = T_n(H~)phi); on == 0 runs the reference's loop
 * (src/KPM_Sqw.jl:103-124), one moment <phi|v_k> per apply.  Same moments up to rounding. */
int sd_ctx_set_kpm_doubling(sd_ctx *ctx, int on);
/* sd_kpm_sqw / sd_kpm_sqw_sharded / sd_lanczos_sqw with a REAL psi0 (Float64, or ComplexF64 whose imaginary parts are all zero -- checked on
 * the device): on != 0 (default) computes the moments once per pair of momenta (q, 2 pi - q) of the list and copies the row,
 * because H is real and phi_{2pi-q} = conj(phi_q) gives the same moments (and the same Lanczos coefficients); on == 0 runs every
 * q on its own, as the reference does (src/KPM_Sqw.jl:218-252, src/LanczosSqw.jl:63-77).  The copied row differs from a recomputed one by the rounding of exp(iqr) only (<= 1e-13
 * on S).  Never used with a caller's operator (sd_ctx_set_apply_callback). */
int sd_ctx_set_kpm_pair_q(sd_ctx *ctx, int on);
/* sd_kpm_sqw / sd_lanczos_sqw on one GPU: the reference threads over the momenta (src/KPM_Sqw.jl:218, src/LanczosSqw.jl:65).  on != 0
 * (default; env SD_Q_BATCH=0 changes the default) lets the momenta's vectors share every launch of the recursion where a single
 * vector cannot fill the chip (vectors of at most 2^22 rows, tiled plans): one batched apply per step for all momenta.  Each
 * momentum sees exactly the arithmetic of a recursion of its own -- S(q, w) is bit-identical; on == 0: one momentum at a time. */
int sd_ctx_set_q_batch(sd_ctx *ctx, int on);
/* sd_lanczos_groundstate re-orthogonalises H v_j against v_1 .. v_{j-1} (src/Lanczos.jl:116-124).  on != 0 (default): in blocks
 * of 8 columns -- the coefficients of a block are its dots with w as it stands when the block begins (classical Gram-Schmidt
 * inside a block, modified between blocks), one pass over w per block instead of per column.  With v_k orthonormal to rounding
 * the coefficients differ from the reference's column-by-column chain by O(eps |coeff|): E0 to 1e-12, the vector to 1e-8.
 * on == 0: the reference's order, column by column. */
int sd_ctx_set_gs_blocked(sd_ctx *ctx, int on);
/* Operator applications (built-in H or the caller's operator) that the recursion-level entry points have queued on this
 * context since it was created: one per recursion step.  The difference across a call says how many steps it really ran
 * (e.g. how soon a queued Lanczos recursion noticed a breakdown). */
int64_t sd_ctx_apply_count(const sd_ctx *ctx);
/* A context keeps between calls: the device staging buffers of the host-pointer operator calls (sd_apply,
 * sd_apply_rescaled: two vectors), its reduction scratch, and the work vectors of the recursion-level calls (a pool of at
 * most SD_POOL_MAX_GB = 96 GB by default: a hipMalloc/hipFree pair costs 0.3-0.6 ms whatever the size, as long as
 * 15-30 recursion steps of a small system -- the role of the reference's `workspace` argument,
 * src/TimeEvolution/Chebyshev.jl:61-66).  This call frees all
 * of them; they are re-created on demand. */
int sd_ctx_release_scratch(sd_ctx *ctx);
int sd_ctx_synchronize(sd_ctx *ctx);
/* last error text of this context ("" if none); valid until the next call */
const char *sd_last_error(const sd_ctx *ctx);
/* text for a status code */
const char *sd_status_string(int status);

/* ---- model (replaces SpinModel.Model, src/SpinModel.jl:6-38) ------------ */
/* nup = -1 selects the full 2^L basis (`nup === nothing`).  states/idxmap are
 * never passed: the basis order of build_sector_basis (src/Basis.jl:37-53) is
 * reproduced from closed-form combinatorial ranking.  ctx may be NULL: the
 * model is then host-only (basis queries work, device applies do not). */
int sd_model_create(sd_ctx *ctx, int L, int nup,
                    int n_hop, const int *hop_i, const int *hop_j, const double *hop_J,
                    int n_zz, const int *zz_i, const int *zz_j, const double *zz_J,
                    const double *field /* L entries or NULL */, sd_model **out);
/* XXZChain(L; Jxy, Jz, hz, nup, boundary) -- src/SpinModel.jl:63-90.
 * boundary: 0 = :open, 1 = :periodic; anything else -> SD_EARG. */
int sd_xxz_chain(sd_ctx *ctx, int L, double Jxy, double Jz, double hz, int nup, int boundary,
                 sd_model **out);
void sd_model_destroy(sd_model *m);
int64_t sd_model_dim(const sd_model *m);      /* length(model.states) */
int sd_model_L(const sd_model *m);
int sd_model_nup(const sd_model *m);          /* -1 for the full basis */
/* which device path the apply takes: 0 generic (per-row rank/unrank), 1 tiled (prefix-run tiles, LDS staged suffix
 * hops), 2 full-basis tiles of 2^10 rows */
int sd_model_path(const sd_model *m);
/* model.states[start .. start+count)  (host computation, 0-based start) */
int sd_model_states(const sd_model *m, int64_t start, int64_t count, uint64_t *states_out);
/* get(model.idxmap, state, 0) - 1 : 0-based index of each state, -1 if absent */
int sd_model_rank(const sd_model *m, const uint64_t *states, int64_t n, int64_t *idx_out);

/* ---- operator level (replaces applyH!(out, psi, model)) ----------------- */
/* apply_H!   src/Hamiltonian.jl:211-273.  out is overwritten, must not alias
 * psi, n must equal sd_model_dim (SD_EDIM otherwise). */
int sd_apply(sd_ctx *ctx, const sd_model *m, int dtype, void *out_host, const void *psi_host, int64_t n);
int sd_apply_dev(sd_ctx *ctx, const sd_model *m, int dtype, void *out_dev, const void *psi_dev, int64_t n);
/* apply_rescaled_H!   src/Hamiltonian.jl:286-301:  out = (H psi - b psi) / a, fused in one pass */
int sd_apply_rescaled(sd_ctx *ctx, const sd_model *m, int dtype, void *out_host, const void *psi_host,
                      int64_t n, double a, double b);
int sd_apply_rescaled_dev(sd_ctx *ctx, const sd_model *m, int dtype, void *out_dev, const void *psi_dev,
                          int64_t n, double a, double b);
/* The operator as a callable at the recursion level.  Every reference solver takes `applyH!` as an argument
 * (src/Lanczos.jl:27-29, src/TimeEvolution/Chebyshev.jl:61-64, src/KPM_Sqw.jl:95-98, ...).  With a callback installed in a
 * context, every recursion entry point called on THAT context (lanczos_*, energy_bounds, krylov / chebyshev evolve, kpm_*,
 * *_sqw, and their _dev / _sharded forms) calls fn for out <- H psi instead of the built-in kernel and applies its own fused
 * step (rescale, recurrence, dot products) in a second, elementwise pass.  fn gets DEVICE pointers of n_local elements of
 * `dtype` and the HIP stream the call must be ordered on (enqueue there, or finish before returning); out never aliases psi;
 * a nonzero return aborts the call with SD_ECOMM.  The operator-level entries (sd_apply*, sd_apply_sharded*) always run the
 * built-in operator, so fn may call them -- on a sharded model sd_apply_sharded, which includes the halo exchange.
 * fn == NULL restores the built-in operator.  The callback belongs to the context, not to the model: models stay immutable
 * and shareable, and another thread's recursions (on its own context) are unaffected. */
typedef int (*sd_apply_fn)(void *user, int dtype, void *out_dev, const void *psi_dev, int64_t n_local, void *hip_stream);
int sd_ctx_set_apply_callback(sd_ctx *ctx, sd_apply_fn fn, void *user);
/* Sz_q_vector   src/Hamiltonian.jl:307-337.  phi_out is always ComplexF64. */
int sd_szq(sd_ctx *ctx, const sd_model *m, int dtype_in, const void *psi0_host, int64_t n, double q,
           void *phi_out_host);
int sd_szq_dev(sd_ctx *ctx, const sd_model *m, int dtype_in, const void *psi0_dev, int64_t n, double q,
               void *phi_out_dev);

/* Fused Chebyshev term on device vectors (all ComplexF64, length n):
 *   phi_next = 2*(H phi_curr - b phi_curr)/a - phi_prev ;  psi_t += c * phi_next
 * = one iteration of src/TimeEvolution/Chebyshev.jl:110-121 in a single pass. */
int sd_cheb_step_dev(sd_ctx *ctx, const sd_model *m, void *phi_next_dev, const void *phi_curr_dev,
                     const void *phi_prev_dev, void *psi_t_dev, int64_t n, double a, double b,
                     double c_re, double c_im);

/* Timed loop for benchmarks: `reps` applies out<-H psi (ping-pong between the
 * two buffers) bracketed by hipEvents on the context's stream; returns the
 * average milliseconds per apply.  Both buffers are device pointers. */
int sd_bench_apply_dev(sd_ctx *ctx, const sd_model *m, int dtype, void *buf_a_dev, void *buf_b_dev,
                       int64_t n, int reps, float *ms_per_apply);

/* ---- recursion level ---------------------------------------------------- */
/* All vectors are host arrays; the recursion runs entirely on the device
 * (apply kernel + BLAS-1 kernels), only scalars cross per iteration.
 * psi0 arguments marked "or NULL" are drawn from the library's counter-based
 * normal generator keyed by `seed` when NULL (the reference uses Julia's
 * randn, whose stream cannot be reproduced outside Julia). */

/* lanczos_extremal   src/Lanczos.jl:27-84   (ComplexF64 start vector) */
int sd_lanczos_extremal(sd_ctx *ctx, const sd_model *m, int lanc_m, double tol,
                        const void *psi0_c128_host /* or NULL */, uint64_t seed, int negate,
                        double *emin, double *emax);
/* estimate_energy_bounds   src/Lanczos.jl:255-271 */
int sd_energy_bounds(sd_ctx *ctx, const sd_model *m, int lanc_m,
                     const void *psi0_a_c128_host /* or NULL */, const void *psi0_b_c128_host /* or NULL */,
                     uint64_t seed, double *emin, double *emax);
/* lanczos_groundstate   src/Lanczos.jl:87-181   (Float64, full re-orthogonalisation) */
int sd_lanczos_groundstate(sd_ctx *ctx, const sd_model *m, int lanc_m, double tol, double orth_tol,
                           const double *psi0_host /* or NULL */, uint64_t seed,
                           double *E0, double *psi_gs_host /* n doubles */, int *m_actual);
/* lanczos_tridiag   src/Lanczos.jl:196-246.  alpha_out[min(lanc_m,n)], beta_out[min(lanc_m,n)-1] */
int sd_lanczos_tridiag(sd_ctx *ctx, const sd_model *m, const void *v_c128_host, int64_t n, int lanc_m,
                       double tol, double *alpha_out, double *beta_out, int *m_eff, double *norm_v);
/* krylov_time_evolve   src/TimeEvolution/Krylov.jl:136-192.  psit_out is ComplexF64. */
int sd_krylov_evolve(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi0_host, int64_t n,
                     double dt, int kry_m, void *psit_out_c128_host);
/* the same on device vectors (psi0 of `dtype`, psit ComplexF64; psit may be a ComplexF64 psi0).  Returns after the stream
 * has finished. */
int sd_krylov_evolve_dev(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi0_dev, int64_t n,
                         double dt, int kry_m, void *psit_out_c128_dev);
/* chebyshev_time_evolve   src/TimeEvolution/Chebyshev.jl:61-124  (psi0 ComplexF64) */
int sd_chebyshev_evolve(sd_ctx *ctx, const sd_model *m, const void *psi0_c128_host, int64_t n, double dt,
                        int cheb_n, double Emin, double Emax, void *psit_out_c128_host);
/* the same on device vectors (ComplexF64, n elements): psi stays on the GPU between the steps of a time evolution.
 * psit_dev may be psi0_dev (in place).  Returns after the stream has finished. */
int sd_chebyshev_evolve_dev(sd_ctx *ctx, const sd_model *m, const void *psi0_c128_dev, int64_t n, double dt,
                            int cheb_n, double Emin, double Emax, void *psit_out_c128_dev);
/* compute_chebyshev_moments   src/KPM_Sqw.jl:95-128 */
int sd_kpm_moments(sd_ctx *ctx, const sd_model *m, const void *phi_c128_host, int64_t n, int M,
                   double a, double b, double *mu_out);
/* get_kernel   src/KPM_Sqw.jl:131-145 (host) */
int sd_kpm_kernel(int M, int kernel, double *g_out);
/* _rescaling_from_bounds   src/KPM_Sqw.jl:13-17 (host) */
int sd_kpm_rescaling_from_bounds(double Emin, double Emax, double *a, double *b);
/* reconstruction part of kpm_sw   src/KPM_Sqw.jl:55-90 (host), moments already damped */
int sd_kpm_reconstruct(const double *mu_damped, int kpm_m, const double *omega, int W, double a, double b,
                       double E0, double *S_out);
/* kpm_sqw   src/KPM_Sqw.jl:191-256.  Smat_out is Qn x W row-major.  When
 * have_ab == 0 the rescaling is estimated as the reference does (two Lanczos
 * runs, lanc_m = 80) with generated start vectors keyed by `seed`. */
int sd_kpm_sqw(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi0_host, int64_t n,
               const double *q, int Qn, const double *omega, int W, int have_ab, double a, double b,
               int kpm_m, int kernel, uint64_t seed, double *Smat_out);
/* spectral_from_tridiagonal   src/LanczosSqw.jl:18-43 (host) */
int sd_spectral_from_tridiagonal(const double *alpha, const double *beta, int m, double norm_phi, double E0,
                                 const double *omega, int W, double eta, int broaden, double *S_out);
/* lanczos_sqw   src/LanczosSqw.jl:49-80 */
int sd_lanczos_sqw(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi0_host, int64_t n,
                   const double *q, int Qn, const double *omega, int W, int lanc_m, double eta, int broaden,
                   double *Smat_out);

/* ---- observables and initial states (reference src/Observables.jl, src/InitialStates.jl) ---- */
/* magnetization_per_site   src/Observables.jl:14-36 : mags_out[L] = <S^z_i> */
int sd_magnetization(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_host, int64_t n, double *mags_out);
int sd_magnetization_dev(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_dev, int64_t n, double *mags_out);
/* connected_correlations   src/Observables.jl:44-94 : C_out[L], C_r = (1/L) sum_i (<S_i S_j> - <S_i><S_j>), j = mod1(i+r,L) */
int sd_connected_correlations(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_host, int64_t n, double *C_out);
int sd_connected_correlations_dev(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_dev, int64_t n, double *C_out);
/* structure_factor_Sq   src/Observables.jl:100-109 : q_out[k] = 2 pi k / L, S_out[k] = real(fft(C_r))[k] */
int sd_structure_factor(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_host, int64_t n, double *q_out, double *S_out);
int sd_structure_factor_dev(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_dev, int64_t n, double *q_out, double *S_out);
/* create_spin_operator(site, op)(psi, model)   src/Hamiltonian.jl:49-136.  site is 1-based.  S^z is diagonal and works
 * in any basis; S^+, S^-, S^x, S^y change the magnetisation and are rejected in a fixed-nup sector (SD_EARG), as the
 * reference does.  out has psi's element type; S^y needs a ComplexF64 psi (SD_EARG otherwise: the reference's
 * accumulation of +-0.5im*psi into a Float64 result raises InexactError). */
#define SD_SPIN_Z 0
#define SD_SPIN_PLUS 1
#define SD_SPIN_MINUS 2
#define SD_SPIN_X 3
#define SD_SPIN_Y 4
int sd_spin_operator(sd_ctx *ctx, const sd_model *m, int dtype, int site, int op, const void *psi_host, int64_t n,
                     void *out_host);
/* InitialStates (src/InitialStates.jl:9-130): 0-based basis index of the one-hot state; SD_EARG when the
 * configuration is not in the basis.  flips: 1-based sites for SD_STATE_POLARIZED_FLIPS. */
#define SD_STATE_DOMAIN_WALL 0
#define SD_STATE_NEEL 1
#define SD_STATE_POLARIZED_UP 2
#define SD_STATE_POLARIZED_DOWN 3
#define SD_STATE_POLARIZED_FLIPS 4
int sd_initial_state_index(const sd_model *m, int kind, const int *flips, int nflips, int64_t *idx0_out);

/* ---- host utilities ------------------------------------------------------ */
/* eigen(SymTridiagonal(d, e)): ascending eigenvalues w[n]; z (n*n column-major) may be NULL */
int sd_symtridiag_eig(int n, const double *d, const double *e, double *w, double *z);
/* Chebyshev expansion coefficients c_k (src/TimeEvolution/Chebyshev.jl:74-79); c_out has 2*cheb_n doubles */
int sd_chebyshev_coeffs(int cheb_n, double a, double b, double dt, double *c_out);
/* counter-based N(0,1) generator used for synthetic vectors: element k of the
 * stream (seed) -- identical on host and device, independent of sharding. */
int sd_fill_randn_dev(sd_ctx *ctx, void *x_dev, int64_t n_doubles, uint64_t seed, uint64_t first_index);
int sd_fill_randn_host(double *x, int64_t n_doubles, uint64_t seed, uint64_t first_index);
/* dot(x, y) = sum conj(x_k) y_k and sum |x_k|^2 over n elements of device vectors (LinearAlgebra.dot / norm^2 as the
 * recursions use them, e.g. src/Lanczos.jl:40,55,59), reduced in a fixed order (same inputs -> same bits, any n up to
 * 2^62).  out_re_im[2] / out[1] are host doubles; the call synchronises the stream. */
int sd_dot_dev(sd_ctx *ctx, int dtype, const void *x_dev, const void *y_dev, int64_t n, double *out_re_im);
int sd_nrm2sq_dev(sd_ctx *ctx, int dtype, const void *x_dev, int64_t n, double *out);

/* ---- index-range sharding (multi-GPU; one process per GPU) -------------- */
/* A shard owns the contiguous basis-index range [row_lo, row_hi) (aligned to
 * tile boundaries).  Vectors of a sharded model are device arrays of
 * n_local + n_halo elements: the first n_local are the owned rows, the tail is
 * the halo filled by the exchange (RCCL send/recv driven by the host layer)
 * before each apply.  The plan lists contiguous slabs to send / receive. */
typedef struct sd_shard_info {
  int rank, nranks;
  int64_t row_lo, row_hi;   /* owned global rows */
  int64_t n_local, n_halo;  /* elements */
  int64_t n_recv_slabs, n_send_slabs;
  int mode;                 /* 0 index ranges (send slabs are slices of psi), 1 popcount cells (send slabs are slices
                               of the packed send buffer filled by sd_shard_pack_dev) */
  int64_t n_send;           /* elements of the packed send buffer (mode 1), else 0 */
  int64_t n_local_tiles;
  int64_t n_pack;           /* entries of the pack list (mode 1) */
  int64_t n_interior_tiles; /* tiles whose hop partners are all owned (sd_apply_sharded_dev part 1) */
  int64_t n_interior_rows;  /* rows of those tiles (the rest of n_local waits for the halo) */
  int64_t packed;           /* mode 1: 1 = the send slabs index the packed send buffer (sd_shard_pack_dev, n_send elements); 0 = they index the
                               vector itself -- the tiles a peer needs form long contiguous runs, one slab per run, no pack, no send buffer */
} sd_shard_info;
typedef struct sd_slab {
  int peer;                 /* rank on the other side */
  int64_t local_offset;     /* element offset in THIS rank's vector (send: within owned rows; recv: >= n_local) */
  int64_t count;            /* elements */
  int64_t global_row;       /* global basis index of the slab's first element */
} sd_slab;
/* Re-plans the model as shard `rank` of `nranks` (nranks == 1 restores the
 * unsharded plan).  Must be called before any apply on that model.  On failure
 * (SD_ENOMEM, SD_EHIP, ...) the model is left WITHOUT a plan: every later call
 * that needs one returns SD_EARG; destroy the model (or call set_shard again). */
int sd_model_set_shard(sd_model *m, int rank, int nranks);
/* One KPM moment step on a shard (src/KPM_Sqw.jl:106-117), ComplexF64 device vectors of n_local elements:
 * first != 0: v_next = (H v_curr - b v_curr)/a; else v_next = 2 (H v_curr - b v_curr)/a - v_prev.
 * sums_out[0..1] = this rank's { Re<phi|v_next>, |v_next|^2 }: all-reduce over the ranks for mu_n and the norm. */
int sd_kpm_step_sharded_dev(sd_ctx *ctx, const sd_model *m, void *v_next_dev, const void *v_curr_dev, const void *halo_dev,
                            const void *v_prev_dev, const void *phi_dev, int64_t n_local, double a, double b, int first,
                            double *sums_out);
/* mode: -1 auto (env SD_SHARD_MODE=range|class, default class when the model allows it), 0 index ranges, 1 popcount cells */
int sd_model_set_shard_mode(sd_model *m, int rank, int nranks, int mode);
/* local tiles in natural order: offset in the local vector, GLOBAL basis index of the first row, rows (arrays of
 * sd_shard_info.n_local_tiles entries; any may be NULL).  local row = local_base + i  <->  global row = global_base + i */
int sd_model_local_tiles(const sd_model *m, int64_t *local_base, int64_t *global_base, int32_t *len);
/* mode 1: the pack list -- tile k of it is psi[src[k] .. src[k]+len[k]) -> sendbuf[dst[k] ..) (n_pack entries each) */
int sd_model_shard_pack_list(const sd_model *m, int64_t *src, int64_t *dst, int32_t *len);
/* mode 1: gather the tiles the peers need into the contiguous send buffer (n_send elements) */
int sd_shard_pack_dev(sd_ctx *ctx, const sd_model *m, int dtype, const void *psi_dev, void *sendbuf_dev);
/* counter-based N(0,1) keyed by the GLOBAL element index into this shard's local vector (any sharding gives the same state) */
int sd_fill_randn_local_dev(sd_ctx *ctx, const sd_model *m, int dtype, void *x_dev, uint64_t seed);
/* Sharded apply with the imported partner tiles in a SEPARATE halo buffer (n_halo elements, filled by the exchange):
 * vectors then hold exactly n_local elements and one halo buffer serves every vector of a recursion.
 * epilogue: 0 out = H psi; 1 out = (H psi - b psi)/a; 2 fused Chebyshev term (ComplexF64; phi_prev, psi_t as in
 * sd_cheb_step_dev); 3 recurrence only, out = 2 (H psi - b psi)/a - phi_prev (psi_t untouched).  part: 0 all tiles; 1 only the interior tiles (every hop partner owned: reads no halo, so it
 * can run while the exchange is in flight); 2 only the boundary tiles.  All pointers are device pointers. */
int sd_apply_sharded_dev(sd_ctx *ctx, const sd_model *m, int dtype, void *out_dev, const void *psi_dev,
                         const void *halo_dev, int64_t n_local, int epilogue, double a, double b,
                         double c_re, double c_im, const void *phi_prev_dev, void *psi_t_dev, int part);
/* The pair form of the Chebyshev term (src/TimeEvolution/Chebyshev.jl:110-121): sd_apply_sharded_dev with epilogue 3
 * computes phi_k = 2 H~ phi_{k-1} - phi_{k-2} without touching psi_t; this call then computes
 * out = phi_{k+1} = 2 H~ psi - phi_prev (psi = phi_k) and psi_t += c0*phi_k; psi_t += c*phi_{k+1}, in that order --
 * the same bits as one accumulation per term, with psi_t read and written once per two terms. */
int sd_apply_sharded_cheb2_dev(sd_ctx *ctx, const sd_model *m, void *out, const void *psi, const void *halo, int64_t n_local,
                               double a, double b, double c0_re, double c0_im, double c_re, double c_im,
                               const void *phi_prev, void *psi_t, int part);
int sd_model_shard_info(const sd_model *m, sd_shard_info *out);
int sd_model_shard_slabs(const sd_model *m, sd_slab *recv_out, sd_slab *send_out);

/* ---- sharded recursions (one process per GPU) ------------------------------------------------------------------
 * The reference has no distributed layer (its only parallel construct above apply_H! is the thread loop over the momenta,
 * src/KPM_Sqw.jl:218).  These are the recursion-level entry points of above for a model re-planned with
 * sd_model_set_shard[_mode]: every vector argument is this rank's OWNED part (n_local elements, device pointer), every
 * rank makes the same call, scalars come back identical on all ranks.  A sd_comm says how halos travel and scalars are
 * summed; with the RCCL communicator a recursion step (pack, grouped send/recv beside the interior tiles, boundary tiles,
 * BLAS-1 passes, ncclAllReduce of the device scalars) is queued without touching the host.  Halo and send buffers come
 * from the context's pool: per rank a recursion holds its vectors (n_local elements each) + n_halo + n_send elements. */
typedef struct sd_comm sd_comm;
typedef struct sd_comm_callbacks {
  void *user;
  /* Post the halo exchange: for every send slab of this rank (sd_model_shard_slabs) send src_dev[local_offset ..
   * local_offset+count) to slab.peer, for every recv slab receive into halo_dev[local_offset - n_local ..).  src_dev is the
   * vector itself (mode 0) or the packed send buffer (mode 1); elements are dtype-sized; both are device pointers whose
   * producers are queued on the context's stream.  May return before the bytes have moved. */
  int (*exchange_start)(void *user, int dtype, const void *src_dev, void *halo_dev);
  /* Return once work queued on the context's stream after this call is ordered behind the completed exchange. */
  int (*exchange_wait)(void *user);
  /* vals[0..count) <- sum over all ranks (host memory; count <= 16). */
  int (*allreduce_sum)(void *user, double *vals, int count);
} sd_comm_callbacks;
/* nonzero return of a callback aborts the call with SD_ECOMM */
int sd_comm_from_callbacks(const sd_comm_callbacks *cb, int rank, int nranks, sd_comm **out);
/* RCCL over xGMI.  Rank 0 obtains a 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means (the
 * Julia front-end: Distributed / MPI; the Python mirror: torch.distributed broadcast); every rank then creates its
 * communicator on its context's device.  Collective: all nranks must call. */
int sd_comm_rccl_unique_id(void *id128);
int sd_comm_rccl_create(sd_ctx *ctx, int rank, int nranks, const void *id128, sd_comm **out);
void sd_comm_destroy(sd_comm *comm);
/* Routed halo exchange (RCCL communicator).  By default an exchange is one grouped ncclSend / ncclRecv of the model's slab lists
 * (sd_model_shard_slabs): every (owner, receiver) pair's bytes ride the one xGMI link that joins the pair, and the busiest pair
 * carries two to three times the mean.  This installs an explicit list instead: operations in ascending `batch` order, one RCCL
 * group per batch on the exchange stream, `buf` naming what `offset` (in elements) counts from -- 0 the vector handed to the
 * exchange (sends only), 1 the halo buffer (receives only), 2 this rank's relay buffer of `relay_elems` elements (a piece
 * received in batch b is forwarded in batch b + 1).  Every rank must install lists that pair up (the Python mirror builds them
 * from one routing plan: dist.relay_routes / dist.relay_ops).  n_ops == 0 restores the default. */
typedef struct sd_xop {
  int batch;        /* ascending through the list */
  int peer;
  int kind;         /* 0 send, 1 receive */
  int buf;          /* 0 vector, 1 halo, 2 relay buffer */
  int64_t offset;   /* elements from the start of that buffer */
  int64_t count;    /* elements */
} sd_xop;
int sd_comm_set_exchange_ops(sd_comm *comm, const sd_xop *ops, int64_t n_ops, int64_t relay_elems);
/* Diagnostic (RCCL communicator): ncclAllReduce of two device doubles and a grouped ncclSend/ncclRecv round the ring of ranks
 * (rank -> rank+1; to itself with nranks == 1), bytes checked.  Collective: with more than one rank all of them must call it. */
int sd_comm_selftest(sd_ctx *ctx, sd_comm *comm);

/* out = H psi on the owned rows, halo exchange included (overlapped with the interior tiles when overlap != 0). */
int sd_apply_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int dtype, void *out_dev, const void *psi_dev,
                     int64_t n_local, int overlap);
/* lanczos_extremal / estimate_energy_bounds (src/Lanczos.jl:27-84, 255-271); psi0_dev NULL: counter-based N(0,1) start
 * vector keyed by the global row index (the same state for every sharding). */
int sd_lanczos_extremal_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int lanc_m, double tol, const void *psi0_dev,
                                uint64_t seed, int negate, double *emin, double *emax);
int sd_energy_bounds_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int lanc_m, uint64_t seed, double *Emin,
                             double *Emax);
/* chebyshev_time_evolve (src/TimeEvolution/Chebyshev.jl:61-124) on ComplexF64 shards; psit_dev may alias psi0_dev. */
int sd_chebyshev_evolve_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, const void *psi0_dev, int64_t n_local,
                                double dt, int cheb_n, double Emin, double Emax, void *psit_dev);
/* krylov_time_evolve (src/TimeEvolution/Krylov.jl:136-192) on shards (psit ComplexF64). */
int sd_krylov_evolve_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int dtype, const void *psi0_dev,
                             int64_t n_local, double dt, int kry_m, void *psit_dev);
/* compute_chebyshev_moments (src/KPM_Sqw.jl:95-128): phi_dev normalised over all ranks; mu (host, M entries) on every rank. */
int sd_kpm_moments_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, const void *phi_dev, int64_t n_local, int M,
                           double a, double b, double *mu);
/* kpm_sqw (src/KPM_Sqw.jl:191-256) on a sharded psi0 (BASELINE config 5); Smat (host, Qn x W row-major) on every rank. */
int sd_kpm_sqw_sharded(sd_ctx *ctx, const sd_model *m, sd_comm *comm, int dtype, const void *psi0_dev, int64_t n_local,
                       const double *q, int Qn, const double *omega, int W, int have_ab, double a, double b, int kpm_m,
                       int kernel, uint64_t seed, double *Smat);
/* <x|y> (re, im) and |x|^2 over all ranks */
int sd_dot_sharded(sd_ctx *ctx, sd_comm *comm, int dtype, const void *x_dev, const void *y_dev, int64_t n_local,
                   double *out2);

#ifdef __cplusplus
}
#endif
#endif /* SPINDYN_H */
