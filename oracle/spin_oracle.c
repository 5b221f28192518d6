/*
 * oracle/spin_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * A plain-C CPU restatement of the hot path of javahedi/SpinDynamics.jl (the
 * matrix-free H|psi> apply and the Lanczos / Krylov / Chebyshev / KPM
 * recursions that call it).  It keeps the reference's *algorithm and data
 * structures*: an explicit `states[]` array built in lexicographic-combination
 * order plus a hash map state -> index (the reference's Dict), a row-owner
 * gather apply, an un-fused rescale pass, and un-fused BLAS-1 style vector
 * passes.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (libspindyn.so) never does.
 *
 * Parity pin: the reference is Julia and no Julia runtime exists in the build
 * container or on the GPU box, so the reference itself cannot be executed.
 * This restatement is pinned by the reference's own known-answer tests
 * (test/test_PublicAPI.jl, test/test_Lanczos.jl, test/test_KPM.jl,
 * test/test_Hamiltonian.jl, test/test_Basis.jl -- see tests/test_oracle_*.py)
 * and by an independent numpy oracle (oracle/dense.py: Kronecker-product H
 * projected on the itertools.combinations order).  The ORDER of states inside
 * a sector is pinned only by Combinatorics.combinations' documented
 * lexicographic order (no reference test indexes states[k]).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference repository root).
 *
 * Arithmetic notes: compiled with -ffp-contract=off so that a*b+c is never
 * fused (Julia does not contract).  Complex numbers are handled as explicit
 * (re, im) pairs, following Julia's Complex formulas.  A Complex{Float64}
 * whose imaginary part is exactly zero multiplied into a complex vector
 * element is evaluated component-wise (identical except for the sign of
 * zero).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SO_OK 0
#define SO_EARG 1     /* ArgumentError in the reference */
#define SO_EDIM 2     /* DimensionMismatch / AssertionError on lengths */
#define SO_EZERO 3    /* error("starting vector has zero norm") */
#define SO_ENOMEM 4

typedef struct so_model {
  int L;
  int nup;            /* -1 == `nothing` (full basis) */
  int64_t N;          /* length(states) */
  uint64_t *states;   /* sector: lex-combination order; full: NULL (state = idx-1) */
  uint64_t *hkeys;    /* open-addressing hash map == the reference's idxmap */
  int64_t *hvals;     /* 1-based index, 0 = empty slot */
  uint64_t hmask;
  int n_hop;
  int *hop_i, *hop_j; /* 1-based sites */
  double *hop_J;
  int n_zz;
  int *zz_i, *zz_j;
  double *zz_J;
  double *field;      /* length L */
} so_model;

/* ------------------------------------------------------------------ */
/* helpers                                                             */
/* ------------------------------------------------------------------ */

/* src/Hamiltonian.jl:19-21 bit_at */
static inline uint64_t bit_at(uint64_t state, int i) { return (state >> i) & 1u; }
/* src/Hamiltonian.jl:23-25 sz_value */
static inline double sz_value(uint64_t bit) { return bit == 1 ? 0.5 : -0.5; }
/* src/Hamiltonian.jl:27-29 flip_bits */
static inline uint64_t flip_bits(uint64_t s, int i, int j) {
  return s ^ ((uint64_t)1 << i) ^ ((uint64_t)1 << j);
}

uint64_t so_bit_at(uint64_t s, int i) { return bit_at(s, i); }
double so_sz_value(uint64_t bit) { return sz_value(bit); }
uint64_t so_flip_bits(uint64_t s, int i, int j) { return flip_bits(s, i, j); }

int64_t so_binomial(int n, int k) {
  if (k < 0 || k > n) return 0;
  if (k > n - k) k = n - k;
  __int128 r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return (int64_t)r;
}

static inline uint64_t hash64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33; return x;
}

/* the reference's get(model.idxmap, state, 0) -- src/Hamiltonian.jl:260 */
static inline int64_t idx_lookup(const so_model *m, uint64_t s) {
  uint64_t h = hash64(s) & m->hmask;
  for (;;) {
    int64_t v = m->hvals[h];
    if (v == 0) return 0;
    if (m->hkeys[h] == s) return v;
    h = (h + 1) & m->hmask;
  }
}

int64_t so_lookup(const so_model *m, uint64_t s) {
  if (m->nup < 0) return (s < (uint64_t)m->N) ? (int64_t)s + 1 : 0;
  return idx_lookup(m, s);
}

/* ------------------------------------------------------------------ */
/* basis + model                                                       */
/* ------------------------------------------------------------------ */

/* src/Basis.jl:9-20 _validate_basis_args */
static int validate_basis_args(int L, int nup) {
  if (L < 1) return SO_EARG;
  if (L > 63) return SO_EARG;
  if (nup >= 0 && nup > L) return SO_EARG;
  if (nup < -1) return SO_EARG;
  return SO_OK;
}

/* src/Basis.jl:37-53 build_sector_basis: lexicographic combinations(1:L,nup),
 * bit (i-1) set for each chosen site i; then idxmap[s] = position (1-based). */
static int build_sector_basis(so_model *m) {
  int L = m->L, t = m->nup;
  int64_t N = so_binomial(L, t);
  m->N = N;
  m->states = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(N > 0 ? N : 1));
  if (!m->states) return SO_ENOMEM;
  int c[64];
  for (int k = 0; k < t; ++k) c[k] = k + 1;
  int64_t n = 0;
  for (;;) {
    uint64_t s = 0;
    for (int k = 0; k < t; ++k) s |= (uint64_t)1 << (c[k] - 1);
    m->states[n++] = s;
    int k = t - 1;
    while (k >= 0 && c[k] == L - t + k + 1) --k;
    if (k < 0) break;
    ++c[k];
    for (int q = k + 1; q < t; ++q) c[q] = c[q - 1] + 1;
  }
  if (n != N) return SO_EDIM;
  uint64_t cap = 16;
  while (cap < (uint64_t)N * 2) cap <<= 1;
  m->hmask = cap - 1;
  m->hkeys = (uint64_t *)calloc(cap, sizeof(uint64_t));
  m->hvals = (int64_t *)calloc(cap, sizeof(int64_t));
  if (!m->hkeys || !m->hvals) return SO_ENOMEM;
  for (int64_t i = 0; i < N; ++i) {
    uint64_t s = m->states[i], h = hash64(s) & m->hmask;
    while (m->hvals[h] != 0) h = (h + 1) & m->hmask;
    m->hkeys[h] = s;
    m->hvals[h] = i + 1;
  }
  return SO_OK;
}

void so_model_free(so_model *m) {
  if (!m) return;
  free(m->states); free(m->hkeys); free(m->hvals);
  free(m->hop_i); free(m->hop_j); free(m->hop_J);
  free(m->zz_i); free(m->zz_j); free(m->zz_J); free(m->field);
  free(m);
}

/* src/SpinModel.jl:23-38 build_model (+ src/Basis.jl:23-34 build_full_basis:
 * full mode is the identity map state = idx-1, never read through the Dict). */
int so_model_create(int L, int nup, int n_hop, const int *hop_i, const int *hop_j,
                    const double *hop_J, int n_zz, const int *zz_i, const int *zz_j,
                    const double *zz_J, const double *field, so_model **out) {
  *out = NULL;
  int rc = validate_basis_args(L, nup);
  if (rc) return rc;
  so_model *m = (so_model *)calloc(1, sizeof(so_model));
  if (!m) return SO_ENOMEM;
  m->L = L; m->nup = nup;
  if (nup < 0) {
    m->N = (int64_t)1 << L;
  } else {
    rc = build_sector_basis(m);
    if (rc) { so_model_free(m); return rc; }
  }
  m->n_hop = n_hop; m->n_zz = n_zz;
  m->hop_i = (int *)malloc(sizeof(int) * (n_hop + 1));
  m->hop_j = (int *)malloc(sizeof(int) * (n_hop + 1));
  m->hop_J = (double *)malloc(sizeof(double) * (n_hop + 1));
  m->zz_i = (int *)malloc(sizeof(int) * (n_zz + 1));
  m->zz_j = (int *)malloc(sizeof(int) * (n_zz + 1));
  m->zz_J = (double *)malloc(sizeof(double) * (n_zz + 1));
  m->field = (double *)malloc(sizeof(double) * L);
  for (int k = 0; k < n_hop; ++k) { m->hop_i[k] = hop_i[k]; m->hop_j[k] = hop_j[k]; m->hop_J[k] = hop_J[k]; }
  for (int k = 0; k < n_zz; ++k) { m->zz_i[k] = zz_i[k]; m->zz_j[k] = zz_j[k]; m->zz_J[k] = zz_J[k]; }
  for (int k = 0; k < L; ++k) m->field[k] = field ? field[k] : 0.0;
  *out = m;
  return SO_OK;
}

/* src/SpinModel.jl:63-90 XXZChain: hopping = Jxy/2, zz = Jz, periodic adds
 * (L,1) only when L > 2, field = fill(hz, L).  boundary: 0 open, 1 periodic. */
int so_xxz_chain(int L, double Jxy, double Jz, double hz, int nup, int boundary,
                 so_model **out) {
  *out = NULL;
  if (boundary != 0 && boundary != 1) return SO_EARG;
  if (L < 1 || L > 63) return SO_EARG;
  int hi[64], hj[64], zi[64], zj[64];
  double hJ[64], zJ[64], f[64];
  int nb = 0;
  for (int i = 1; i <= L - 1; ++i) {
    hi[nb] = i; hj[nb] = i + 1; hJ[nb] = Jxy / 2; zi[nb] = i; zj[nb] = i + 1; zJ[nb] = Jz; ++nb;
  }
  if (boundary == 1 && L > 2) {
    hi[nb] = L; hj[nb] = 1; hJ[nb] = Jxy / 2; zi[nb] = L; zj[nb] = 1; zJ[nb] = Jz; ++nb;
  }
  for (int i = 0; i < L; ++i) f[i] = hz;
  return so_model_create(L, nup, nb, hi, hj, hJ, nb, zi, zj, zJ, f, out);
}

int64_t so_dim(const so_model *m) { return m->N; }
int so_L(const so_model *m) { return m->L; }
const uint64_t *so_states_ptr(const so_model *m) { return m->states; }

void so_states(const so_model *m, uint64_t *out) {
  if (m->nup < 0) for (int64_t i = 0; i < m->N; ++i) out[i] = (uint64_t)i;
  else memcpy(out, m->states, sizeof(uint64_t) * (size_t)m->N);
}

/* ------------------------------------------------------------------ */
/* the operator                                                        */
/* ------------------------------------------------------------------ */

/* src/Hamiltonian.jl:211-273 apply_H!  (row-owner gather; per-row order:
 * fields i=1..L, zz in list order, value = diag*psi[idx], hops in list order
 * value += T(Jxy)*psi[new_idx], store).  nc = 1 (Float64) or 2 (ComplexF64,
 * interleaved).  N = length(psi); only length(out)==N is asserted (:220). */
int so_apply_H(const so_model *m, int nc, double *out, const double *psi, int64_t N) {
  const int L = m->L;
  const int full = (m->nup < 0);
  if (!full && N != m->N) return SO_EDIM;
#pragma omp parallel for schedule(static)
  for (int64_t idx = 0; idx < N; ++idx) {
    uint64_t state = full ? (uint64_t)idx : m->states[idx];
    double diag = 0.0;
    for (int i = 1; i <= L; ++i) diag += m->field[i - 1] * sz_value(bit_at(state, i - 1));
    for (int k = 0; k < m->n_zz; ++k)
      diag += m->zz_J[k] * sz_value(bit_at(state, m->zz_i[k] - 1)) *
              sz_value(bit_at(state, m->zz_j[k] - 1));
    double v0 = diag * psi[idx * nc], v1 = (nc == 2) ? diag * psi[idx * nc + 1] : 0.0;
    for (int k = 0; k < m->n_hop; ++k) {
      int i = m->hop_i[k], j = m->hop_j[k];
      if (bit_at(state, i - 1) != bit_at(state, j - 1)) {
        uint64_t ns = flip_bits(state, i - 1, j - 1);
        int64_t nidx = full ? (int64_t)ns + 1 : idx_lookup(m, ns);
        if (nidx != 0) {
          double J = m->hop_J[k];
          v0 += J * psi[(nidx - 1) * nc];
          if (nc == 2) v1 += J * psi[(nidx - 1) * nc + 1];
        }
      }
    }
    out[idx * nc] = v0;
    if (nc == 2) out[idx * nc + 1] = v1;
  }
  return SO_OK;
}

/* src/Hamiltonian.jl:286-301 apply_rescaled_H!: out = (H psi - b psi)/a with a
 * true division, as a separate (un-fused, serial in the reference) pass. */
int so_apply_rescaled_H(const so_model *m, int nc, double *out, const double *psi,
                        int64_t N, double a, double b) {
  int rc = so_apply_H(m, nc, out, psi, N);
  if (rc) return rc;
  int64_t n = N * nc;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) out[i] = (out[i] - b * psi[i]) / a;
  return SO_OK;
}

/* negated operator used by estimate_energy_bounds (src/Lanczos.jl:261-265) */
static int apply_H_sign(const so_model *m, int nc, double *out, const double *psi,
                        int64_t N, int negate) {
  int rc = so_apply_H(m, nc, out, psi, N);
  if (rc) return rc;
  if (negate) { int64_t n = N * nc; for (int64_t i = 0; i < n; ++i) out[i] = -out[i]; }
  return SO_OK;
}

/* src/Hamiltonian.jl:307-337 Sz_q_vector: phi[idx] = L^{-1/2} *
 * (sum_{r=0}^{L-1} e^{iqr} s_r) * ComplexF64(psi0[idx]); phases = exp.(im*q*(0:L-1)). */
int so_szq(const so_model *m, int nc_in, const double *psi0, int64_t N, double q,
           double *phi /* c128, N */) {
  const int L = m->L;
  const int full = (m->nup < 0);
  if (!full && N != m->N) return SO_EDIM;
  double normfact = 1.0 / sqrt((double)L);
  double pr[64], pi[64];
  for (int r = 0; r < L; ++r) { double x = q * (double)r; pr[r] = cos(x); pi[r] = sin(x); }
#pragma omp parallel for schedule(static)
  for (int64_t idx = 0; idx < N; ++idx) {
    uint64_t state = full ? (uint64_t)idx : m->states[idx];
    double sr = 0.0, si = 0.0;
    for (int r = 0; r < L; ++r) {
      double s = sz_value(bit_at(state, r));
      sr += pr[r] * s; si += pi[r] * s;
    }
    /* normfact * sq  (real * complex), then * ComplexF64(psi0) (complex * complex) */
    double ar = normfact * sr, ai = normfact * si;
    double xr = psi0[idx * nc_in], xi = (nc_in == 2) ? psi0[idx * nc_in + 1] : 0.0;
    phi[2 * idx] = ar * xr - ai * xi;
    phi[2 * idx + 1] = ar * xi + ai * xr;
  }
  return SO_OK;
}

/* ------------------------------------------------------------------ */
/* BLAS-1 style helpers (sequential sums; the reference's BLAS order is */
/* unspecified, hence tolerance-based parity on reductions)             */
/* ------------------------------------------------------------------ */
static double vnorm(const double *x, int64_t n) {
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += x[i] * x[i];
  return sqrt(s);
}
/* dot(x,y) = sum conj(x_i) y_i */
static void cdot(const double *x, const double *y, int64_t N, double *re, double *im) {
  double r = 0.0, s = 0.0;
  for (int64_t i = 0; i < N; ++i) {
    double xr = x[2 * i], xi = x[2 * i + 1], yr = y[2 * i], yi = y[2 * i + 1];
    r += xr * yr + xi * yi;
    s += xr * yi - xi * yr;
  }
  *re = r; *im = s;
}
static double rdot(const double *x, const double *y, int64_t n) {
  double r = 0.0;
  for (int64_t i = 0; i < n; ++i) r += x[i] * y[i];
  return r;
}

/* ------------------------------------------------------------------ */
/* symmetric tridiagonal eigen-solver (implicit QL with Wilkinson      */
/* shifts).  Stands in for LAPACK eigvals/eigen(SymTridiagonal) used at */
/* src/Lanczos.jl:80-83,164-165, src/TimeEvolution/Krylov.jl:175-176,   */
/* src/LanczosSqw.jl:23-24.  d[n] diagonal, e[n-1] off-diagonal.        */
/* On return w[n] ascending eigenvalues, z (n*n, column-major,          */
/* z[i + n*k] = component i of eigenvector k) if z != NULL.             */
/* ------------------------------------------------------------------ */
int so_symtridiag_eig(int n, const double *d_in, const double *e_in, double *w, double *z) {
  if (n <= 0) return SO_EARG;
  double *d = (double *)malloc(sizeof(double) * n);
  double *e = (double *)calloc(n, sizeof(double));
  for (int i = 0; i < n; ++i) d[i] = d_in[i];
  for (int i = 0; i + 1 < n; ++i) e[i] = e_in[i];
  if (z) { memset(z, 0, sizeof(double) * (size_t)n * n); for (int i = 0; i < n; ++i) z[i + (size_t)n * i] = 1.0; }
  for (int l = 0; l < n; ++l) {
    int iter = 0, mm;
    do {
      for (mm = l; mm < n - 1; ++mm) {
        double dd = fabs(d[mm]) + fabs(d[mm + 1]);
        if (fabs(e[mm]) <= 2.220446049250313e-16 * dd) break;
      }
      if (mm != l) {
        if (iter++ == 200) { free(d); free(e); return SO_EARG; }
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = hypot(g, 1.0);
        g = d[mm] - d[l] + e[l] / (g + (g >= 0 ? fabs(r) : -fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = mm - 1; i >= l; --i) {
          double f = s * e[i], b = c * e[i];
          e[i + 1] = (r = hypot(f, g));
          if (r == 0.0) { d[i + 1] -= p; e[mm] = 0.0; break; }
          s = f / r; c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          d[i + 1] = g + (p = s * r);
          g = c * r - b;
          if (z) for (int k = 0; k < n; ++k) {
            double f2 = z[k + (size_t)n * (i + 1)];
            z[k + (size_t)n * (i + 1)] = s * z[k + (size_t)n * i] + c * f2;
            z[k + (size_t)n * i] = c * z[k + (size_t)n * i] - s * f2;
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p; e[l] = g; e[mm] = 0.0;
      }
    } while (mm != l);
  }
  /* sort ascending (selection sort, n is small) */
  for (int i = 0; i < n - 1; ++i) {
    int k = i; double p = d[i];
    for (int j = i + 1; j < n; ++j) if (d[j] < p) { k = j; p = d[j]; }
    if (k != i) {
      d[k] = d[i]; d[i] = p;
      if (z) for (int j = 0; j < n; ++j) {
        double t = z[j + (size_t)n * i]; z[j + (size_t)n * i] = z[j + (size_t)n * k]; z[j + (size_t)n * k] = t;
      }
    }
  }
  for (int i = 0; i < n; ++i) w[i] = d[i];
  free(d); free(e);
  return SO_OK;
}

/* ------------------------------------------------------------------ */
/* Lanczos family                                                      */
/* ------------------------------------------------------------------ */

/* src/Lanczos.jl:27-84 lanczos_extremal.  The reference draws
 * psi0 = randn(rng, ComplexF64, N) (:39); Julia's stream cannot be reproduced
 * outside Julia, so the start vector is injected (un-normalised, c128).
 * negate != 0 runs on -H (the closure at :261-265). */
int so_lanczos_extremal(const so_model *m, int lanc_m, double tol, const double *psi0,
                        int negate, double *emin, double *emax, int *m_used) {
  int64_t N = m->N;
  int mm = lanc_m < N ? lanc_m : (int)N;
  if (mm < 1) return SO_EARG;
  double *alpha = (double *)calloc(mm, sizeof(double));
  double *beta = (double *)calloc(mm, sizeof(double));
  double *v_prev = (double *)malloc(sizeof(double) * 2 * N);
  double *w = (double *)malloc(sizeof(double) * 2 * N);
  double *v_curr = (double *)calloc(2 * N, sizeof(double));
  double nrm = vnorm(psi0, 2 * N);
  for (int64_t i = 0; i < 2 * N; ++i) v_prev[i] = psi0[i] / nrm;      /* :40 */
  int actual = mm;
  for (int j = 1; j <= mm; ++j) {
    apply_H_sign(m, 2, w, v_prev, N, negate);                         /* :51 */
    double re, im; cdot(v_prev, w, N, &re, &im);
    alpha[j - 1] = re;                                                /* :55 */
    if (j == 1) {
      for (int64_t i = 0; i < 2 * N; ++i) w[i] -= alpha[0] * v_prev[i];          /* :59 */
    } else {
      double a = alpha[j - 1], b = beta[j - 2];
      for (int64_t i = 0; i < 2 * N; ++i) w[i] -= a * v_prev[i] + b * v_curr[i]; /* :61 */
    }
    if (j < mm) {
      beta[j - 1] = vnorm(w, 2 * N);                                  /* :65 */
      if (beta[j - 1] < tol) { actual = j; break; }                   /* :66-70 */
      /* v_curr, v_prev = v_prev, w / beta[j]   (:71) */
      double *t = v_curr; v_curr = v_prev; v_prev = t;
      double bj = beta[j - 1];
      for (int64_t i = 0; i < 2 * N; ++i) v_prev[i] = w[i] / bj;
    }
  }
  double *ev = (double *)malloc(sizeof(double) * actual);
  int rc = so_symtridiag_eig(actual, alpha, beta, ev, NULL);           /* :80-83 */
  if (!rc) { *emin = ev[0]; *emax = ev[actual - 1]; }
  if (m_used) *m_used = actual;
  free(ev); free(alpha); free(beta); free(v_prev); free(w); free(v_curr);
  return rc;
}

/* src/Lanczos.jl:255-271 estimate_energy_bounds: Emax from H, Emin = -Emax(-H);
 * two independent random starts in the reference (rng not forwarded) ->
 * two injected start vectors here. */
int so_estimate_energy_bounds(const so_model *m, int lanc_m, const double *psi0_a,
                              const double *psi0_b, double *Emin, double *Emax) {
  double lo, hi;
  int rc = so_lanczos_extremal(m, lanc_m, 1e-12, psi0_a, 0, &lo, &hi, NULL);
  if (rc) return rc;
  *Emax = hi;
  rc = so_lanczos_extremal(m, lanc_m, 1e-12, psi0_b, 1, &lo, &hi, NULL);
  if (rc) return rc;
  *Emin = -hi;
  return SO_OK;
}

/* src/Lanczos.jl:87-181 lanczos_groundstate (full re-orthogonalisation, real
 * Float64 vectors, V stored as N x m column-major).  psi0 injected (:99). */
int so_lanczos_groundstate(const so_model *m, int lanc_m, double tol, double orth_tol,
                           const double *psi0, double *E0, double *psi_gs, int *m_actual_out) {
  int64_t N = m->N;
  int mm = lanc_m < N ? lanc_m : (int)N;
  if (mm < 1) return SO_EARG;
  double *alpha = (double *)calloc(mm, sizeof(double));
  double *beta = (double *)calloc(mm, sizeof(double));
  double *V = (double *)malloc(sizeof(double) * (size_t)N * mm);
  double *w = (double *)malloc(sizeof(double) * N);
  double *tmp = (double *)malloc(sizeof(double) * N);
  double nrm = vnorm(psi0, N);
  for (int64_t i = 0; i < N; ++i) V[i] = psi0[i] / nrm;                /* :100,105 */
  int m_actual = mm;
  for (int j = 1; j <= mm; ++j) {
    double *vj = V + (size_t)N * (j - 1);
    so_apply_H(m, 1, w, vj, N);                                        /* :113 */
    for (int k = 1; k <= j - 1; ++k) {                                 /* :116-122 */
      double *vk = V + (size_t)N * (k - 1);
      double coeff = rdot(vk, w, N);
      for (int64_t i = 0; i < N; ++i) w[i] -= coeff * vk[i];
    }
    alpha[j - 1] = rdot(vj, w, N);                                     /* :124 */
    if (j == 1) {
      for (int64_t i = 0; i < N; ++i) w[i] = w[i] - alpha[0] * vj[i];  /* :127 */
    } else {
      double *vp = V + (size_t)N * (j - 2);
      double a = alpha[j - 1], b = beta[j - 2];
      for (int64_t i = 0; i < N; ++i) w[i] = w[i] - a * vj[i] - b * vp[i]; /* :129 */
    }
    if (j < mm) {
      beta[j - 1] = vnorm(w, N);                                       /* :133 */
      if (beta[j - 1] < tol) { m_actual = j; break; }                  /* :136-139 */
      for (int k = 1; k <= j; ++k) {                                   /* :142-153 */
        double *vk = V + (size_t)N * (k - 1);
        double bj = beta[j - 1];
        for (int64_t i = 0; i < N; ++i) tmp[i] = w[i] / bj;
        double overlap = fabs(rdot(vk, tmp, N));
        if (overlap > orth_tol) {
          double c = rdot(vk, w, N);
          for (int64_t i = 0; i < N; ++i) w[i] -= c * vk[i];
          beta[j - 1] = vnorm(w, N);
          if (beta[j - 1] < tol) { m_actual = j; break; }  /* inner break only (:150) */
        }
      }
      double bj = beta[j - 1];
      double *vn = V + (size_t)N * j;
      for (int64_t i = 0; i < N; ++i) vn[i] = w[i] / bj;                /* :155 */
    }
  }
  int nb = m_actual - 1 < mm - 1 ? m_actual - 1 : mm - 1;              /* :161 */
  (void)nb;
  double *ev = (double *)malloc(sizeof(double) * m_actual);
  double *Z = (double *)malloc(sizeof(double) * (size_t)m_actual * m_actual);
  int rc = so_symtridiag_eig(m_actual, alpha, beta, ev, Z);            /* :164-165 */
  if (!rc) {
    *E0 = ev[0];                                                       /* findmin :167 */
    for (int64_t i = 0; i < N; ++i) {                                  /* :170 */
      double s = 0.0;
      for (int k = 0; k < m_actual; ++k) s += V[i + (size_t)N * k] * Z[k];
      psi_gs[i] = s;
    }
    double n2 = vnorm(psi_gs, N);
    for (int64_t i = 0; i < N; ++i) psi_gs[i] /= n2;                   /* :171 */
  }
  if (m_actual_out) *m_actual_out = m_actual;
  free(ev); free(Z); free(alpha); free(beta); free(V); free(w); free(tmp);
  return rc;
}

/* src/Lanczos.jl:196-246 lanczos_tridiag (complex start vector v, not
 * normalised).  alpha_out[m], beta_out[m-1]; *m_eff = length(alpha). */
int so_lanczos_tridiag(const so_model *m, const double *v, int64_t n, int lanc_m, double tol,
                       double *alpha, double *beta, int *m_eff_out, double *normv_out) {
  int mm = lanc_m < n ? lanc_m : (int)n;
  if (mm < 1) return SO_EARG;
  double normv = vnorm(v, 2 * n);
  if (normv == 0) return SO_EZERO;                                     /* :210-212 */
  double *vprev = (double *)calloc(2 * n, sizeof(double));
  double *vcur = (double *)malloc(sizeof(double) * 2 * n);
  double *w = (double *)calloc(2 * n, sizeof(double));
  for (int64_t i = 0; i < 2 * n; ++i) vcur[i] = v[i] / normv;          /* :214 */
  for (int k = 0; k < mm; ++k) alpha[k] = 0.0;
  for (int k = 0; k + 1 < mm; ++k) beta[k] = 0.0;
  int m_eff = mm;
  for (int j = 1; j <= mm - 1; ++j) {
    so_apply_H(m, 2, w, vcur, n);                                      /* :218 */
    double re, im; cdot(vcur, w, n, &re, &im);
    alpha[j - 1] = re;                                                 /* :219 */
    { double a = alpha[j - 1]; for (int64_t i = 0; i < 2 * n; ++i) w[i] -= a * vcur[i]; } /* :222 */
    if (j > 1) { double b = beta[j - 2]; for (int64_t i = 0; i < 2 * n; ++i) w[i] -= b * vprev[i]; } /* :224 */
    beta[j - 1] = vnorm(w, 2 * n);                                     /* :227 */
    if (beta[j - 1] < tol) { m_eff = j; break; }                       /* :228-231 */
    double bj = beta[j - 1];
    double *t = vprev; vprev = vcur; vcur = t;
    for (int64_t i = 0; i < 2 * n; ++i) vcur[i] = w[i] / bj;           /* :233 */
  }
  if (m_eff == mm) {                                                   /* :237-239 */
    so_apply_H(m, 2, w, vcur, n);
    double re, im; cdot(vcur, w, n, &re, &im);
    alpha[mm - 1] = re;
  }
  *m_eff_out = m_eff; *normv_out = normv;
  free(vprev); free(vcur); free(w);
  return SO_OK;
}

/* ------------------------------------------------------------------ */
/* Krylov time evolution                                               */
/* ------------------------------------------------------------------ */

/* src/TimeEvolution/Krylov.jl:136-192 krylov_time_evolve.  nc: components of
 * psi0 (1 real, 2 complex).  Output ComplexF64, normalised (:190).
 * Deviation (documented): the reference stores alpha = dot(V[j], w) as
 * ComplexF64, diagonalises the general complex tridiagonal and uses Q' as
 * Q^-1 (:141-142,175-181); here the real part of alpha enters the symmetric
 * tridiagonal solver (the imaginary part is rounding noise of a Hermitian
 * Rayleigh quotient), while the full complex alpha is used in the vector
 * update (:156) as in the reference. */
int so_krylov_time_evolve(const so_model *m, int nc, const double *psi0, double dt, int kry_m,
                          double *psit /* c128 */) {
  int64_t n = m->N;
  if (kry_m < 1) return SO_EARG;
  double norm0 = vnorm(psi0, nc * n);
  if (norm0 == 0) {                                                    /* :145-147 */
    for (int64_t i = 0; i < n; ++i) { psit[2 * i] = psi0[nc * i]; psit[2 * i + 1] = nc == 2 ? psi0[2 * i + 1] : 0.0; }
    return SO_OK;
  }
  /* work in complex throughout; a real psi0 keeps exactly-zero imaginary parts */
  double **V = (double **)calloc(kry_m, sizeof(double *));
  double *alr = (double *)calloc(kry_m, sizeof(double));
  double *ali = (double *)calloc(kry_m, sizeof(double));
  double *beta = (double *)calloc(kry_m, sizeof(double));
  double *w = (double *)calloc(2 * n, sizeof(double));
  V[0] = (double *)malloc(sizeof(double) * 2 * n);
  for (int64_t i = 0; i < n; ++i) {
    V[0][2 * i] = psi0[nc * i] / norm0;
    V[0][2 * i + 1] = nc == 2 ? psi0[2 * i + 1] / norm0 : 0.0;
  }
  int m_eff = kry_m;
  for (int j = 1; j <= kry_m; ++j) {
    so_apply_H(m, 2, w, V[j - 1], n);                                  /* :153 */
    cdot(V[j - 1], w, n, &alr[j - 1], &ali[j - 1]);                    /* :155 */
    {
      double ar = alr[j - 1], ai = ali[j - 1]; const double *vj = V[j - 1];
      for (int64_t i = 0; i < n; ++i) {                                /* :156 */
        double xr = vj[2 * i], xi = vj[2 * i + 1];
        w[2 * i] -= ar * xr - ai * xi;
        w[2 * i + 1] -= ar * xi + ai * xr;
      }
    }
    if (j > 1) {                                                       /* :157-159 */
      double b = beta[j - 2]; const double *vp = V[j - 2];
      for (int64_t i = 0; i < 2 * n; ++i) w[i] -= b * vp[i];
    }
    if (j < kry_m) {
      beta[j - 1] = vnorm(w, 2 * n);                                   /* :161 */
      if (fabs(beta[j - 1]) < 1e-14) { m_eff = j; break; }             /* :162-168 */
      V[j] = (double *)malloc(sizeof(double) * 2 * n);
      double bj = beta[j - 1];
      for (int64_t i = 0; i < 2 * n; ++i) V[j][i] = w[i] / bj;         /* :169 */
    }
  }
  double *ev = (double *)malloc(sizeof(double) * m_eff);
  double *Q = (double *)malloc(sizeof(double) * (size_t)m_eff * m_eff);
  int rc = so_symtridiag_eig(m_eff, alr, beta, ev, Q);                 /* :175-178 */
  if (!rc) {
    /* y = Q exp(-i D dt) Q' (norm0 e1)  (:180-182) */
    double *yr = (double *)calloc(m_eff, sizeof(double)), *yi = (double *)calloc(m_eff, sizeof(double));
    for (int l = 0; l < m_eff; ++l) {
      double ph = -ev[l] * dt, cr = cos(ph), ci = sin(ph);
      double q0 = Q[0 + (size_t)m_eff * l] * norm0;
      for (int k = 0; k < m_eff; ++k) {
        double qk = Q[k + (size_t)m_eff * l];
        yr[k] += qk * cr * q0; yi[k] += qk * ci * q0;
      }
    }
    for (int64_t i = 0; i < 2 * n; ++i) psit[i] = 0.0;                 /* :185 */
    for (int k = 0; k < m_eff; ++k) {                                  /* :186-188 */
      const double *vk = V[k]; double a = yr[k], b = yi[k];
      for (int64_t i = 0; i < n; ++i) {
        double xr = vk[2 * i], xi = vk[2 * i + 1];
        psit[2 * i] += a * xr - b * xi;
        psit[2 * i + 1] += a * xi + b * xr;
      }
    }
    double nn = vnorm(psit, 2 * n);
    for (int64_t i = 0; i < 2 * n; ++i) psit[i] /= nn;                 /* :190 */
    free(yr); free(yi);
  }
  for (int k = 0; k < kry_m; ++k) free(V[k]);
  free(V); free(alr); free(ali); free(beta); free(w); free(ev); free(Q);
  return rc;
}

/* ------------------------------------------------------------------ */
/* Chebyshev time evolution                                            */
/* ------------------------------------------------------------------ */

/* Coefficients c_k = (2 - delta_k0) (-i)^k J_k(a dt) exp(-i b dt)
 * (src/TimeEvolution/Chebyshev.jl:74-79); besselj -> libm jn. */
void so_chebyshev_coeffs(int cheb_n, double a, double b, double dt, double *c /* 2*cheb_n */) {
  double ph = b * dt, pr = cos(ph), pi = -sin(ph);
  for (int k = 0; k < cheb_n; ++k) {
    double f = (k == 0) ? 1.0 : 2.0;
    double J = jn(k, a * dt);
    double xr, xi;
    switch (k & 3) {                 /* (-i)^k */
      case 0: xr = f; xi = 0; break;
      case 1: xr = 0; xi = -f; break;
      case 2: xr = -f; xi = 0; break;
      default: xr = 0; xi = f; break;
    }
    xr *= J; xi *= J;
    c[2 * k] = xr * pr - xi * pi;
    c[2 * k + 1] = xr * pi + xi * pr;
  }
}

/* src/TimeEvolution/Chebyshev.jl:61-124 chebyshev_time_evolve (psi0 must be
 * ComplexF64; not renormalised). a=(Emax-Emin)/(2*0.9999), b=(Emax+Emin)/2. */
int so_chebyshev_time_evolve(const so_model *m, const double *psi0, double dt, int cheb_n,
                             double Emin, double Emax, double *psit) {
  if (cheb_n < 1) return SO_EARG;                                      /* :65 */
  int64_t N = m->N;
  double a = (Emax - Emin) / (2 * 0.9999), b = (Emax + Emin) / 2;      /* :70-71 */
  double *c = (double *)malloc(sizeof(double) * 2 * cheb_n);
  so_chebyshev_coeffs(cheb_n, a, b, dt, c);
  double *pprev = (double *)malloc(sizeof(double) * 2 * N);
  double *pcur = (double *)malloc(sizeof(double) * 2 * N);
  double *pnext = (double *)malloc(sizeof(double) * 2 * N);
  memcpy(pprev, psi0, sizeof(double) * 2 * N);                         /* :90 */
  so_apply_rescaled_H(m, 2, pcur, pprev, N, a, b);                     /* :93 */
  for (int64_t i = 0; i < N; ++i) {                                    /* :96-102 */
    double tr = 0.0, ti = 0.0;
    double xr = pprev[2 * i], xi = pprev[2 * i + 1];
    tr += c[0] * xr - c[1] * xi; ti += c[0] * xi + c[1] * xr;
    if (cheb_n >= 2) {
      double yr = pcur[2 * i], yi = pcur[2 * i + 1];
      tr += c[2] * yr - c[3] * yi; ti += c[2] * yi + c[3] * yr;
    }
    psit[2 * i] = tr; psit[2 * i + 1] = ti;
  }
  for (int k = 2; k <= cheb_n - 1; ++k) {                              /* :110-121 */
    so_apply_rescaled_H(m, 2, pnext, pcur, N, a, b);
    double cr = c[2 * k], ci = c[2 * k + 1];
    for (int64_t i = 0; i < N; ++i) {
      double nr = 2 * pnext[2 * i] - pprev[2 * i];
      double ni = 2 * pnext[2 * i + 1] - pprev[2 * i + 1];
      pnext[2 * i] = nr; pnext[2 * i + 1] = ni;
      psit[2 * i] += cr * nr - ci * ni;
      psit[2 * i + 1] += cr * ni + ci * nr;
    }
    double *t = pprev; pprev = pcur; pcur = pnext; pnext = t;
  }
  free(c); free(pprev); free(pcur); free(pnext);
  return SO_OK;
}

/* ------------------------------------------------------------------ */
/* KPM                                                                 */
/* ------------------------------------------------------------------ */

/* src/KPM_Sqw.jl:13-17 _rescaling_from_bounds */
void so_rescaling_from_bounds(double Emin, double Emax, double *a, double *b) {
  *a = (Emax - Emin) / (2 * 0.99);
  *b = (Emax + Emin) / 2;
}

/* src/KPM_Sqw.jl:131-145 get_kernel: kind 0 jackson, 1 lorentz (lambda=3),
 * anything else -> ones. */
void so_get_kernel(int M, int kind, double *g) {
  const double PI = 3.14159265358979323846;
  for (int n = 0; n < M; ++n) g[n] = 1.0;
  if (kind == 0) {
    for (int n = 0; n < M; ++n)
      g[n] = ((M - n + 1) * cos(PI * n / (M + 1)) +
              sin(PI * n / (M + 1)) * (1.0 / tan(PI / (M + 1)))) / (M + 1);
  } else if (kind == 1) {
    double lam = 3.0;
    for (int n = 0; n < M; ++n) g[n] = sinh(lam * (1 - (double)n / M)) / sinh(lam);
  }
}

/* src/KPM_Sqw.jl:95-128 compute_chebyshev_moments */
int so_chebyshev_moments(const so_model *m, const double *phi, int M, double a, double b,
                         double *mu) {
  int64_t N = m->N;
  if (M < 2) return SO_EARG;   /* mu[2] is written unconditionally (:107) */
  double *vprev = (double *)malloc(sizeof(double) * 2 * N);
  double *vcur = (double *)malloc(sizeof(double) * 2 * N);
  double *vnext = (double *)malloc(sizeof(double) * 2 * N);
  memcpy(vprev, phi, sizeof(double) * 2 * N);
  for (int k = 0; k < M; ++k) mu[k] = 0.0;
  double re, im;
  cdot(phi, vprev, N, &re, &im); mu[0] = re;                           /* :103 */
  so_apply_rescaled_H(m, 2, vcur, vprev, N, a, b);                     /* :106 */
  cdot(phi, vcur, N, &re, &im); mu[1] = re;                            /* :107 */
  for (int mm = 2; mm <= M - 1; ++mm) {
    so_apply_rescaled_H(m, 2, vnext, vcur, N, a, b);                   /* :111 */
    for (int64_t i = 0; i < 2 * N; ++i) vnext[i] = 2.0 * vnext[i] - vprev[i]; /* :112 */
    cdot(phi, vnext, N, &re, &im); mu[mm] = re;                        /* :114 */
    double nv = vnorm(vnext, 2 * N);                                   /* :117 */
    if (nv > 1e3) for (int64_t i = 0; i < 2 * N; ++i) vnext[i] /= nv;  /* :118-121 */
    double *t = vprev; vprev = vcur; vcur = vnext; vnext = t;          /* :124 */
  }
  free(vprev); free(vcur); free(vnext);
  return SO_OK;
}

/* src/KPM_Sqw.jl:55-90: reconstruction from (already damped) moments. */
void so_kpm_reconstruct(const double *mu_damped, int kpm_m, const double *omega, int W,
                        double a, double b, double E0, double *S) {
  const double PI = 3.14159265358979323846;
  double *T = (double *)malloc(sizeof(double) * (kpm_m > 2 ? kpm_m : 2));
  for (int iw = 0; iw < W; ++iw) {
    double x = (omega[iw] + E0 - b) / a;                               /* :61 */
    if (fabs(x) >= 1.0) { S[iw] = 0.0; continue; }                     /* :63-66 */
    T[0] = 1.0;
    if (kpm_m >= 2) T[1] = x;
    for (int n = 2; n < kpm_m; ++n) T[n] = 2.0 * x * T[n - 1] - T[n - 2]; /* :75-77 */
    double sum_val = mu_damped[0] * T[0];
    for (int n = 1; n < kpm_m; ++n) sum_val += 2.0 * mu_damped[n] * T[n]; /* :81-83 */
    double denom = PI * sqrt(1.0 - x * x);
    double v = sum_val / (a * denom);
    S[iw] = v > 0.0 ? v : 0.0;                                         /* :89 */
  }
  free(T);
}

/* src/KPM_Sqw.jl:34-93 kpm_sw */
int so_kpm_sw(const so_model *m, const double *phi, const double *omega, int W, double a,
              double b, double E0, int kpm_m, int kernel, double *S) {
  double *mu = (double *)malloc(sizeof(double) * kpm_m);
  double *g = (double *)malloc(sizeof(double) * kpm_m);
  int rc = so_chebyshev_moments(m, phi, kpm_m, a, b, mu);
  if (!rc) {
    so_get_kernel(kpm_m, kernel, g);
    for (int n = 0; n < kpm_m; ++n) mu[n] *= g[n];                     /* :53 */
    so_kpm_reconstruct(mu, kpm_m, omega, W, a, b, E0, S);
  }
  free(mu); free(g);
  return rc;
}

/* src/KPM_Sqw.jl:191-256 kpm_sqw with explicit (a,b) (the estimated ones are
 * random in the reference; parity runs inject them).  Smat is Qn x W,
 * row-major here (Smat[iq*W + iw]). */
int so_kpm_sqw(const so_model *m, int nc, const double *psi0, const double *q, int Qn,
               const double *omega, int W, double a, double b, int kpm_m, int kernel,
               double *Smat) {
  int64_t N = m->N;
  double *psic = (double *)malloc(sizeof(double) * 2 * N);
  double *tmp = (double *)malloc(sizeof(double) * 2 * N);
  double *phi = (double *)malloc(sizeof(double) * 2 * N);
  for (int64_t i = 0; i < N; ++i) { psic[2 * i] = psi0[nc * i]; psic[2 * i + 1] = nc == 2 ? psi0[2 * i + 1] : 0.0; }
  so_apply_H(m, 2, tmp, psic, N);                                      /* :208 */
  double E0, im; cdot(psic, tmp, N, &E0, &im);                         /* :209 */
  int rc = SO_OK;
  for (int iq = 0; iq < Qn && !rc; ++iq) {
    so_szq(m, 2, psic, N, q[iq], phi);                                 /* :223 */
    double norm_phi = vnorm(phi, 2 * N);
    if (norm_phi == 0) { for (int iw = 0; iw < W; ++iw) Smat[(size_t)iq * W + iw] = 0.0; continue; }
    for (int64_t i = 0; i < 2 * N; ++i) phi[i] /= norm_phi;            /* :231 */
    rc = so_kpm_sw(m, phi, omega, W, a, b, E0, kpm_m, kernel, Smat + (size_t)iq * W);
    double n2 = norm_phi * norm_phi;
    for (int iw = 0; iw < W; ++iw) Smat[(size_t)iq * W + iw] *= n2;    /* :252 */
  }
  free(psic); free(tmp); free(phi);
  return rc;
}

/* ------------------------------------------------------------------ */
/* Lanczos S(q,w) ("next" row f1)                                      */
/* ------------------------------------------------------------------ */

/* src/LanczosSqw.jl:18-43 spectral_from_tridiagonal; broaden 0 lorentz, 1 gauss */
int so_spectral_from_tridiagonal(const double *alpha, const double *beta, int mt, double norm_phi,
                                 double E0, const double *omega, int W, double eta, int broaden,
                                 double *S) {
  const double PI = 3.14159265358979323846;
  if (broaden != 0 && broaden != 1) return SO_EARG;
  double *th = (double *)malloc(sizeof(double) * mt);
  double *Q = (double *)malloc(sizeof(double) * (size_t)mt * mt);
  int rc = so_symtridiag_eig(mt, alpha, beta, th, Q);
  if (!rc) {
    for (int iw = 0; iw < W; ++iw) {
      double s = 0.0;
      for (int k = 0; k < mt; ++k) {
        double q1 = Q[0 + (size_t)mt * k];
        double wgt = q1 * q1 * (norm_phi * norm_phi);
        double sh = omega[iw] - (th[k] - E0);
        double f = broaden == 0 ? (1 / PI) * (eta / (sh * sh + eta * eta))
                                : (1 / (sqrt(2 * PI) * eta)) * exp(-(sh * sh) / (2 * eta * eta));
        s += f * wgt;
      }
      S[iw] = s;
    }
  }
  free(th); free(Q);
  return rc;
}

/* src/LanczosSqw.jl:49-80 lanczos_sqw */
int so_lanczos_sqw(const so_model *m, int nc, const double *psi0, const double *q, int Qn,
                   const double *omega, int W, int lanc_m, double eta, int broaden, double *Smat) {
  int64_t N = m->N;
  double *psic = (double *)malloc(sizeof(double) * 2 * N);
  double *tmp = (double *)malloc(sizeof(double) * 2 * N);
  double *phi = (double *)malloc(sizeof(double) * 2 * N);
  int mm = lanc_m < N ? lanc_m : (int)N;
  double *alpha = (double *)malloc(sizeof(double) * mm), *beta = (double *)malloc(sizeof(double) * mm);
  if (N < 1 || !psic || !tmp || !phi || !alpha || !beta) {
    free(psic); free(tmp); free(phi); free(alpha); free(beta);
    return SO_EARG;
  }
  for (int64_t i = 0; i < N; ++i) { psic[2 * i] = psi0[nc * i]; psic[2 * i + 1] = nc == 2 ? psi0[2 * i + 1] : 0.0; }
  so_apply_H(m, 2, tmp, psic, N);                                      /* :58 */
  /* E0 = real(dot(conj(psi0c), tmp)) = Re sum psi_i * tmp_i  (:59, sic) */
  double E0 = 0.0;
  for (int64_t i = 0; i < N; ++i) E0 += psic[2 * i] * tmp[2 * i] - psic[2 * i + 1] * tmp[2 * i + 1];
  int rc = SO_OK;
  for (int iq = 0; iq < Qn && !rc; ++iq) {
    so_szq(m, 2, psic, N, q[iq], phi);
    if (vnorm(phi, 2 * N) == 0) { for (int iw = 0; iw < W; ++iw) Smat[(size_t)iq * W + iw] = 0.0; continue; }
    int m_eff; double normv;
    rc = so_lanczos_tridiag(m, phi, N, lanc_m, 1e-12, alpha, beta, &m_eff, &normv);
    if (rc) break;
    rc = so_spectral_from_tridiagonal(alpha, beta, m_eff, normv, E0, omega, W, eta, broaden,
                                      Smat + (size_t)iq * W);
  }
  free(psic); free(tmp); free(phi); free(alpha); free(beta);
  return rc;
}


/* ------------------------------------------------------------------ */
/* Observables ("next" row f2) and InitialStates (f3)                  */
/* ------------------------------------------------------------------ */

/* src/Observables.jl:14-36 magnetization_per_site: mags[i] = sum_idx |psi|^2 s_i */
int so_magnetization_per_site(const so_model *m, int nc, const double *psi, int64_t N, double *mags) {
  const int L = m->L, full = (m->nup < 0);
  if (!full && N != m->N) return SO_EDIM;
  for (int i = 0; i < L; ++i) mags[i] = 0.0;
  for (int64_t idx = 0; idx < N; ++idx) {
    double re = psi[idx * nc], im = nc == 2 ? psi[idx * nc + 1] : 0.0;
    double prob = re * re + im * im;
    if (prob != 0.0) {
      uint64_t state = full ? (uint64_t)idx : m->states[idx];
      for (int i = 0; i < L; ++i) mags[i] += prob * sz_value(bit_at(state, i));
    }
  }
  return SO_OK;
}

/* src/Observables.jl:44-94 connected_correlations: C_r = (1/L) sum_i (<S_i S_j> - <S_i><S_j>), j = mod1(i+r, L)
 * (periodic wrap even for open chains, :86) */
int so_connected_correlations(const so_model *m, int nc, const double *psi, int64_t N, double *C_r) {
  const int L = m->L, full = (m->nup < 0);
  if (!full && N != m->N) return SO_EDIM;
  double *SzSz = (double *)calloc((size_t)L * L, sizeof(double));
  double *S_i = (double *)calloc(L, sizeof(double));
  for (int64_t idx = 0; idx < N; ++idx) {
    double re = psi[idx * nc], im = nc == 2 ? psi[idx * nc + 1] : 0.0;
    double amp2 = re * re + im * im;
    if (amp2 == 0.0) continue;
    uint64_t state = full ? (uint64_t)idx : m->states[idx];
    for (int i = 0; i < L; ++i) {
      double szi = sz_value(bit_at(state, i));
      S_i[i] += amp2 * szi;
      for (int j = 0; j < L; ++j) SzSz[i + (size_t)L * j] += amp2 * szi * sz_value(bit_at(state, j));
    }
  }
  for (int r = 0; r < L; ++r) {
    double tmp = 0.0;
    for (int i = 1; i <= L; ++i) {
      int j = ((i + r - 1) % L) + 1;                /* mod1(i+r, L) */
      tmp += SzSz[(i - 1) + (size_t)L * (j - 1)] - S_i[i - 1] * S_i[j - 1];
    }
    C_r[r] = tmp / L;
  }
  free(SzSz); free(S_i);
  return SO_OK;
}

/* src/Observables.jl:100-109 structure_factor_Sq: real(fft(C_r))[n], q_n = 2 pi (n-1)/L  (plain DFT, L is tiny) */
int so_structure_factor_Sq(const so_model *m, int nc, const double *psi, int64_t N, double *q, double *Sq) {
  const double PI = 3.14159265358979323846;
  const int L = m->L;
  double C[64];
  int rc = so_connected_correlations(m, nc, psi, N, C);
  if (rc) return rc;
  for (int n = 0; n < L; ++n) {
    double s = 0.0;
    for (int r = 0; r < L; ++r) s += C[r] * cos(2 * PI * n * r / L);
    q[n] = 2 * PI * n / L;
    Sq[n] = s;
  }
  return SO_OK;
}

/* src/InitialStates.jl:9-130: one-hot Float64 vectors.  kind: 0 domain wall (:9-34), 1 Neel (:40-63),
 * 2 polarized up, 3 polarized down (:70-89), 4 polarized with flips (:97-130; flips are 1-based sites).
 * Returns SO_EARG when the configuration is not in the basis (ArgumentError in the reference). */
int so_initial_state(const so_model *m, int kind, const int *flips, int nflips, double *psi0) {
  const int L = m->L;
  uint64_t s = 0;
  if (kind == 0) {
    int nup = m->nup >= 0 ? m->nup : (L + 1) / 2;        /* Int(ceil(L/2)) */
    for (int i = 0; i < nup; ++i) s |= (uint64_t)1 << i;
  } else if (kind == 1) {
    for (int i = 0; i < L; ++i) if (((i + 1) & 1) == 1) s |= (uint64_t)1 << i;
  } else if (kind == 2) {
    s = ((uint64_t)1 << L) - 1;
  } else if (kind == 3) {
    s = 0;
  } else if (kind == 4) {
    for (int k = 0; k < nflips; ++k) if (flips[k] < 1 || flips[k] > L) return SO_EARG;
    s = ((uint64_t)1 << L) - 1;
    for (int k = 0; k < nflips; ++k) s ^= (uint64_t)1 << (flips[k] - 1);
  } else return SO_EARG;
  int64_t idx = so_lookup(m, s);
  if (idx == 0) return SO_EARG;
  for (int64_t i = 0; i < m->N; ++i) psi0[i] = 0.0;
  psi0[idx - 1] = 1.0;
  return SO_OK;
}

void so_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int so_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
