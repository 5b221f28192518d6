/* A plain C99 consumer of include/spindyn.h, compiled with gcc (-Wall -Werror -pedantic) and linked against
 * libspindyn.so: what a foreign-language binding (the Julia ccall stubs of INTEGRATION.md) sees.  The C compiler checks
 * every argument TYPE of the prototypes used here, which the name-only comparison of tests/test_cabi_host.py cannot.
 *
 * usage: cabi_consumer <fixture.bin>
 * fixture (little endian, written by tests/test_cabi_consumer.py from tests/golden/L12n6_open.npz, i.e. from the
 * independent dense numpy oracle): int64 L, nup, N, W, M; double Jxy, Jz, hz, t, Emin, Emax, a, b;
 * uint64 states[N]; c128 psi[N]; c128 Hpsi[N]; c128 psi0[N]; c128 expm_psi0[N]; c128 gs[N]; double mu[M].
 * exit status: 0 all checks passed, 77 no GPU (the library has no CPU fallback), 1 a check failed. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spindyn.h"

#define CHECK(call)                                                                              \
  do {                                                                                           \
    int rc_ = (call);                                                                            \
    if (rc_ != SD_OK) {                                                                          \
      fprintf(stderr, "%s -> %d (%s): %s\n", #call, rc_, sd_status_string(rc_), ctx ? sd_last_error(ctx) : ""); \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)

static double maxdiff(const double *a, const double *b, int64_t n) {
  double d = 0.0;
  int64_t i;
  for (i = 0; i < n; ++i) { double x = fabs(a[i] - b[i]); if (x > d) d = x; }
  return d;
}

/* the reference's applyH! argument as a C callback (sd_apply_fn): forwards to the built-in operator and counts its calls */
struct fwd { sd_ctx *ctx; const sd_model *model; int calls; };
static int forward_apply(void *user, int dtype, void *out_dev, const void *psi_dev, int64_t n, void *hip_stream) {
  struct fwd *f = (struct fwd *)user;
  (void)hip_stream;                       /* sd_apply_dev runs on the context's stream, which is the one handed in */
  f->calls++;
  return sd_apply_dev(f->ctx, f->model, dtype, out_dev, psi_dev, n);
}

int main(int argc, char **argv) {
  sd_ctx *ctx = NULL;
  sd_model *model = NULL;
  FILE *f;
  int64_t hdr[5], L, nup, N, M;
  double par[8];
  uint64_t *states, *got_states;
  double *psi, *Hpsi, *psi0, *expm, *gs, *mu, *out, *phi, *mu_got, nrm = 0.0;
  int64_t i;
  printf("%s, %d device(s)\n", sd_version(), sd_device_count());
  if (sd_device_count() <= 0) { printf("SKIP: no GPU\n"); return 77; }
  if (argc < 2) { fprintf(stderr, "usage: %s fixture.bin\n", argv[0]); return 1; }
  f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  if (fread(hdr, sizeof(int64_t), 5, f) != 5 || fread(par, sizeof(double), 8, f) != 8) return 1;
  L = hdr[0]; nup = hdr[1]; N = hdr[2]; M = hdr[4];
  states = malloc(sizeof(uint64_t) * (size_t)N); got_states = malloc(sizeof(uint64_t) * (size_t)N);
  psi = malloc(16 * (size_t)N); Hpsi = malloc(16 * (size_t)N); psi0 = malloc(16 * (size_t)N);
  expm = malloc(16 * (size_t)N); gs = malloc(16 * (size_t)N); out = malloc(16 * (size_t)N); phi = malloc(16 * (size_t)N);
  mu = malloc(sizeof(double) * (size_t)M); mu_got = malloc(sizeof(double) * (size_t)M);
  if (fread(states, 8, (size_t)N, f) != (size_t)N || fread(psi, 16, (size_t)N, f) != (size_t)N ||
      fread(Hpsi, 16, (size_t)N, f) != (size_t)N || fread(psi0, 16, (size_t)N, f) != (size_t)N ||
      fread(expm, 16, (size_t)N, f) != (size_t)N || fread(gs, 16, (size_t)N, f) != (size_t)N ||
      fread(mu, 8, (size_t)M, f) != (size_t)M) { fprintf(stderr, "short fixture\n"); return 1; }
  fclose(f);

  CHECK(sd_ctx_create(0, &ctx));
  CHECK(sd_xxz_chain(ctx, (int)L, par[0], par[1], par[2], (int)nup, 0, &model));      /* XXZChain, src/SpinModel.jl:63-90 */
  if (sd_model_dim(model) != N || sd_model_L(model) != (int)L || sd_model_nup(model) != (int)nup) { fprintf(stderr, "model dims\n"); return 1; }
  CHECK(sd_model_states(model, 0, N, got_states));                                    /* basis order: bit exact */
  if (memcmp(states, got_states, sizeof(uint64_t) * (size_t)N) != 0) { fprintf(stderr, "basis order differs\n"); return 1; }

  CHECK(sd_apply(ctx, model, SD_C128, out, psi, N));                                  /* apply_H!, src/Hamiltonian.jl:211-273 */
  if (maxdiff(out, Hpsi, 2 * N) > 1e-13) { fprintf(stderr, "sd_apply: %g\n", maxdiff(out, Hpsi, 2 * N)); return 1; }
  if (sd_apply(ctx, model, SD_C128, out, psi, N + 1) != SD_EDIM) { fprintf(stderr, "length check missing\n"); return 1; }
  if (sd_apply(ctx, model, SD_C128, psi, psi, N) != SD_EARG) { fprintf(stderr, "alias check missing\n"); return 1; }

  CHECK(sd_chebyshev_evolve(ctx, model, psi0, N, par[3], 50, par[4], par[5], out));   /* src/TimeEvolution/Chebyshev.jl:61-124 */
  if (maxdiff(out, expm, 2 * N) > 1e-10) { fprintf(stderr, "sd_chebyshev_evolve: %g\n", maxdiff(out, expm, 2 * N)); return 1; }
  CHECK(sd_krylov_evolve(ctx, model, SD_C128, psi0, N, par[3], 30, out));             /* src/TimeEvolution/Krylov.jl:136-192 */
  if (maxdiff(out, expm, 2 * N) > 1e-10) { fprintf(stderr, "sd_krylov_evolve: %g\n", maxdiff(out, expm, 2 * N)); return 1; }

  CHECK(sd_szq(ctx, model, SD_C128, gs, N, 3.14159265358979323846, phi));             /* Sz_q_vector, src/Hamiltonian.jl:307-337 */
  for (i = 0; i < 2 * N; ++i) nrm += phi[i] * phi[i];
  nrm = sqrt(nrm);
  for (i = 0; i < 2 * N; ++i) phi[i] /= nrm;
  CHECK(sd_kpm_moments(ctx, model, phi, N, (int)M, par[6], par[7], mu_got));          /* src/KPM_Sqw.jl:95-128 */
  if (maxdiff(mu_got, mu, M) > 1e-11) { fprintf(stderr, "sd_kpm_moments: %g\n", maxdiff(mu_got, mu, M)); return 1; }

  {   /* the same moments through a caller-supplied operator */
    struct fwd f;
    double *mu_cb = (double *)malloc(sizeof(double) * (size_t)M);
    f.ctx = ctx; f.model = model; f.calls = 0;
    CHECK(sd_ctx_set_apply_callback(ctx, forward_apply, &f));
    CHECK(sd_kpm_moments(ctx, model, phi, N, (int)M, par[6], par[7], mu_cb));
    CHECK(sd_ctx_set_apply_callback(ctx, NULL, NULL));
    if (f.calls < 1 || maxdiff(mu_cb, mu_got, M) > 1e-13) {
      fprintf(stderr, "apply callback: %d calls, moments differ by %g\n", f.calls, maxdiff(mu_cb, mu_got, M));
      return 1;
    }
    free(mu_cb);
  }

  CHECK(sd_ctx_synchronize(ctx));
  sd_model_destroy(model);
  sd_ctx_destroy(ctx);
  printf("cabi_consumer: all checks passed (L=%d N=%lld)\n", (int)L, (long long)N);
  return 0;
}
