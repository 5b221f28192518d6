// BLAS-1 style device kernels used by the on-device recursions (the reference
// uses LinearAlgebra.dot / norm / broadcast on host vectors: src/Lanczos.jl,
// src/TimeEvolution/Krylov.jl, src/TimeEvolution/Chebyshev.jl, src/KPM_Sqw.jl).
// All are single-pass HBM streams with 16-byte accesses; reductions are
// two-stage with a fixed order (deterministic run to run).
#include <hip/hip_runtime.h>

#include "device_common.hpp"

using sd_dev::block_reduce2;

namespace {

constexpr int RED_BLOCKS = 16384;    // most blocks of a streaming pass (= partial pairs of its reduction): long streams run 5-10 % faster with 16384 blocks than with 2048 (profiles/probes/stream_mix_probe.hip)
constexpr int BS = 256;

// conj(x).y for complex (nc=2) or x.y for real (nc=1); n2 = number of double2 elements when vectorised.
// CONJ = false (complex only): the plain product sum x_i*y_i -- LanczosSqw.jl:59 forms dot(conj(psi), H psi)
template <int NC, bool CONJ = true>
__global__ __launch_bounds__(BS) void k_dot(const double *__restrict__ x, const double *__restrict__ y, int64_t N,
                                            double *__restrict__ partials) {
  __shared__ double red[32];
  double a = 0.0, b = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (NC == 2) {
    const double2 *x2 = (const double2 *)x, *y2 = (const double2 *)y;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
      double2 u = x2[i], v = y2[i];
      if (CONJ) { a += u.x * v.x + u.y * v.y; b += u.x * v.y - u.y * v.x; }
      else { a += u.x * v.x - u.y * v.y; b += u.x * v.y + u.y * v.x; }
    }
  } else {
    const int64_t n2 = (((uintptr_t)x | (uintptr_t)y) & 15) ? 0 : N / 2;
    const double2 *x2 = (const double2 *)x, *y2 = (const double2 *)y;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
      double2 u = x2[i], v = y2[i];
      a += u.x * v.x + u.y * v.y;
    }
    for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) a += x[i] * y[i];
  }
  block_reduce2(a, b, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = b; }
}

// s[c] = sum_i V[c*ld + i] * y[i] for c = 0..nc-1 (real vectors, nc <= SD_MDOT_MAXC): the orthogonality checks of
// lanczos_groundstate (src/Lanczos.jl:142-153 computes dot(V[:,k], w/beta) for every k) in one pass over y instead of one
// pass per column.  Per-block partials [block][c], summed in block order by k_mdot_reduce (deterministic).
constexpr int SD_MDOT_MAXC = 16;
__global__ __launch_bounds__(BS) void k_mdot(const double *__restrict__ V, int64_t ld, int nc, const double *__restrict__ y,
                                             int64_t N, double *__restrict__ partials) {
  __shared__ double red[BS / 64][SD_MDOT_MAXC];
  double acc[SD_MDOT_MAXC];
#pragma unroll
  for (int c = 0; c < SD_MDOT_MAXC; ++c) acc[c] = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    const double yi = y[i];
#pragma unroll
    for (int c = 0; c < SD_MDOT_MAXC; ++c)
      if (c < nc) acc[c] += V[(int64_t)c * ld + i] * yi;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < SD_MDOT_MAXC; ++c) {
    double a = acc[c];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if (lane == 0) red[wv][c] = a;
  }
  __syncthreads();
  if (threadIdx.x < SD_MDOT_MAXC) {
    double a = 0.0;
    for (int w = 0; w < BS / 64; ++w) a += red[w][threadIdx.x];
    partials[(size_t)blockIdx.x * SD_MDOT_MAXC + threadIdx.x] = a;
  }
}
__global__ __launch_bounds__(64 * SD_MDOT_MAXC) void k_mdot_reduce(const double *__restrict__ partials, int nb, int nc,
                                                                   double *__restrict__ dst) {
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;     // one wave per column
  double a = 0.0;
  for (int b = lane; b < nb; b += 64) a += partials[(size_t)b * SD_MDOT_MAXC + c];
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  if (lane == 0 && c < nc) dst[c] = a;
}

// Blocked Gram-Schmidt link (lanczos_groundstate, src/Lanczos.jl:116-124, default form):
//   w -= sum_{c < nc1} coef[c] * V1[:,c]      (coefficients on the device: the dots of the previous pass)
//   partial sums of V2[:,c] . w_new  for c < nc2  (the next block's coefficients, or the column of alpha_j)
// One read-modify-write of w per nc1 + nc2 column reads, where the sequential chain (k_sub_dot) passes over w once per column.
// Per element the subtraction runs in column order with the arithmetic of the sequential form (x - c*v).
constexpr int SD_BGS_B = 8;
__global__ __launch_bounds__(BS) void k_proj_dots(double *__restrict__ w, const double *__restrict__ V1, int64_t ld, int nc1,
                                                  const double *__restrict__ coef_dev, const double *__restrict__ V2, int nc2,
                                                  int64_t N, double *__restrict__ partials) {
  __shared__ double red[BS / 64][SD_BGS_B];
  double cf[SD_BGS_B], acc[SD_BGS_B];
#pragma unroll
  for (int c = 0; c < SD_BGS_B; ++c) { cf[c] = c < nc1 ? coef_dev[c] : 0.0; acc[c] = 0.0; }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool vec = (ld % 2 == 0) && !(((uintptr_t)w | (uintptr_t)V1 | (uintptr_t)V2) & 15);
  const int64_t n2 = vec ? N / 2 : 0;
  double2 *w2 = (double2 *)w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 x = w2[i];
#pragma unroll
    for (int c = 0; c < SD_BGS_B; ++c)
      if (c < nc1) { const double2 v = ((const double2 *)(V1 + (int64_t)c * ld))[i]; x.x = x.x - cf[c] * v.x; x.y = x.y - cf[c] * v.y; }
    if (nc1 > 0) w2[i] = x;
#pragma unroll
    for (int c = 0; c < SD_BGS_B; ++c)
      if (c < nc2) { const double2 v = ((const double2 *)(V2 + (int64_t)c * ld))[i]; acc[c] += v.x * x.x + v.y * x.y; }
  }
  for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    double x = w[i];
#pragma unroll
    for (int c = 0; c < SD_BGS_B; ++c)
      if (c < nc1) x = x - cf[c] * V1[(int64_t)c * ld + i];
    if (nc1 > 0) w[i] = x;
#pragma unroll
    for (int c = 0; c < SD_BGS_B; ++c)
      if (c < nc2) acc[c] += V2[(int64_t)c * ld + i] * x;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < SD_BGS_B; ++c) {
    double a = acc[c];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if (lane == 0) red[wv][c] = a;
  }
  __syncthreads();
  if (threadIdx.x < SD_BGS_B) {
    double a = 0.0;
    for (int k = 0; k < BS / 64; ++k) a += red[k][threadIdx.x];
    partials[(size_t)blockIdx.x * SD_BGS_B + threadIdx.x] = a;
  }
}
__global__ __launch_bounds__(64 * SD_BGS_B) void k_bgs_reduce(const double *__restrict__ partials, int nb, int nc,
                                                              double *__restrict__ dst) {
  const int c = threadIdx.x >> 6, lane = threadIdx.x & 63;     // one wave per column, fixed order
  double a = 0.0;
  for (int b = lane; b < nb; b += 64) a += partials[(size_t)b * SD_BGS_B + c];
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  if (lane == 0 && c < nc) dst[c] = a;
}

// ---- lanczos_groundstate without reduction launches and with the orthogonality check riding along (round 4) ----
// The step of src/Lanczos.jl:111-156 as a chain of passes that hand their per-block partial sums straight to the next pass:
// every block of the consumer sums the producer's list itself (column c by one wave, lane-strided then shuffled: one fixed
// order on every block), so a pass is ONE launch and nothing but the check results crosses to the host.
//   k_gs_pass: the blocked Gram-Schmidt link of k_proj_dots (same per-element arithmetic) with
//     coef[c] = sum of the previous pass's partials, and -- CHK -- the dots of the same V2 columns with a second vector y: the
//     orthogonality check of the PREVIOUS step (src/Lanczos.jl:142-153 wants dot(V[:,k], w/beta) for every k <= j; y = V[:,j+1]
//     is that very vector), which so costs one more read of y per pass instead of a sweep over all columns of its own.
__device__ __forceinline__ void gs_sum_cols(const double *__restrict__ part, int nb, int nc, double *out /* LDS, SD_BGS_B */) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c = wv; c < SD_BGS_B; c += BS / 64) {
    double a = 0.0;
    if (c < nc) for (int b = lane; b < nb; b += 64) a += part[(size_t)b * SD_BGS_B + c];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if (lane == 0) out[c] = a;
  }
  __syncthreads();
}
template <bool CHK>
__global__ __launch_bounds__(BS) void k_gs_pass(double *__restrict__ w, const double *__restrict__ V1, int64_t ld, int nc1,
                                                const double *__restrict__ part_in, int nb_in, const double *__restrict__ V2, int nc2,
                                                const double *__restrict__ y, int64_t N, double *__restrict__ part_out,
                                                double *__restrict__ chk_out) {
  __shared__ double red[BS / 64][2 * SD_BGS_B];
  __shared__ double cfs[SD_BGS_B];
  if (nc1 > 0) gs_sum_cols(part_in, nb_in, nc1, cfs);
  double cf[SD_BGS_B], acc[SD_BGS_B], ac2[SD_BGS_B];
#pragma unroll
  for (int c = 0; c < SD_BGS_B; ++c) { cf[c] = c < nc1 ? cfs[c] : 0.0; acc[c] = 0.0; ac2[c] = 0.0; }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool vec = (ld % 2 == 0) && !(((uintptr_t)w | (uintptr_t)V1 | (uintptr_t)V2 | (uintptr_t)(CHK ? y : w)) & 15);
  const int64_t n2 = vec ? N / 2 : 0;
  double2 *w2 = (double2 *)w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 x = w2[i];
#pragma unroll
    for (int c = 0; c < SD_BGS_B; ++c)
      if (c < nc1) { const double2 v = ((const double2 *)(V1 + (int64_t)c * ld))[i]; x.x = x.x - cf[c] * v.x; x.y = x.y - cf[c] * v.y; }
    if (nc1 > 0) w2[i] = x;
    double2 yy = make_double2(0.0, 0.0);
    if (CHK) yy = ((const double2 *)y)[i];
#pragma unroll
    for (int c = 0; c < SD_BGS_B; ++c)
      if (c < nc2) {
        const double2 v = ((const double2 *)(V2 + (int64_t)c * ld))[i];
        acc[c] += v.x * x.x + v.y * x.y;
        if (CHK) ac2[c] += v.x * yy.x + v.y * yy.y;
      }
  }
  for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    double x = w[i];
#pragma unroll
    for (int c = 0; c < SD_BGS_B; ++c)
      if (c < nc1) x = x - cf[c] * V1[(int64_t)c * ld + i];
    if (nc1 > 0) w[i] = x;
    const double yi = CHK ? y[i] : 0.0;
#pragma unroll
    for (int c = 0; c < SD_BGS_B; ++c)
      if (c < nc2) { const double v = V2[(int64_t)c * ld + i]; acc[c] += v * x; if (CHK) ac2[c] += v * yi; }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < SD_BGS_B; ++c) {
    double a = acc[c], b = ac2[c];
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); if (CHK) b += __shfl_down(b, off, 64); }
    if (lane == 0) { red[wv][c] = a; red[wv][SD_BGS_B + c] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * SD_BGS_B) {
    double a = 0.0;
    for (int k = 0; k < BS / 64; ++k) a += red[k][threadIdx.x];
    if (threadIdx.x < SD_BGS_B) part_out[(size_t)blockIdx.x * SD_BGS_B + threadIdx.x] = a;
    else if (CHK) chk_out[(size_t)blockIdx.x * SD_BGS_B + (threadIdx.x - SD_BGS_B)] = a;
  }
}
//   k_gs_update: alpha_j = sum of the last pass's partials (column 0), beta_{j-1} = sqrt(sum of the previous step's |w|^2 partials);
//     w = (w - alpha v_j) - beta_{j-1} v_{j-1}  (src/Lanczos.jl:127-129).  The full re-orthogonalisation has already removed the
//     v_{j-1} component, so this puts -beta_{j-1} v_{j-1} back in, and the reference's check loop (:142-153) finds it at k = j-1 on
//     EVERY step and repairs it.  That repair is part of the normal flow: the pass also forms the partials of
//     d = dot(v_{j-1}, w) (column 1) for k_gs_correct.  Column 0 of the output: |w|^2 partials (used when there is no v_{j-1}).
__global__ __launch_bounds__(BS) void k_gs_update(double *__restrict__ w, const double *__restrict__ vj, const double *__restrict__ vjm1,
                                                  int64_t N, const double *__restrict__ alpha_part, int nb_a,
                                                  const double *__restrict__ n2_prev, int nb_n, double *__restrict__ store_alpha,
                                                  double *__restrict__ store_beta, double *__restrict__ out_part) {
  __shared__ double red[32];
  __shared__ double sc[SD_BGS_B], sn[SD_BGS_B];
  gs_sum_cols(alpha_part, nb_a, 1, sc);
  if (vjm1) gs_sum_cols(n2_prev, nb_n, 1, sn);
  const double a = sc[0], b = vjm1 ? sqrt(sn[0]) : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { *store_alpha = a; if (vjm1) *store_beta = b; }
  double s = 0.0, d = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool vec = !(((uintptr_t)w | (uintptr_t)vj | (uintptr_t)(vjm1 ? vjm1 : vj)) & 15);
  const int64_t n2 = vec ? N / 2 : 0;
  double2 *w2 = (double2 *)w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 x = w2[i];
    const double2 v = ((const double2 *)vj)[i];
    x.x = x.x - a * v.x; x.y = x.y - a * v.y;
    if (vjm1) {
      const double2 u = ((const double2 *)vjm1)[i];
      x.x = x.x - b * u.x; x.y = x.y - b * u.y;
      d += u.x * x.x + u.y * x.y;
    }
    w2[i] = x;
    s += x.x * x.x + x.y * x.y;
  }
  for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    double x = w[i] - a * vj[i];
    if (vjm1) { x = x - b * vjm1[i]; d += vjm1[i] * x; }
    w[i] = x;
    s += x * x;
  }
  __syncthreads();
  block_reduce2(s, d, red);
  if (threadIdx.x == 0) { out_part[(size_t)blockIdx.x * SD_BGS_B] = s; out_part[(size_t)blockIdx.x * SD_BGS_B + 1] = d; }
}
//   k_gs_correct: d = sum of column 1 of the update's partials; w -= d v_{j-1}  (the reference's correction at k = j-1, :147);
//     |w|^2 partials out (column 0): beta_j = norm(w) after the correction (:148).
__global__ __launch_bounds__(BS) void k_gs_correct(double *__restrict__ w, const double *__restrict__ vjm1, int64_t N,
                                                   const double *__restrict__ upd_part, int nb_u, double *__restrict__ n2_out) {
  __shared__ double red[32];
  __shared__ double sc[SD_BGS_B];
  gs_sum_cols(upd_part + 1, nb_u, 1, sc);                   // column 1
  const double d = sc[0];
  double s = 0.0, s1 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool vec = !(((uintptr_t)w | (uintptr_t)vjm1) & 15);
  const int64_t n2 = vec ? N / 2 : 0;
  double2 *w2 = (double2 *)w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 x = w2[i];
    const double2 u = ((const double2 *)vjm1)[i];
    x.x = x.x - d * u.x; x.y = x.y - d * u.y;
    w2[i] = x;
    s += x.x * x.x + x.y * x.y;
  }
  for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    const double x = w[i] - d * vjm1[i];
    w[i] = x;
    s += x * x;
  }
  __syncthreads();
  block_reduce2(s, s1, red);
  if (threadIdx.x == 0) n2_out[(size_t)blockIdx.x * SD_BGS_B] = s;
}
//   k_gs_scale: v_{j+1} = w / beta_j with beta_j = sqrt(sum of |w|^2 partials)  (src/Lanczos.jl:155); one thread files beta_j.
__global__ __launch_bounds__(BS) void k_gs_scale(double *__restrict__ vnext, const double *__restrict__ w, int64_t N,
                                                 const double *__restrict__ n2_part, int nb_n, double *__restrict__ store_beta) {
  __shared__ double sn[SD_BGS_B];
  gs_sum_cols(n2_part, nb_n, 1, sn);
  const double b = sqrt(sn[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) *store_beta = b;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool vec = !(((uintptr_t)w | (uintptr_t)vnext) & 15);
  const int64_t n2 = vec ? N / 2 : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    const double2 x = ((const double2 *)w)[i];
    ((double2 *)vnext)[i] = make_double2(x.x / b, x.y / b);
  }
  for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) vnext[i] = w[i] / b;
}
//   k_gs_chk_reduce: the check dots of all passes of a chain (pass p, column c -> chk[8 p + c]); block p reduces pass p.
__global__ __launch_bounds__(BS) void k_gs_chk_reduce(const double *__restrict__ chk_part, int nb, double *__restrict__ chk) {
  __shared__ double out[SD_BGS_B];
  gs_sum_cols(chk_part + (size_t)blockIdx.x * nb * SD_BGS_B, nb, SD_BGS_B, out);
  if (threadIdx.x < SD_BGS_B) chk[(size_t)blockIdx.x * SD_BGS_B + threadIdx.x] = out[threadIdx.x];
}

// One link of the modified Gram-Schmidt chain of lanczos_groundstate (src/Lanczos.jl:116-124) without a host round trip:
//   if (v_sub) w -= (*s_dev) * v_sub;      then   partial sums of  v_dot . w   (reduced into a device scalar by k_reduce_to)
// The next link reads that scalar on the device.  Same element order per thread as k_dot<1> / k_ew2 when the vectors are
// 16-byte aligned; arithmetic w - s*v as in the un-fused pass.
__global__ __launch_bounds__(BS) void k_sub_dot(double *__restrict__ w, const double *__restrict__ v_sub,
                                                const double *__restrict__ s_dev, const double *__restrict__ v_dot, int64_t N,
                                                double *__restrict__ partials) {
  __shared__ double red[32];
  const double sv = v_sub ? *s_dev : 0.0;
  double a = 0.0, b = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool vec = !(((uintptr_t)w | (uintptr_t)v_dot | (uintptr_t)(v_sub ? v_sub : v_dot)) & 15);
  const int64_t n2 = vec ? N / 2 : 0;
  double2 *w2 = (double2 *)w;
  const double2 *s2 = (const double2 *)v_sub, *d2 = (const double2 *)v_dot;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    double2 x = w2[i];
    if (v_sub) { const double2 u = s2[i]; x.x = x.x - sv * u.x; x.y = x.y - sv * u.y; w2[i] = x; }
    const double2 v = d2[i];
    a += v.x * x.x + v.y * x.y;
  }
  for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    double x = w[i];
    if (v_sub) { x = x - sv * v_sub[i]; w[i] = x; }
    a += v_dot[i] * x;
  }
  block_reduce2(a, b, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = b; }
}

__global__ __launch_bounds__(BS) void k_nrm2sq(const double *__restrict__ x, int64_t n, double *__restrict__ partials) {
  __shared__ double red[32];
  double a = 0.0, b = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n2 = ((uintptr_t)x & 15) ? 0 : n / 2;
  const double2 *x2 = (const double2 *)x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) { double2 v = x2[i]; a += v.x * v.x + v.y * v.y; }
  for (int64_t i = 2 * n2 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { double v = x[i]; a += v * v; }
  block_reduce2(a, b, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = b; }
}

// number of ComplexF64 elements with a non-zero imaginary part (as a double; exact below 2^53) -> partials
__global__ __launch_bounds__(BS) void k_imag_count(const double2 *__restrict__ x, int64_t n, double *__restrict__ partials) {
  __shared__ double red[32];
  double a = 0.0, b = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a += (x[i].y != 0.0) ? 1.0 : 0.0;
  block_reduce2(a, b, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = b; }
}

__global__ __launch_bounds__(1024) void k_reduce_to(const double *__restrict__ partials, int n, double *__restrict__ dst) {
  __shared__ double red[32];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { a += partials[2 * i]; b += partials[2 * i + 1]; }
  block_reduce2(a, b, red);
  if (threadIdx.x == 0) { dst[0] = a; dst[1] = b; }
}

enum { OP_SCALE_DIV, OP_NEG, OP_SUB2, OP_SUB2_1 };

template <int OP>
__device__ __forceinline__ double ew_op(double w, double v, double u, double a, double b) {
  if (OP == OP_SCALE_DIV) return v / a;
  if (OP == OP_NEG) return -w;
  if (OP == OP_SUB2) return (w - a * v) - b * u;
  return w - a * v;  // OP_SUB2_1
}
template <int OP> struct ew_reads { static constexpr bool w = OP != OP_SCALE_DIV, v = OP != OP_NEG, u = OP == OP_SUB2; };

// elementwise pass with 16-byte accesses (n2 = number of double2 elements); NORM accumulates |w_new|^2
template <int OP, bool NORM>
__global__ __launch_bounds__(BS) void k_ew2(double2 *__restrict__ w, const double2 *__restrict__ v,
                                            const double2 *__restrict__ u, int64_t n2, double a, double b,
                                            double *__restrict__ partials) {
  __shared__ double red[32];
  double s = 0.0, s1 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
    const double2 z = make_double2(0.0, 0.0);
    const double2 ww = ew_reads<OP>::w ? w[i] : z, vv = ew_reads<OP>::v ? v[i] : z, uu = ew_reads<OP>::u ? u[i] : z;
    const double2 r = make_double2(ew_op<OP>(ww.x, vv.x, uu.x, a, b), ew_op<OP>(ww.y, vv.y, uu.y, a, b));
    w[i] = r;
    if (NORM) s += r.x * r.x + r.y * r.y;
  }
  if (NORM) {
    block_reduce2(s, s1, red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s; partials[2 * blockIdx.x + 1] = 0.0; }
  }
}

// scalar fallback (odd length or 8-byte aligned columns)
template <int OP, bool NORM>
__global__ __launch_bounds__(BS) void k_ew(double *__restrict__ w, const double *__restrict__ v,
                                           const double *__restrict__ u, int64_t n, double a, double b,
                                           double *__restrict__ partials) {
  __shared__ double red[32];
  double s = 0.0, s1 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double r = ew_op<OP>(ew_reads<OP>::w ? w[i] : 0.0, ew_reads<OP>::v ? v[i] : 0.0, ew_reads<OP>::u ? u[i] : 0.0, a, b);
    w[i] = r;
    if (NORM) s += r * r;
  }
  if (NORM) {
    block_reduce2(s, s1, red);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s; partials[2 * blockIdx.x + 1] = 0.0; }
  }
}

// ---- Lanczos / Krylov three-term update on UN-NORMALISED vectors (one pass instead of update + normalise) ----
// The recursions keep u_j = beta_{j-1} v_j (the un-normalised w of the step before; |u_j|^2 = n2c on the device) instead of
// v_j, so the pass  v_{j+1} = w / beta_j  (a read and a write of the whole vector) disappears.  With t = H u_cur this kernel forms
//   alpha = <u_cur, t> / |u_cur|^2          (= <v, H v>, the Rayleigh quotient)
//   w     = H v - alpha v - beta v_prev     with  H v = t / b_c,  v = u_cur / b_c,  v_prev = u_prev / b_p   per element
// (b_c = sqrt(n2c), b_p = sqrt(n2p); true divisions, so v and v_prev are the very numbers the normalising pass would have
// stored), writes w over t and accumulates |w|^2.  form 0: w - (alpha v + beta v_prev)  (src/Lanczos.jl:58-62);
// form 1: (w - alpha v) - beta v_prev  (src/Lanczos.jl:222-224); form 2: the same with complex alpha
// (src/TimeEvolution/Krylov.jl:156-159).  One thread stores alpha (1 or 2 doubles) and b_c.
typedef double sd_d2nt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld_nt(const double2 *p) {
  const sd_d2nt w = __builtin_nontemporal_load(reinterpret_cast<const sd_d2nt *>(p));
  return make_double2(w.x, w.y);
}
__device__ __forceinline__ void st_nt(double2 *p, double2 v) {
  sd_d2nt w; w.x = v.x; w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<sd_d2nt *>(p));
}
struct FoldScalars { double ar, ai, bc, bp; };
__device__ __forceinline__ FoldScalars fold_resolve(const double *dot, const double *n2c, const double *n2p, int form,
                                                    double *store_alpha, double *store_bc, bool writer) {
  FoldScalars f;
  const double nc = n2c ? *n2c : 1.0;
  f.bc = n2c ? sqrt(nc) : 1.0;
  f.bp = n2p ? sqrt(*n2p) : 1.0;
  f.ar = dot[0] / nc;
  f.ai = form == 2 ? dot[1] / nc : 0.0;
  if (writer) {
    if (store_alpha) { store_alpha[0] = f.ar; if (form == 2) store_alpha[1] = f.ai; }
    if (store_bc && n2c) *store_bc = f.bc;
  }
  return f;
}
// One element of the three-term update (shared by the streaming kernel below and k_lanczos_fold_p): same operations, same order.
template <int FORM, bool HAVE_U>
__device__ __forceinline__ double2 fold_elem(const double2 tt, const double2 cc, const double2 pp, const FoldScalars &f) {
  const double wx = tt.x / f.bc, wy = tt.y / f.bc, vx = cc.x / f.bc, vy = cc.y / f.bc;
  double ux = 0.0, uy = 0.0;
  if (HAVE_U) { ux = pp.x / f.bp; uy = pp.y / f.bp; }
  double2 r;
  if (FORM == 0) {
    r.x = HAVE_U ? wx - (f.ar * vx + f.bc * ux) : wx - f.ar * vx;
    r.y = HAVE_U ? wy - (f.ar * vy + f.bc * uy) : wy - f.ar * vy;
  } else if (FORM == 1) {
    r.x = wx - f.ar * vx; r.y = wy - f.ar * vy;
    if (HAVE_U) { r.x -= f.bc * ux; r.y -= f.bc * uy; }
  } else {
    r.x = wx - (f.ar * vx - f.ai * vy); r.y = wy - (f.ar * vy + f.ai * vx);
    if (HAVE_U) { r.x -= f.bc * ux; r.y -= f.bc * uy; }
  }
  return r;
}
// Streams: t, u_cur and u_prev are read once and t written once per pass (64 B per element: at L=32 38.5 GB), nothing is reused
// before the whole vector has gone by -- non-temporal loads and stores, and two 16-byte elements per array in flight per lane
// (the one-element loop ran at 4.9 TB/s of the 6.29 TB/s a float4 copy reaches).
template <int FORM, bool HAVE_U>
__global__ __launch_bounds__(BS) void k_lanczos_fold(double2 *__restrict__ t, const double2 *__restrict__ uc,
                                                     const double2 *__restrict__ up, int64_t N, const double *__restrict__ dot,
                                                     const double *__restrict__ n2c, const double *__restrict__ n2p,
                                                     double *__restrict__ store_alpha, double *__restrict__ store_bc,
                                                     double *__restrict__ partials) {
  __shared__ double red[32];
  const FoldScalars f = fold_resolve(dot, n2c, n2p, FORM, store_alpha, store_bc, blockIdx.x == 0 && threadIdx.x == 0);
  double s = 0.0, s1 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const double2 z = make_double2(0.0, 0.0);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < N; i += 2 * stride) {
    const double2 t0 = ld_nt(t + i), t1 = ld_nt(t + i + stride);
    const double2 c0 = ld_nt(uc + i), c1 = ld_nt(uc + i + stride);
    const double2 p0 = HAVE_U ? ld_nt(up + i) : z, p1 = HAVE_U ? ld_nt(up + i + stride) : z;
    const double2 r0 = fold_elem<FORM, HAVE_U>(t0, c0, p0, f), r1 = fold_elem<FORM, HAVE_U>(t1, c1, p1, f);
    st_nt(t + i, r0); st_nt(t + i + stride, r1);
    s += r0.x * r0.x + r0.y * r0.y;
    s += r1.x * r1.x + r1.y * r1.y;
  }
  if (i < N) {
    const double2 r0 = fold_elem<FORM, HAVE_U>(ld_nt(t + i), ld_nt(uc + i), HAVE_U ? ld_nt(up + i) : z, f);
    st_nt(t + i, r0);
    s += r0.x * r0.x + r0.y * r0.y;
  }
  block_reduce2(s, s1, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s; partials[2 * blockIdx.x + 1] = 0.0; }
}
// ---- the same update for launch-bound sizes: the reductions' second stages folded into the consuming pass, batched ----
// A Lanczos step of a small system is four launches of a few microseconds of host time each (apply + its reduction, update +
// its reduction): 19-22 us per step whatever the dimension.  Here every block of the update sums the producer's partial lists
// itself -- the apply's per-tile pairs of <u|t>, the previous updates' per-block |w|^2 -- in one fixed order (so all blocks of a
// vector agree to the bit), and writes its own |w|^2 partial for the next step: two launches per step.  grid.y = vectors of a
// batch stored `bstride` elements apart (the momenta of lanczos_sqw, src/LanczosSqw.jl:65), each with its own lists and scalars.
__device__ __forceinline__ void sum_list(const double *__restrict__ p, int n, double *red, double *bc /* 2 doubles LDS */) {
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { a += p[2 * i]; b += p[2 * i + 1]; }
  __syncthreads();
  block_reduce2(a, b, red);
  if (threadIdx.x == 0) { bc[0] = a; bc[1] = b; }
  __syncthreads();
}
// the three lists of a fold pass in ONE sweep: <u|t> pairs, |u_cur|^2 and |u_prev|^2 (first components; the seconds are zero) --
// per list the same thread-local order and the same tree as sum_list, i.e. the same bits, with three barriers instead of nine
__device__ __forceinline__ void sum_lists3(const double *__restrict__ pd, int nd, const double *__restrict__ pc,
                                           const double *__restrict__ pp, int nn, double *red /* 64 doubles LDS */,
                                           double *sc /* 6 doubles LDS */) {
  double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
  for (int i = threadIdx.x; i < nd; i += blockDim.x) { a += pd[2 * i]; b += pd[2 * i + 1]; }
  if (pc) for (int i = threadIdx.x; i < nn; i += blockDim.x) c += pc[2 * i];
  if (pp) for (int i = threadIdx.x; i < nn; i += blockDim.x) d += pp[2 * i];
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64);
    c += __shfl_down(c, off, 64); d += __shfl_down(d, off, 64);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) { red[4 * wv] = a; red[4 * wv + 1] = b; red[4 * wv + 2] = c; red[4 * wv + 3] = d; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double x = 0.0, y = 0.0, z = 0.0, w = 0.0;
    for (int k = 0; k < nw; ++k) { x += red[4 * k]; y += red[4 * k + 1]; z += red[4 * k + 2]; w += red[4 * k + 3]; }
    sc[0] = x; sc[1] = y; sc[2] = z; sc[3] = 0.0; sc[4] = w; sc[5] = 0.0;
  }
  __syncthreads();
}
template <int FORM, bool HAVE_U>
__global__ __launch_bounds__(BS) void k_lanczos_fold_p(double2 *__restrict__ t, const double2 *__restrict__ uc,
                                                       const double2 *__restrict__ up, int64_t N, int64_t bstride,
                                                       const double *__restrict__ dotp, int ndot,
                                                       const double *__restrict__ n2cp, const double *__restrict__ n2pp, int nn2,
                                                       double *__restrict__ store_alpha, double *__restrict__ store_bc,
                                                       int64_t store_stride, double *__restrict__ n2out) {
  __shared__ double red[64];
  __shared__ double sc[6];
  const int q = blockIdx.y;
  t += (int64_t)q * bstride; uc += (int64_t)q * bstride;
  if (HAVE_U) up += (int64_t)q * bstride;
  sum_lists3(dotp + 2 * (size_t)ndot * q, ndot, n2cp ? n2cp + 2 * (size_t)nn2 * q : nullptr, n2pp ? n2pp + 2 * (size_t)nn2 * q : nullptr,
             nn2, red, sc);
  const FoldScalars f = fold_resolve(sc, n2cp ? sc + 2 : nullptr, n2pp ? sc + 4 : nullptr, FORM,
                                     store_alpha ? store_alpha + (int64_t)q * store_stride : nullptr,
                                     store_bc ? store_bc + (int64_t)q * store_stride : nullptr, blockIdx.x == 0 && threadIdx.x == 0);
  double s = 0.0, s1 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    const double2 r = fold_elem<FORM, HAVE_U>(t[i], uc[i], HAVE_U ? up[i] : make_double2(0.0, 0.0), f);
    t[i] = r;
    s += r.x * r.x + r.y * r.y;
  }
  __syncthreads();
  block_reduce2(s, s1, red);
  if (threadIdx.x == 0) { n2out[2 * ((size_t)gridDim.x * q + blockIdx.x)] = s; n2out[2 * ((size_t)gridDim.x * q + blockIdx.x) + 1] = 0.0; }
}
// the scalars of the last step (alpha_m and beta_{m-1}) from the partial lists; one block per vector
__global__ __launch_bounds__(BS) void k_lanczos_fold_scalars_p(const double *__restrict__ dotp, int ndot, const double *__restrict__ n2cp,
                                                               int nn2, int form, double *store_alpha, double *store_bc,
                                                               int64_t store_stride) {
  __shared__ double red[32];
  __shared__ double sc[4];
  const int q = blockIdx.x;
  sum_list(dotp + 2 * (size_t)ndot * q, ndot, red, sc);
  if (n2cp) sum_list(n2cp + 2 * (size_t)nn2 * q, nn2, red, sc + 2);
  (void)fold_resolve(sc, n2cp ? sc + 2 : nullptr, nullptr, form, store_alpha + (int64_t)q * store_stride,
                     store_bc ? store_bc + (int64_t)q * store_stride : nullptr, threadIdx.x == 0);
}

// the scalars of the last step alone (the recursion ends on alpha_m: no vector update behind it)
__global__ void k_lanczos_fold_scalars(const double *dot, const double *n2c, int form, double *store_alpha, double *store_bc) {
  (void)fold_resolve(dot, n2c, nullptr, form, store_alpha, store_bc, true);
}

// y (+)= sum_k (cr[k] + i ci[k]) * X_k, columns given by pointer, accumulated in column order with exactly the arithmetic
// of one complex axpy pass per column, y += (cr + i ci) x evaluated component-wise (psi_t .+= y[k] * V[k], src/TimeEvolution/Krylov.jl:186-188): same bits, one read of
// every column and one write of y instead of a read-modify-write of y per column.
constexpr int SD_CCOMB_MAXC = 16;
struct CCombArgs { const double2 *x[SD_CCOMB_MAXC]; double cr[SD_CCOMB_MAXC], ci[SD_CCOMB_MAXC]; };
__global__ __launch_bounds__(BS) void k_ccombine(double2 *__restrict__ y, int64_t N, int nc, CCombArgs a, int init) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    double2 acc = init ? make_double2(0.0, 0.0) : y[i];
    for (int k = 0; k < nc; ++k) {
      const double2 x = a.x[k][i];
      const double tr = a.cr[k] * x.x - a.ci[k] * x.y, ti = a.cr[k] * x.y + a.ci[k] * x.x;
      acc.x += tr; acc.y += ti;
    }
    y[i] = acc;
  }
}

__global__ __launch_bounds__(BS) void k_cheb_init(double2 *__restrict__ y, const double2 *__restrict__ x0,
                                                  const double2 *__restrict__ x1, int64_t N, double c0r, double c0i,
                                                  double c1r, double c1i, int have1) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    // psi_t = 0; psi_t += c[1]*phi_prev; psi_t += c[2]*phi_curr   (src/TimeEvolution/Chebyshev.jl:96-102)
    double tr = 0.0, ti = 0.0;
    double2 a = x0[i];
    tr += c0r * a.x - c0i * a.y; ti += c0r * a.y + c0i * a.x;
    if (have1) { double2 b = x1[i]; tr += c1r * b.x - c1i * b.y; ti += c1r * b.y + c1i * b.x; }
    y[i] = make_double2(tr, ti);
  }
}

__global__ __launch_bounds__(BS) void k_promote(double2 *__restrict__ y, const double *__restrict__ x, int nc, int64_t N) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride)
    y[i] = nc == 2 ? make_double2(x[2 * i], x[2 * i + 1]) : make_double2(x[i], 0.0);
}

#define SD_GEMV_MAXC 256
struct GemvCoef { double c[SD_GEMV_MAXC]; };
__global__ __launch_bounds__(BS) void k_gemv_cols(double *__restrict__ y, const double *__restrict__ V, int64_t N,
                                                  int k0, int ncols, GemvCoef coef, int first) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    double s = first ? 0.0 : y[i];
    for (int k = 0; k < ncols; ++k) s += V[i + N * (int64_t)(k0 + k)] * coef.c[k];
    y[i] = s;
  }
}

__global__ __launch_bounds__(BS) void k_fill_randn(double *__restrict__ x, int64_t n, uint64_t seed, uint64_t first) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = sd_dev::sd_randn_at(seed, first + (uint64_t)i);
}

inline unsigned grid_for(int64_t n) {
  int64_t nb = (n + BS - 1) / BS;
  if (nb > 16384) nb = 16384;
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}

}  // namespace

double sd_randn_host(uint64_t seed, uint64_t k) { return sd_dev::sd_randn_at(seed, k); }

int sd_k_dot(sd_ctx *ctx, int nc, const double *x, const double *y, int64_t N, int slot) {
  int rc = sd_ensure_partials(ctx, 2 * RED_BLOCKS); if (rc) return rc;
  int nb = (int)std::min<int64_t>(RED_BLOCKS, std::max<int64_t>(1, (N + BS - 1) / BS));
  if (nc == 2) hipLaunchKernelGGL(k_dot<2>, dim3(nb), dim3(BS), 0, ctx->stream, x, y, N, ctx->d_partials);
  else hipLaunchKernelGGL(k_dot<1>, dim3(nb), dim3(BS), 0, ctx->stream, x, y, N, ctx->d_partials);
  hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_partials, nb, ctx->d_scalars + slot);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

// the same reductions filed at any device address (batched recursions keep one scalar pair per vector)
int sd_k_dot_to(sd_ctx *ctx, int nc, const double *x, const double *y, int64_t N, double *dst_dev) {
  int rc = sd_ensure_partials(ctx, 2 * RED_BLOCKS); if (rc) return rc;
  int nb = (int)std::min<int64_t>(RED_BLOCKS, std::max<int64_t>(1, (N + BS - 1) / BS));
  if (nc == 2) hipLaunchKernelGGL(k_dot<2>, dim3(nb), dim3(BS), 0, ctx->stream, x, y, N, ctx->d_partials);
  else hipLaunchKernelGGL(k_dot<1>, dim3(nb), dim3(BS), 0, ctx->stream, x, y, N, ctx->d_partials);
  hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_partials, nb, dst_dev);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
int sd_k_nrm2sq_to(sd_ctx *ctx, const double *x, int64_t n, double *dst_dev) {
  int rc = sd_ensure_partials(ctx, 2 * RED_BLOCKS); if (rc) return rc;
  int nb = (int)std::min<int64_t>(RED_BLOCKS, std::max<int64_t>(1, (n + BS - 1) / BS));
  hipLaunchKernelGGL(k_nrm2sq, dim3(nb), dim3(BS), 0, ctx->stream, x, n, ctx->d_partials);
  hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_partials, nb, dst_dev);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

int sd_k_dotu(sd_ctx *ctx, const double *x, const double *y, int64_t N, int slot) {   // sum x_i*y_i (complex, no conjugation)
  int rc = sd_ensure_partials(ctx, 2 * RED_BLOCKS); if (rc) return rc;
  int nb = (int)std::min<int64_t>(RED_BLOCKS, std::max<int64_t>(1, (N + BS - 1) / BS));
  hipLaunchKernelGGL((k_dot<2, false>), dim3(nb), dim3(BS), 0, ctx->stream, x, y, N, ctx->d_partials);
  hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_partials, nb, ctx->d_scalars + slot);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

int sd_k_mgs_chain(sd_ctx *ctx, double *w, const double *V, int64_t ld, int ncols, int64_t N, int slot) {
  // for k = 0..ncols-2: w -= dot(V[:,k], w) * V[:,k];  then dot(V[:,ncols-1], w) -> d_scalars[slot] (read it with sd_read_scalars).
  // Every dot stays on the device (slot is re-used link after link); no host synchronisation inside the chain.
  if (ncols <= 0) return SD_OK;
  int rc = sd_ensure_partials(ctx, 2 * RED_BLOCKS); if (rc) return rc;
  const int nb = (int)std::min<int64_t>(RED_BLOCKS, std::max<int64_t>(1, (N / 2 + BS - 1) / BS));
  for (int k = 0; k < ncols; ++k) {
    hipLaunchKernelGGL(k_sub_dot, dim3(nb), dim3(BS), 0, ctx->stream, w, k > 0 ? V + (int64_t)(k - 1) * ld : nullptr,
                       ctx->d_scalars + slot, V + (int64_t)k * ld, N, ctx->d_partials);
    hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_partials, nb, ctx->d_scalars + slot);
  }
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

int sd_k_bgs_chain(sd_ctx *ctx, double *w, const double *V, int64_t ld, int ncols, int64_t N, int slot) {
  // Blocked form of sd_k_mgs_chain: the columns 0..ncols-2 are projected out of w in blocks of SD_BGS_B -- the coefficients of
  // a block are its dots with w as it stands when the block begins (classical Gram-Schmidt inside a block, modified between
  // blocks) -- then dot(V[:,ncols-1], w) -> d_scalars[slot].  Every pass updates w with one block and forms the dots of the
  // next: about 2 column reads per column and 2/SD_BGS_B passes over w, against 4 vector streams per column of the
  // sequential chain.  No host synchronisation inside.
  if (ncols <= 0) return SD_OK;
  const int nb = (int)std::min<int64_t>(RED_BLOCKS, std::max<int64_t>(1, (N / 2 + BS - 1) / BS));
  int rc = sd_ensure_partials(ctx, (size_t)nb * SD_BGS_B + 4 * SD_BGS_B); if (rc) return rc;
  double *coef[2] = {ctx->d_partials + (size_t)nb * SD_BGS_B, ctx->d_partials + (size_t)nb * SD_BGS_B + 2 * SD_BGS_B};
  const int nproj = ncols - 1;
  const double *vlast = V + (int64_t)(ncols - 1) * ld;
  int cur = 0;
  // first pass: dots only (nothing to subtract yet)
  {
    const int nc2 = nproj > 0 ? std::min(SD_BGS_B, nproj) : 1;
    const double *V2 = nproj > 0 ? V : vlast;
    double *dst = nproj > 0 ? coef[cur] : ctx->d_scalars + slot;
    hipLaunchKernelGGL(k_proj_dots, dim3(nb), dim3(BS), 0, ctx->stream, w, V, ld, 0, coef[cur], V2, nc2, N, ctx->d_partials);
    hipLaunchKernelGGL(k_bgs_reduce, dim3(1), dim3(64 * SD_BGS_B), 0, ctx->stream, ctx->d_partials, nb, nc2, dst);
  }
  for (int c0 = 0; c0 < nproj; c0 += SD_BGS_B) {
    const int nc1 = std::min(SD_BGS_B, nproj - c0), c1 = c0 + nc1;
    const bool more = c1 < nproj;
    const int nc2 = more ? std::min(SD_BGS_B, nproj - c1) : 1;
    const double *V2 = more ? V + (int64_t)c1 * ld : vlast;
    double *dst = more ? coef[cur ^ 1] : ctx->d_scalars + slot;
    hipLaunchKernelGGL(k_proj_dots, dim3(nb), dim3(BS), 0, ctx->stream, w, V + (int64_t)c0 * ld, ld, nc1, coef[cur], V2, nc2, N,
                       ctx->d_partials);
    hipLaunchKernelGGL(k_bgs_reduce, dim3(1), dim3(64 * SD_BGS_B), 0, ctx->stream, ctx->d_partials, nb, nc2, dst);
    cur ^= 1;
  }
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

// blocks of the fused groundstate passes (their partial lists are summed by every block of the consumer: at most 1024)
int sd_k_gs_blocks(int64_t N) { return (int)std::min<int64_t>(1024, std::max<int64_t>(1, (N / 2 + BS - 1) / BS)); }
// Blocked Gram-Schmidt of w against V[:,0..ncols-2] (blocks of 8, as sd_k_bgs_chain) ending with the partials of
// alpha = V[:,ncols-1] . w in alpha_part[block][0]; one launch per pass.  y != null: the dots of y with V[:,0..ncols-2] ride along
// and land in chk_dev[0..ncols-2] (device).  scratch: (2 + npass) * nb * 8 doubles, npass = (ncols-2)/8 + 2.
int sd_k_gs_chain(sd_ctx *ctx, double *w, const double *V, int64_t ld, int ncols, int64_t N, const double *y, double *scratch,
                  double *alpha_part, double *chk_dev) {
  const int nb = sd_k_gs_blocks(N);
  const int nproj = ncols - 1;
  const double *vlast = V + (int64_t)(ncols - 1) * ld;
  double *part[2] = {scratch, scratch + (size_t)nb * SD_BGS_B};
  double *chk_part = scratch + 2 * (size_t)nb * SD_BGS_B;
  int cur = 0, pass = 0;
  auto launch = [&](const double *V1, int nc1, const double *pin, const double *V2, int nc2, double *pout, bool chk) {
    double *co = chk_part + (size_t)pass * nb * SD_BGS_B;
    if (chk) hipLaunchKernelGGL(k_gs_pass<true>, dim3(nb), dim3(BS), 0, ctx->stream, w, V1, ld, nc1, pin, nb, V2, nc2, y, N, pout, co);
    else hipLaunchKernelGGL(k_gs_pass<false>, dim3(nb), dim3(BS), 0, ctx->stream, w, V1, ld, nc1, pin, nb, V2, nc2, y, N, pout, co);
  };
  // first pass: dots only (nothing to subtract yet)
  if (nproj > 0) { launch(V, 0, part[cur], V, std::min(SD_BGS_B, nproj), part[cur], y != nullptr); ++pass; }
  else launch(V, 0, part[cur], vlast, 1, alpha_part, false);
  for (int c0 = 0; c0 < nproj; c0 += SD_BGS_B) {
    const int nc1 = std::min(SD_BGS_B, nproj - c0), c1 = c0 + nc1;
    const bool more = c1 < nproj;
    const int nc2 = more ? std::min(SD_BGS_B, nproj - c1) : 1;
    launch(V + (int64_t)c0 * ld, nc1, part[cur], more ? V + (int64_t)c1 * ld : vlast, nc2, more ? part[cur ^ 1] : alpha_part,
           more && y != nullptr);
    if (more) ++pass;
    cur ^= 1;
  }
  if (y && pass > 0) hipLaunchKernelGGL(k_gs_chk_reduce, dim3(pass), dim3(BS), 0, ctx->stream, chk_part, nb, chk_dev);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
// w = (w - alpha v_j) - beta_{j-1} v_{j-1}, then (vjm1 != null) the reference's repair of the v_{j-1} component it has just put back:
// w -= dot(v_{j-1}, w) v_{j-1}.  n2_out: |w|^2 partials of the final w (column 0 of an [nb][8] list); upd_scratch: one more list.
int sd_k_gs_update(sd_ctx *ctx, double *w, const double *vj, const double *vjm1, int64_t N, const double *alpha_part,
                   const double *n2_prev, double *store_alpha, double *store_beta, double *n2_out, double *upd_scratch) {
  const int nb = sd_k_gs_blocks(N);
  hipLaunchKernelGGL(k_gs_update, dim3(nb), dim3(BS), 0, ctx->stream, w, vj, vjm1, N, alpha_part, nb, n2_prev, nb, store_alpha,
                     store_beta, vjm1 ? upd_scratch : n2_out);
  if (vjm1) hipLaunchKernelGGL(k_gs_correct, dim3(nb), dim3(BS), 0, ctx->stream, w, vjm1, N, upd_scratch, nb, n2_out);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
int sd_k_gs_scale(sd_ctx *ctx, double *vnext, const double *w, int64_t N, const double *n2_part, double *store_beta) {
  const int nb = sd_k_gs_blocks(N);
  hipLaunchKernelGGL(k_gs_scale, dim3(std::min(nb * 2, 2048)), dim3(BS), 0, ctx->stream, vnext, w, N, n2_part, nb, store_beta);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

int sd_k_mdot(sd_ctx *ctx, const double *V, int64_t ld, int ncols, const double *y, int64_t N, double *out_host) {
  // out_host[c] = V[:,c] . y for c < ncols (any ncols); one stream synchronisation at the end
  if (ncols <= 0) return SD_OK;
  const int groups = (ncols + SD_MDOT_MAXC - 1) / SD_MDOT_MAXC;
  const int nb = (int)std::min<int64_t>(RED_BLOCKS, std::max<int64_t>(1, (N + BS - 1) / BS));
  int rc = sd_ensure_partials(ctx, (size_t)nb * SD_MDOT_MAXC + (size_t)groups * SD_MDOT_MAXC); if (rc) return rc;
  double *res = ctx->d_partials + (size_t)nb * SD_MDOT_MAXC;            // reduced dots of all groups
  for (int g = 0; g < groups; ++g) {
    const int c0 = g * SD_MDOT_MAXC, nc = std::min(SD_MDOT_MAXC, ncols - c0);
    hipLaunchKernelGGL(k_mdot, dim3(nb), dim3(BS), 0, ctx->stream, V + (int64_t)c0 * ld, ld, nc, y, N, ctx->d_partials);
    hipLaunchKernelGGL(k_mdot_reduce, dim3(1), dim3(64 * SD_MDOT_MAXC), 0, ctx->stream, ctx->d_partials, nb, nc,
                       res + (size_t)g * SD_MDOT_MAXC);
  }
  SD_HIP(ctx, hipGetLastError());
  SD_HIP(ctx, hipMemcpyAsync(out_host, res, sizeof(double) * (size_t)ncols, hipMemcpyDeviceToHost, ctx->stream));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SD_OK;
}

int sd_k_nrm2sq(sd_ctx *ctx, const double *x, int64_t n, int slot) {
  int rc = sd_ensure_partials(ctx, 2 * RED_BLOCKS); if (rc) return rc;
  int nb = (int)std::min<int64_t>(RED_BLOCKS, std::max<int64_t>(1, (n + BS - 1) / BS));
  hipLaunchKernelGGL(k_nrm2sq, dim3(nb), dim3(BS), 0, ctx->stream, x, n, ctx->d_partials);
  hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_partials, nb, ctx->d_scalars + slot);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

int sd_k_imag_count(sd_ctx *ctx, const double *xc, int64_t N, int slot) {
  int rc = sd_ensure_partials(ctx, 2 * RED_BLOCKS); if (rc) return rc;
  int nb = (int)std::min<int64_t>(RED_BLOCKS, std::max<int64_t>(1, (N + BS - 1) / BS));
  hipLaunchKernelGGL(k_imag_count, dim3(nb), dim3(BS), 0, ctx->stream, (const double2 *)xc, N, ctx->d_partials);
  hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_partials, nb, ctx->d_scalars + slot);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}

int sd_read_scalars(sd_ctx *ctx, int slot, int count, double *out) {
  SD_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + slot, ctx->d_scalars + slot, sizeof(double) * count,
                             hipMemcpyDeviceToHost, ctx->stream));
  SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < count; ++i) out[i] = ctx->h_scalars[slot + i];
  return SD_OK;
}

namespace {
// one elementwise pass; when slot >= 0 the pass also reduces |w_new|^2 into ctx->d_scalars[slot]
template <int OP>
int launch_ew(sd_ctx *ctx, double *w, const double *v, const double *u, int64_t n, double a, double b, int slot) {
  const bool vec = (n % 2 == 0) && !((((uintptr_t)w) | ((uintptr_t)v) | ((uintptr_t)u)) & 15);
  const int64_t items = vec ? n / 2 : n;
  unsigned nb = grid_for(items);
  if (slot >= 0) {
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    int rc = sd_ensure_partials(ctx, 2 * RED_BLOCKS); if (rc) return rc;
    if (vec) hipLaunchKernelGGL((k_ew2<OP, true>), dim3(nb), dim3(BS), 0, ctx->stream, (double2 *)w, (const double2 *)v, (const double2 *)u, items, a, b, ctx->d_partials);
    else hipLaunchKernelGGL((k_ew<OP, true>), dim3(nb), dim3(BS), 0, ctx->stream, w, v, u, items, a, b, ctx->d_partials);
    hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_partials, (int)nb, ctx->d_scalars + slot);
  } else {
    if (vec) hipLaunchKernelGGL((k_ew2<OP, false>), dim3(nb), dim3(BS), 0, ctx->stream, (double2 *)w, (const double2 *)v, (const double2 *)u, items, a, b, (double *)nullptr);
    else hipLaunchKernelGGL((k_ew<OP, false>), dim3(nb), dim3(BS), 0, ctx->stream, w, v, u, items, a, b, (double *)nullptr);
  }
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
}  // namespace

int sd_k_scale_div(sd_ctx *ctx, double *y, const double *x, int64_t n, double d) {
  return launch_ew<OP_SCALE_DIV>(ctx, y, x, nullptr, n, d, 0.0, -1);
}
int sd_k_sub2(sd_ctx *ctx, double *w, const double *v, const double *u, int64_t n, double a, double b) {
  return u ? launch_ew<OP_SUB2>(ctx, w, v, u, n, a, b, -1) : launch_ew<OP_SUB2_1>(ctx, w, v, nullptr, n, a, b, -1);
}
int sd_k_lanczos_fold(sd_ctx *ctx, double *t, const double *uc, const double *up, int64_t N, int form, const double *dot_dev,
                      const double *n2c_dev, const double *n2p_dev, double *store_alpha, double *store_bc, double *n2_out) {
  // see k_lanczos_fold; N complex elements; |w|^2 -> n2_out[0] (n2_out[1] = 0), a device address
  // up to 16384 blocks: a 3-read : 1-write stream of this length reaches 5.5 TB/s with them against 5.2 with 2048
  // (profiles/probes/stream_mix_probe.hip; the 1 : 1 copy of the same width: 5.9)
  int rc = sd_ensure_partials(ctx, 2 * RED_BLOCKS); if (rc) return rc;
  unsigned nb = grid_for(N);
  if (nb > RED_BLOCKS) nb = RED_BLOCKS;
  void (*k)(double2 *, const double2 *, const double2 *, int64_t, const double *, const double *, const double *, double *, double *,
            double *) = nullptr;
  switch (form * 2 + (up ? 1 : 0)) {
    case 0: k = k_lanczos_fold<0, false>; break;
    case 1: k = k_lanczos_fold<0, true>; break;
    case 2: k = k_lanczos_fold<1, false>; break;
    case 3: k = k_lanczos_fold<1, true>; break;
    case 4: k = k_lanczos_fold<2, false>; break;
    case 5: k = k_lanczos_fold<2, true>; break;
    default: return sd_set_err(ctx, SD_EINTERNAL, "bad Lanczos update form");
  }
  hipLaunchKernelGGL(k, dim3(nb), dim3(BS), 0, ctx->stream, (double2 *)t, (const double2 *)uc, (const double2 *)up, N, dot_dev,
                     n2c_dev, n2p_dev, store_alpha, store_bc, ctx->d_partials);
  hipLaunchKernelGGL(k_reduce_to, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_partials, (int)nb, n2_out);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
// launch-bound sizes (see k_lanczos_fold_p): `batch` vectors `bstride` elements apart; dotp: the apply's per-tile pairs (ndot per
// vector), n2cp / n2pp: the per-block |w|^2 pairs of the two previous updates (nn2 per vector; null: normalised vector);
// n2out receives nb pairs per vector.  form 0, 1 or 2 (complex alpha: store_alpha receives (re, im)).
int sd_k_lanczos_fold_p(sd_ctx *ctx, double *t, const double *uc, const double *up, int64_t N, int batch, int64_t bstride, int form,
                        const double *dotp, int ndot, const double *n2cp, const double *n2pp, int nn2, double *store_alpha,
                        double *store_bc, int64_t store_stride, double *n2out, int nb) {
  void (*k)(double2 *, const double2 *, const double2 *, int64_t, int64_t, const double *, int, const double *, const double *, int,
            double *, double *, int64_t, double *) = nullptr;
  switch (form * 2 + (up ? 1 : 0)) {
    case 0: k = k_lanczos_fold_p<0, false>; break;
    case 1: k = k_lanczos_fold_p<0, true>; break;
    case 2: k = k_lanczos_fold_p<1, false>; break;
    case 3: k = k_lanczos_fold_p<1, true>; break;
    case 4: k = k_lanczos_fold_p<2, false>; break;
    case 5: k = k_lanczos_fold_p<2, true>; break;
    default: return sd_set_err(ctx, SD_EINTERNAL, "bad Lanczos update form");
  }
  hipLaunchKernelGGL(k, dim3((unsigned)nb, (unsigned)batch), dim3(BS), 0, ctx->stream, (double2 *)t, (const double2 *)uc,
                     (const double2 *)up, N, bstride, dotp, ndot, n2cp, n2pp, nn2, store_alpha, store_bc, store_stride, n2out);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
// few, fat blocks: every block first sums three partial lists (a few microseconds of latency whatever their length), so a block
// should have at least ~2048 elements of its own; at most 256 blocks (= the length of the |w|^2 lists the next step sums)
int sd_k_lanczos_fold_blocks(int64_t N) {
  int64_t nb = (N + 2047) / 2048;
  return (int)std::max<int64_t>(1, std::min<int64_t>(nb, 256));
}
int sd_k_lanczos_fold_scalars_p(sd_ctx *ctx, int batch, int form, const double *dotp, int ndot, const double *n2cp, int nn2,
                                double *store_alpha, double *store_bc, int64_t store_stride) {
  hipLaunchKernelGGL(k_lanczos_fold_scalars_p, dim3((unsigned)batch), dim3(BS), 0, ctx->stream, dotp, ndot, n2cp, nn2, form, store_alpha,
                     store_bc, store_stride);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
int sd_k_lanczos_fold_scalars(sd_ctx *ctx, int form, const double *dot_dev, const double *n2c_dev, double *store_alpha,
                              double *store_bc) {
  hipLaunchKernelGGL(k_lanczos_fold_scalars, dim3(1), dim3(1), 0, ctx->stream, dot_dev, n2c_dev, form, store_alpha, store_bc);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
int sd_k_ccombine(sd_ctx *ctx, double *y, const double *const *cols, int64_t N, int ncols, const double *cr, const double *ci) {
  if (ncols == 0) { SD_HIP(ctx, hipMemsetAsync(y, 0, sizeof(double) * 2 * N, ctx->stream)); return SD_OK; }
  for (int k0 = 0; k0 < ncols; k0 += SD_CCOMB_MAXC) {
    CCombArgs a;
    const int nc = std::min(SD_CCOMB_MAXC, ncols - k0);
    for (int k = 0; k < SD_CCOMB_MAXC; ++k) {
      a.x[k] = k < nc ? (const double2 *)cols[k0 + k] : nullptr;
      a.cr[k] = k < nc ? cr[k0 + k] : 0.0; a.ci[k] = k < nc ? ci[k0 + k] : 0.0;
    }
    hipLaunchKernelGGL(k_ccombine, dim3(grid_for(N)), dim3(BS), 0, ctx->stream, (double2 *)y, N, nc, a, k0 == 0 ? 1 : 0);
    SD_HIP(ctx, hipGetLastError());
  }
  return SD_OK;
}
int sd_k_cheb_init(sd_ctx *ctx, double *y, const double *x0, const double *x1, int64_t N, double c0r, double c0i,
                   double c1r, double c1i, int have1) {
  hipLaunchKernelGGL(k_cheb_init, dim3(grid_for(N)), dim3(BS), 0, ctx->stream, (double2 *)y, (const double2 *)x0,
                     (const double2 *)x1, N, c0r, c0i, c1r, c1i, have1);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
int sd_k_promote(sd_ctx *ctx, double *yc, const double *x, int nc_in, int64_t N) {
  hipLaunchKernelGGL(k_promote, dim3(grid_for(N)), dim3(BS), 0, ctx->stream, (double2 *)yc, x, nc_in, N);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
int sd_k_gemv_cols(sd_ctx *ctx, double *y, const double *V, int64_t N, int ncols, const double *coef_host) {
  for (int k0 = 0; k0 < ncols; k0 += SD_GEMV_MAXC) {
    GemvCoef c;
    int nc = std::min(SD_GEMV_MAXC, ncols - k0);
    for (int k = 0; k < nc; ++k) c.c[k] = coef_host[k0 + k];
    hipLaunchKernelGGL(k_gemv_cols, dim3(grid_for(N)), dim3(BS), 0, ctx->stream, y, V, N, k0, nc, c, k0 == 0 ? 1 : 0);
  }
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
int sd_k_fill_randn(sd_ctx *ctx, double *x, int64_t n, uint64_t seed, uint64_t first) {
  hipLaunchKernelGGL(k_fill_randn, dim3(grid_for(n)), dim3(BS), 0, ctx->stream, x, n, seed, first);
  SD_HIP(ctx, hipGetLastError());
  return SD_OK;
}
