// What does a 3-read : 1-write stream (the Lanczos update pass: t, u_cur, u_prev -> t) reach on this chip, and with which loop shape?
//   hipcc --offload-arch=gfx950 -O3 -o profiles/_stream_mix_probe profiles/probes/stream_mix_probe.hip && profiles/_stream_mix_probe
// Variants: grid-stride with E elements in flight per lane (1, 2, 4), plain / non-temporal accesses, blocks per launch; block-contiguous
// chunks; and the 1 : 1 copy of the same width for reference.  N = 601 080 390 double2 elements (the L=32 sector).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ d2 ld(const d2 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(d2 *p, d2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// t <- t/a - b*uc/a - c*up  (the arithmetic of the update, roughly)
__device__ __forceinline__ d2 upd(d2 t, d2 u, d2 p, double a, double b, double c) {
  d2 r;
  r.x = t.x / a - (b * (u.x / a) + c * p.x);
  r.y = t.y / a - (b * (u.y / a) + c * p.y);
  return r;
}

template <int E, bool NT, bool CHUNK>
__global__ __launch_bounds__(256) void k_mix(d2 *__restrict__ t, const d2 *__restrict__ uc, const d2 *__restrict__ up, long N, double a,
                                             double b, double c, double *sink) {
  double s = 0.0;
  if (CHUNK) {
    const long per = (N + gridDim.x - 1) / gridDim.x, lo = (long)blockIdx.x * per, hi = lo + per < N ? lo + per : N;
    for (long i = lo + threadIdx.x; i < hi; i += 256L * E) {
      d2 tv[E], uv[E], pv[E];
#pragma unroll
      for (int e = 0; e < E; ++e) if (i + 256L * e < hi) { tv[e] = ld<NT>(t + i + 256L * e); uv[e] = ld<NT>(uc + i + 256L * e); pv[e] = ld<NT>(up + i + 256L * e); }
#pragma unroll
      for (int e = 0; e < E; ++e) if (i + 256L * e < hi) { const d2 r = upd(tv[e], uv[e], pv[e], a, b, c); st<NT>(t + i + 256L * e, r); s += r.x * r.x + r.y * r.y; }
    }
  } else {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += stride * E) {
      d2 tv[E], uv[E], pv[E];
#pragma unroll
      for (int e = 0; e < E; ++e) if (i + stride * e < N) { tv[e] = ld<NT>(t + i + stride * e); uv[e] = ld<NT>(uc + i + stride * e); pv[e] = ld<NT>(up + i + stride * e); }
#pragma unroll
      for (int e = 0; e < E; ++e) if (i + stride * e < N) { const d2 r = upd(tv[e], uv[e], pv[e], a, b, c); st<NT>(t + i + stride * e, r); s += r.x * r.x + r.y * r.y; }
    }
  }
  if (s == 12345.678) *sink = s;
}

template <bool NT>
__global__ __launch_bounds__(256) void k_copy(d2 *__restrict__ dst, const d2 *__restrict__ src, long N) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += stride) st<NT>(dst + i, ld<NT>(src + i));
}

int main() {
  const long N = 601080390L;
  d2 *t, *uc, *up;
  double *sink;
  CK(hipMalloc(&t, N * 16)); CK(hipMalloc(&uc, N * 16)); CK(hipMalloc(&up, N * 16)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(t, 0, N * 16)); CK(hipMemset(uc, 0, N * 16)); CK(hipMemset(up, 0, N * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](const char *name, int nb, double bytes, auto launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, 0));
      launch();
      CK(hipGetLastError());
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    printf("{\"variant\": \"%s\", \"blocks\": %d, \"ms\": %.3f, \"TBs\": %.3f}\n", name, nb, best, bytes / best / 1e9);
    fflush(stdout);
  };
  const double mixb = 64.0 * N, cpb = 32.0 * N;
  for (int nb : {1024, 2048, 4096, 8192, 16384}) {
    time_it("copy plain", nb, cpb, [&] { hipLaunchKernelGGL(k_copy<false>, dim3(nb), dim3(256), 0, 0, t, uc, N); });
    time_it("copy nt", nb, cpb, [&] { hipLaunchKernelGGL(k_copy<true>, dim3(nb), dim3(256), 0, 0, t, uc, N); });
    time_it("mix E=1 plain stride", nb, mixb, [&] { hipLaunchKernelGGL((k_mix<1, false, false>), dim3(nb), dim3(256), 0, 0, t, uc, up, N, 1.1, 0.3, 0.2, sink); });
    time_it("mix E=1 nt stride", nb, mixb, [&] { hipLaunchKernelGGL((k_mix<1, true, false>), dim3(nb), dim3(256), 0, 0, t, uc, up, N, 1.1, 0.3, 0.2, sink); });
    time_it("mix E=2 nt stride", nb, mixb, [&] { hipLaunchKernelGGL((k_mix<2, true, false>), dim3(nb), dim3(256), 0, 0, t, uc, up, N, 1.1, 0.3, 0.2, sink); });
    time_it("mix E=2 plain stride", nb, mixb, [&] { hipLaunchKernelGGL((k_mix<2, false, false>), dim3(nb), dim3(256), 0, 0, t, uc, up, N, 1.1, 0.3, 0.2, sink); });
    time_it("mix E=4 nt stride", nb, mixb, [&] { hipLaunchKernelGGL((k_mix<4, true, false>), dim3(nb), dim3(256), 0, 0, t, uc, up, N, 1.1, 0.3, 0.2, sink); });
    time_it("mix E=2 nt chunk", nb, mixb, [&] { hipLaunchKernelGGL((k_mix<2, true, true>), dim3(nb), dim3(256), 0, 0, t, uc, up, N, 1.1, 0.3, 0.2, sink); });
    time_it("mix E=4 plain chunk", nb, mixb, [&] { hipLaunchKernelGGL((k_mix<4, false, true>), dim3(nb), dim3(256), 0, 0, t, uc, up, N, 1.1, 0.3, 0.2, sink); });
  }
  return 0;
}
