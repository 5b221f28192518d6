// What does a grid-wide barrier cost on this chip?  (decides whether a whole launch-bound recursion can live in ONE launch)
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/grid_barrier_probe profiles/probes/grid_barrier_probe.hip && gpurun_out/grid_barrier_probe
// G workgroups of 256 threads meet K times at a counter in device memory (release add, acquire spin, bounded); between two
// barriers every workgroup writes one line and reads its neighbour's (so the numbers include making data visible across XCDs).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned *ctr, unsigned target, unsigned *abort_flag) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 22) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return ok;     // (thread 0's verdict; the others learn of an abort at the next barrier)
}

__global__ __launch_bounds__(256) void k_probe(unsigned *ctr, unsigned *abort_flag, double *buf, int K, int work, double *sink) {
  const unsigned G = gridDim.x;
  double acc = 0.0;
  for (int k = 1; k <= K; ++k) {
    if (work) {
      buf[(size_t)blockIdx.x * 256 + threadIdx.x] = (double)k + threadIdx.x;
    }
    grid_barrier(ctr, (unsigned)k * G, abort_flag);
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    if (work) {
      const unsigned nb = (blockIdx.x + 1) % G;
      const double v = buf[(size_t)nb * 256 + threadIdx.x];
      if (v != (double)k + threadIdx.x) acc += 1.0;      // a stale line would be counted
    }
    if (work == 2) {                                     // second barrier per step, as a Lanczos step needs
      grid_barrier(ctr + 32, (unsigned)k * G, abort_flag);
    }
  }
  if (work) atomicAdd(sink, acc);
}

int main() {
  unsigned *ctr, *ab;
  double *buf, *sink;
  CK(hipMalloc(&ctr, 64 * sizeof(unsigned)));
  CK(hipMalloc(&ab, sizeof(unsigned)));
  CK(hipMalloc(&buf, 2048 * 256 * sizeof(double)));
  CK(hipMalloc(&sink, sizeof(double)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int K = 2000;
  for (int coop = 0; coop < 2; ++coop)
    for (int work = 0; work < 3; ++work)
      for (int G : {1, 8, 16, 64, 256, 512, 1024}) {
        CK(hipMemset(ctr, 0, 64 * sizeof(unsigned)));
        CK(hipMemset(ab, 0, sizeof(unsigned)));
        CK(hipMemset(sink, 0, sizeof(double)));
        int Kv = K, wv = work;
        void *args[] = {&ctr, &ab, &buf, &Kv, &wv, &sink};
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(ctr, 0, 64 * sizeof(unsigned)));
          CK(hipEventRecord(e0, 0));
          if (coop) CK(hipLaunchCooperativeKernel((const void *)k_probe, dim3(G), dim3(256), args, 0, 0));
          else hipLaunchKernelGGL(k_probe, dim3(G), dim3(256), 0, 0, ctr, ab, buf, Kv, wv, sink);
          CK(hipGetLastError());
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        unsigned habort;
        double hs;
        CK(hipMemcpy(&habort, ab, sizeof(unsigned), hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hs, sink, sizeof(double), hipMemcpyDeviceToHost));
        printf("{\"coop\": %d, \"work\": %d, \"G\": %d, \"us_per_step\": %.3f, \"aborted\": %u, \"stale\": %.0f}\n", coop, work, G,
               best * 1e3 / K, habort, hs);
        fflush(stdout);
        if (habort) return 1;
      }
  return 0;
}
