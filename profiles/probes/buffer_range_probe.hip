// Is the range check of a raw buffer load applied per dword or per instruction on gfx950?  (decides whether two Float64 rows
// can share one 16-byte far-bond load when a tile has an odd number of rows)
//   hipcc --offload-arch=gfx950 -O3 -o profiles/_buffer_range_probe profiles/probes/buffer_range_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *p, int nbytes, double *out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p), 0, nbytes, 0x00020000);
  const u4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16u, 0, 0);
  out[2 * threadIdx.x] = __hiloint2double((int)raw.y, (int)raw.x);
  out[2 * threadIdx.x + 1] = __hiloint2double((int)raw.w, (int)raw.z);
  // second experiment: a load whose byte offset is "negative" (wrapped) by 8: rows below a window
  const u4 raw2 = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16u - 8u, 0, 0);
  out[128 + 2 * threadIdx.x] = __hiloint2double((int)raw2.y, (int)raw2.x);
  out[128 + 2 * threadIdx.x + 1] = __hiloint2double((int)raw2.w, (int)raw2.z);
}
int main() {
  double h[64], *d, *o, ho[256];
  for (int i = 0; i < 64; ++i) h[i] = i + 1;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, 5 * 8, o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  printf("num_records = 40 bytes (5 doubles), b128 at lane*16:\n");
  for (int t = 0; t < 4; ++t) printf("  lane %d: %g %g\n", t, ho[2 * t], ho[2 * t + 1]);
  printf("b128 at lane*16 - 8 (wrapped for lane 0):\n");
  for (int t = 0; t < 4; ++t) printf("  lane %d: %g %g\n", t, ho[128 + 2 * t], ho[128 + 2 * t + 1]);
  return 0;
}
