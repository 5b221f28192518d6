// Two-pass probe, pass "P2" (VERDICT r02 item 1): a stand-alone gfx950 kernel that computes only the hops on the bonds
// 1..m (sites 1..m+1, the TOP of the basis tree) of an open XXZ chain in a fixed-nup sector:
//     out[row] (+)= sum_{b <= m, bond b flippable in A} J_b * psi[partner_b(row)]
// Rows sharing a configuration A of sites 1..a (a = m+1) are one contiguous block; a hop on a bond b <= m maps block A onto
// block A ^ (3 << (b-1)) at the SAME offset.  A "column super-tile" = {all A of one filling} x {one chunk of G consecutive
// offsets} is closed under all m bonds; its psi rows (n_A * G * 16 B, e.g. 462 * 4 KiB = 1.9 MB) are meant to sit in ONE
// XCD's L2 while the workgroups (one per (A, chunk)) of that super-tile run on that XCD, so every psi row leaves HBM once:
// 16 B/row read + 16 B/row written (MODE 0: out = S) or + 16 B/row read (MODE 1: out += S).
// Not product code: built and driven by profiles/probe_twopass.py only.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Item { int64_t row0; int32_t n; uint32_t A; };   // first row of the segment (global index), rows (<= G), prefix configuration

typedef double d2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double2 buf_load2(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const u4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  double2 v;
  v.x = __hiloint2double((int)raw.y, (int)raw.x);
  v.y = __hiloint2double((int)raw.w, (int)raw.z);
  return v;
}

template <int R, int BLOCK, int MAXM, int MODE>
__global__ __launch_bounds__(BLOCK) void k_p2(const double2 *__restrict__ psi, double2 *__restrict__ out,
                                              const Item *__restrict__ items, const int64_t *__restrict__ blk_base, int m,
                                              const double *__restrict__ J) {
  const Item it = items[blockIdx.x];
  if (it.n <= 0) return;
  const uint32_t off0 = (uint32_t)threadIdx.x * 16u;
  const int64_t off = it.row0 - blk_base[it.A];       // offset of the segment inside its block: the same in every partner block
  double2 v[MAXM][R];
  double Jb[MAXM];
#pragma unroll
  for (int b = 0; b < MAXM; ++b) {
    const bool fl = b < m && (((it.A >> b) ^ (it.A >> (b + 1))) & 1u);
    const uint32_t Ap = fl ? (it.A ^ (3u << b)) : it.A;
    const double2 *pb = psi + (blk_base[Ap] + off);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(pb, fl ? (uint32_t)it.n * 16u : 0u);    // no hop: every load returns 0, no traffic
    Jb[b] = fl ? J[b] : 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) v[b][r] = buf_load2(rs, off0 + (uint32_t)(r * BLOCK) * 16u);
  }
  double2 acc[R];
  if (MODE == 1) {
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out + it.row0, (uint32_t)it.n * 16u);
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = buf_load2(ro, off0 + (uint32_t)(r * BLOCK) * 16u);
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_double2(0.0, 0.0);
  }
#pragma unroll
  for (int b = 0; b < MAXM; ++b)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      acc[r].x = __builtin_fma(Jb[b], v[b][r].x, acc[r].x);
      acc[r].y = __builtin_fma(Jb[b], v[b][r].y, acc[r].y);
    }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = (int)threadIdx.x + r * BLOCK;
    if (i < it.n) {
      d2v w; w.x = acc[r].x; w.y = acc[r].y;
      __builtin_nontemporal_store(w, reinterpret_cast<d2v *>(out + it.row0 + i));
    }
  }
}

// LDS form: one workgroup per (filling, chunk of G rows): the n_A x G rows are gathered into LDS (segments of G*16 B), the
// m bonds are LDS reads.  cfg[k] lists the A of a filling in a fixed order; nbr[k][j][b] = position of A_j ^ bond b in that
// list or -1.  Traffic is 32 / 48 B/row whatever the L2 does; the price is short segments (G*16 B) and one big LDS image.
struct LItem { int64_t off; int32_t n; int32_t cls; };   // offset inside every block of the class, rows (<= G), class index
template <int G, int BLOCK, int MAXM, int MODE>
__global__ __launch_bounds__(BLOCK) void k_p2_lds(const double2 *__restrict__ psi, double2 *__restrict__ out,
                                                  const LItem *__restrict__ items, const int32_t *__restrict__ cls_off,
                                                  const int64_t *__restrict__ cls_base, const int16_t *__restrict__ nbr, int m,
                                                  const double *__restrict__ J) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2 *img = reinterpret_cast<double2 *>(smem);        // [nA][G]
  const LItem it = items[blockIdx.x];
  if (it.n <= 0) return;
  const int a0 = cls_off[it.cls], nA = cls_off[it.cls + 1] - a0;
  const int total = nA * G;
  const int g = threadIdx.x % G;
  for (int e = threadIdx.x; e < total; e += BLOCK) {
    const int j = e / G;
    double2 x = make_double2(0.0, 0.0);
    if (g < it.n) x = psi[cls_base[a0 + j] + it.off + g];
    img[e] = x;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < total; e += BLOCK) {
    const int j = e / G;
    if (g >= it.n) continue;
    const int64_t row = cls_base[a0 + j] + it.off + g;
    double2 acc = MODE == 1 ? out[row] : make_double2(0.0, 0.0);
    const int16_t *nb = nbr + (size_t)(a0 + j) * MAXM;
#pragma unroll
    for (int b = 0; b < MAXM; ++b) {
      const int q = b < m ? nb[b] : -1;
      if (q >= 0) {
        const double2 x = img[q * G + g];
        acc.x = __builtin_fma(J[b], x.x, acc.x);
        acc.y = __builtin_fma(J[b], x.y, acc.y);
      }
    }
    d2v w; w.x = acc.x; w.y = acc.y;
    __builtin_nontemporal_store(w, reinterpret_cast<d2v *>(out + row));
  }
}

template <int R, int BLOCK>
static int launch(int mode, const void *psi, void *out, const void *items, int64_t n_items, const void *blk_base, int m,
                  const void *J, hipStream_t st) {
  if (mode == 0)
    hipLaunchKernelGGL((k_p2<R, BLOCK, 12, 0>), dim3((unsigned)n_items), dim3(BLOCK), 0, st, (const double2 *)psi, (double2 *)out,
                       (const Item *)items, (const int64_t *)blk_base, m, (const double *)J);
  else
    hipLaunchKernelGGL((k_p2<R, BLOCK, 12, 1>), dim3((unsigned)n_items), dim3(BLOCK), 0, st, (const double2 *)psi, (double2 *)out,
                       (const Item *)items, (const int64_t *)blk_base, m, (const double *)J);
  return (int)hipGetLastError();
}

extern "C" int probe_p2_launch(int G, int mode, const void *psi, void *out, const void *items, int64_t n_items,
                               const void *blk_base, int m, const void *J, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (m > 12) return -1;
  switch (G) {
    case 64: return launch<1, 64>(mode, psi, out, items, n_items, blk_base, m, J, st);
    case 128: return launch<1, 128>(mode, psi, out, items, n_items, blk_base, m, J, st);
    case 256: return launch<1, 256>(mode, psi, out, items, n_items, blk_base, m, J, st);
    case 512: return launch<2, 256>(mode, psi, out, items, n_items, blk_base, m, J, st);
    case 1024: return launch<2, 512>(mode, psi, out, items, n_items, blk_base, m, J, st);
  }
  return -2;
}

template <int G, int BLOCK>
static int launch_lds(int mode, const void *psi, void *out, const void *items, int64_t n_items, const void *cls_off,
                      const void *cls_base, const void *nbr, int m, const void *J, int max_nA, hipStream_t st) {
  const size_t shmem = (size_t)max_nA * G * 16;
  auto k0 = k_p2_lds<G, BLOCK, 12, 0>;
  auto k1 = k_p2_lds<G, BLOCK, 12, 1>;
  if (shmem > 48 * 1024) {
    hipFuncSetAttribute((const void *)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  }
  if (mode == 0)
    hipLaunchKernelGGL(k0, dim3((unsigned)n_items), dim3(BLOCK), shmem, st, (const double2 *)psi, (double2 *)out,
                       (const LItem *)items, (const int32_t *)cls_off, (const int64_t *)cls_base, (const int16_t *)nbr, m,
                       (const double *)J);
  else
    hipLaunchKernelGGL(k1, dim3((unsigned)n_items), dim3(BLOCK), shmem, st, (const double2 *)psi, (double2 *)out,
                       (const LItem *)items, (const int32_t *)cls_off, (const int64_t *)cls_base, (const int16_t *)nbr, m,
                       (const double *)J);
  return (int)hipGetLastError();
}

extern "C" int probe_p2_lds_launch(int G, int mode, const void *psi, void *out, const void *items, int64_t n_items,
                                   const void *cls_off, const void *cls_base, const void *nbr, int m, const void *J, int max_nA,
                                   void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (m > 12) return -1;
  switch (G) {
    case 8: return launch_lds<8, 512>(mode, psi, out, items, n_items, cls_off, cls_base, nbr, m, J, max_nA, st);
    case 16: return launch_lds<16, 1024>(mode, psi, out, items, n_items, cls_off, cls_base, nbr, m, J, max_nA, st);
    case 32: return launch_lds<32, 1024>(mode, psi, out, items, n_items, cls_off, cls_base, nbr, m, J, max_nA, st);
  }
  return -2;
}
