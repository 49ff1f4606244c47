// fetch_calib_probe.hip — what does the PMC counter FETCH_SIZE report for a KNOWN number of bytes, by load instruction?
// "calibrate FETCH_SIZE for global_load_lds on a known byte count".  Three kernels read the same 256 MiB buffer exactly once
// (each 16-byte chunk by exactly one lane): (a) global_load_dwordx4 into registers, (b) the same non-temporal, (c) global_load_lds_dwordx4 (LDS-DMA, the
// instruction of kernels/gemm_dma.h).  A fourth reads a 16 MiB buffer EIGHT times, once per XCD-resident workgroup set (block id % 8 = XCD), to show
// what per-XCD re-reads of an operand cost in FETCH_SIZE (the prefill GEMMs' activation tiles).
//   hipcc -O3 --offload-arch=gfx950 fetch_calib_probe.hip -o build/fetch_calib_probe
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o f -- build/fetch_calib_probe      (then tools/rocpd_pmc.py on the .db)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int PER_WG = 64 * 1024;            // bytes per workgroup: 256 threads x 16 loads x 16 bytes

__global__ __launch_bounds__(256) void k_plain(const u32x4* src, unsigned* sink) {
  const u32x4* p = src + (size_t)blockIdx.x * (PER_WG / 16) + threadIdx.x;
  u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 16; i++) { const u32x4 v = p[i * 256]; acc ^= v; }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1u;
}
__global__ __launch_bounds__(256) void k_nt(const u32x4* src, unsigned* sink) {
  const u32x4* p = src + (size_t)blockIdx.x * (PER_WG / 16) + threadIdx.x;
  u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 16; i++) { const u32x4 v = __builtin_nontemporal_load(p + i * 256); acc ^= v; }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1u;
}
__device__ __forceinline__ void dma_1k(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int per_xcd_copy>
__global__ __launch_bounds__(256) void k_dma(const u32x4* src, unsigned* sink, size_t wrap_chunks) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // per_xcd_copy: the workgroups of one XCD (block id % 8) together read the (small) buffer once -> 8 reads of it chip-wide
  const size_t wg = per_xcd_copy ? (size_t)(blockIdx.x / 8) : (size_t)blockIdx.x;
  const u32x4* p = src + (wg * (PER_WG / 16)) % wrap_chunks + wv * 64 + lane;
  const unsigned base = (unsigned)(size_t)lds;
#pragma unroll
  for (int i = 0; i < 16; i++) dma_1k(p + i * 256, base + (unsigned)((i * 4 + wv) * 1024));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned v = reinterpret_cast<const unsigned*>(lds)[threadIdx.x * 7 % 16384];
  if (v == 0x12345678u) *sink = 1u;
}

int main() {
  const size_t big = 256ull << 20, small = 16ull << 20;
  unsigned char *a, *b; unsigned* sink;
  CK(hipMalloc(&a, big)); CK(hipMalloc(&b, small)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, big)); CK(hipMemset(b, 2, small)); CK(hipMemset(sink, 0, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dma<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const int nwg = (int)(big / PER_WG);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_plain, dim3(nwg), dim3(256), 0, 0, (const u32x4*)a, sink);
    hipLaunchKernelGGL(k_nt, dim3(nwg), dim3(256), 0, 0, (const u32x4*)a, sink);
    hipLaunchKernelGGL(k_dma<0>, dim3(nwg), dim3(256), 65536, 0, (const u32x4*)a, sink, big / 16);
    // 16 MiB read once per XCD: 8 x 256 workgroups, workgroup w of XCD x reads chunk w
    hipLaunchKernelGGL(k_dma<1>, dim3((int)(small / PER_WG) * 8), dim3(256), 65536, 0, (const u32x4*)b, sink, small / 16);
    CK(hipDeviceSynchronize());
  }
  printf("bytes per launch: k_plain / k_nt / k_dma %zu (read once); k_dma per-XCD copy: %zu algorithmic, %zu with one copy per XCD\n", big, small, small * 8);
  return 0;
}
