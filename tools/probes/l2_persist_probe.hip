// l2_persist_probe.hip — do the XCD L2s keep lines across a kernel boundary on MI355X (ROCm 7.2)?
// If they do, a latency-bound kernel (attention, qkv) could pull the NEXT kernel's first weights into the L2 of the XCD whose workgroups
// will read them; if the boundary's acquire drops them, only LDS (a persistent kernel) or the memory-side Infinity Cache survive a boundary.
//   pair S: kernel A reads region X (default policy), kernel B reads region X with the same workgroup -> bytes map   (L2 hits if lines persist)
//   pair D: kernel A reads region Y,                  kernel B reads region X                                        (B always from memory)
// Both pairs move the same bytes; time(S) < time(D) only if B hits in L2.  Regions rotate through a 1 GiB pool so that nothing repeats
// out of the Infinity Cache between iterations.  Also: one kernel reading the same region twice (in-kernel L2 reuse) for scale.
// Build: hipcc -O3 --offload-arch=gfx950 l2_persist_probe.hip -o build/l2_persist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ p, size_t n16_per_wg, unsigned* sink, int passes) {
  const u32x4* q = p + (size_t)blockIdx.x * n16_per_wg;
  unsigned acc = 0;
  for (int ps = 0; ps < passes; ps++)
    for (size_t i = threadIdx.x; i < n16_per_wg; i += 256) {
      const u32x4 v = NT ? __builtin_nontemporal_load(q + i) : q[i];
      acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
  if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t pool_bytes = (size_t)1 << 30;
  unsigned char* pool; CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 1, pool_bytes));
  unsigned* sink; CK(hipMalloc(&sink, 1 << 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int WG = 1024;
  for (size_t mb : {4, 8, 16, 24, 48}) {
    const size_t bytes = mb << 20, n16 = bytes / 16 / WG;
    const int nreg = (int)(pool_bytes / bytes);
    auto run = [&](int mode, bool nt_b) {   // mode 0: S pairs, 1: D pairs, 2: single kernel two passes, 3: B alone
      const int iters = 64;
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, st));
        for (int it = 0; it < iters; it++) {
          const u32x4* X = (const u32x4*)(pool + (size_t)((2 * it) % nreg) * bytes);
          const u32x4* Y = (const u32x4*)(pool + (size_t)((2 * it + 1) % nreg) * bytes);
          if (mode == 2) { hipLaunchKernelGGL(read_kernel<false>, dim3(WG), dim3(256), 0, st, X, n16, sink, 2); continue; }
          if (mode != 3) hipLaunchKernelGGL(read_kernel<false>, dim3(WG), dim3(256), 0, st, mode == 0 ? X : Y, n16, sink, 1);
          if (nt_b) hipLaunchKernelGGL(read_kernel<true>, dim3(WG), dim3(256), 0, st, X, n16, sink, 1);
          else hipLaunchKernelGGL(read_kernel<false>, dim3(WG), dim3(256), 0, st, X, n16, sink, 1);
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      return best * 1000.0f / iters;
    };
    const float s = run(0, false), d = run(1, false), s_nt = run(0, true), d_nt = run(1, true), two = run(2, false), one = run(3, false);
    printf("%3zu MB: A(X)+B(X) %7.2f us   A(Y)+B(X) %7.2f us   [B nt: %7.2f / %7.2f]   one kernel, two passes %7.2f us   B alone %7.2f us\n", mb, s, d, s_nt, d_nt, two, one);
  }
  return 0;
}
