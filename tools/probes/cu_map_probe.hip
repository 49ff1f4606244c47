// cu_map_probe.hip — which CU does workgroup i of a grid of 512 x 512-thread workgroups (two resident per CU: 43 KB of LDS each, like the key-split prompt
// attention) land on, and when does it start?  (round 6: is "heaviest first" paired (i, i + 256) or (i, i + 8) on the CUs?)
// Build: hipcc -O3 --offload-arch=gfx950 cu_map_probe.hip -o build/cu_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void k(unsigned* out, unsigned long long* t, int spin) {
  extern __shared__ char lds[];
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
  if (threadIdx.x == 0) { out[2 * lin] = xcc; out[2 * lin + 1] = hw; t[lin] = wall_clock64(); }
  // stay resident for a while so that the whole grid is co-resident (work proportional to `spin`)
  float v = threadIdx.x;
  for (int i = 0; i < spin; i++) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) lds[0] = 1;
}
int main() {
  const int N = 512;
  unsigned* d; unsigned long long* t;
  hipMalloc(&d, N * 8); hipMalloc(&t, N * 8);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(32, 16), dim3(512), 43008, 0, d, t, 20000);
  std::vector<unsigned> h(2 * N); std::vector<unsigned long long> ht(N);
  hipMemcpy(h.data(), d, N * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t, N * 8, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx90a+: se_id wider) ...
  std::map<unsigned, std::vector<int>> by_cu;
  unsigned long long t0 = ht[0];
  for (int i = 0; i < N; i++) t0 = ht[i] < t0 ? ht[i] : t0;
  for (int i = 0; i < N; i++) {
    const unsigned key = ((h[2 * i] & 15) << 16) | ((h[2 * i + 1] >> 8) & 0xff) | (((h[2 * i + 1] >> 13) & 7) << 8 << 4);
    by_cu[key].push_back(i);
  }
  printf("distinct (xcc, se, sh, cu): %zu for %d workgroups\n", by_cu.size(), N);
  int shown = 0, d256 = 0, d8 = 0, other = 0;
  for (auto& kv : by_cu) {
    if (shown++ < 12) { printf("  cu key 0x%05x:", kv.first); for (int w : kv.second) printf(" %d (t+%llu0 ns)", w, ht[w] - t0); printf("\n"); }
    if (kv.second.size() == 2) { const int df = kv.second[1] - kv.second[0]; if (df == 256) d256++; else if (df == 8) d8++; else other++; }
  }
  printf("CUs holding 2 workgroups: id difference 256: %d, 8: %d, other: %d\n", d256, d8, other);
  return 0;
}
