// lane_ops_probe.hip — what v_permlane32_swap / v_permlane16_swap / DPP row_ror:8 do on gfx950 (semantics check for kernels/engine.h eng_reduce8)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned lane = threadIdx.x;
  unsigned a = lane, b = 100 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[lane] = r[0]; o[64 + lane] = r[1];
  auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[128 + lane] = s[0]; o[192 + lane] = s[1];
  o[256 + lane] = __builtin_amdgcn_update_dpp(0, (int)a, 0x128, 0xf, 0xf, false);
  o[320 + lane] = __builtin_amdgcn_update_dpp(0, (int)a, 0x141, 0xf, 0xf, false);
}
int main() {
  unsigned* d; hipMalloc(&d, 384 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[384]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[6] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1", "row_ror8", "half_mirror"};
  for (int t = 0; t < 6; t++) { printf("%-12s", nm[t]); for (int l = 0; l < 64; l++) printf(" %3u", h[t * 64 + l]); printf("\n"); }
  return 0;
}
