// xcc_map_probe.hip — which XCD (XCC_ID) does workgroup (x, y, z) of a grid run on?  (L2-prefetch chaining needs the producer of a
// line and its consumer on the same XCD.)  Build: hipcc -O3 --offload-arch=gfx950 xcc_map_probe.hip -o build/xcc_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (threadIdx.x == 0) out[lin] = id;
}
static void run(dim3 g, int threads, size_t lds) {
  unsigned* d; hipMalloc(&d, 65536 * 4);
  hipLaunchKernelGGL(k, g, dim3(threads), lds, 0, d);
  std::vector<unsigned> h(g.x * g.y * g.z); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (size_t i = 0; i < h.size(); i++) if ((h[i] & 15) != (i % 8)) bad++;
  printf("grid (%u,%u,%u) x %d threads: first 24 XCC_ID:", g.x, g.y, g.z, threads);
  for (size_t i = 0; i < 24 && i < h.size(); i++) printf(" %u", h[i] & 15);
  printf("  | raw[0] 0x%x | workgroups off the (linear id %% 8) rule: %d of %zu\n", h[0], bad, h.size());
  hipFree(d);
}
int main() {
  run(dim3(256), 256, 0); run(dim3(1024), 256, 0); run(dim3(1003), 256, 0); run(dim3(136, 1, 2), 256, 0); run(dim3(32, 1), 256, 0); run(dim3(256), 320, 0); run(dim3(4096), 64, 0);
  return 0;
}
