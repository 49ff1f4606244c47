// tr_read_probe.hip — what does ds_read_b64_tr_b16 return?  LDS holds lds[i] = i; every lane of a 16-lane group points at 4 contiguous elements:
// lane i -> row (i >> 2), columns 4 (i & 3) .. +3 of a [4][16] block with row stride RS; the question is whether lane i then holds column i.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(unsigned short* out, int RS) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  const int off = (i >> 2) * RS + (i & 3) * 4 + g * 16;      // group g: columns 16 g .. 16 g + 15
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
  for (int j = 0; j < 4; j++) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int RS : {64, 68, 72}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, RS);
    std::vector<unsigned short> h(256); hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("RS = %d\n", RS);
    for (int l = 0; l < 64; l++) { if (l % 16 < 6 || l % 16 == 15) printf("  lane %2d: %5d %5d %5d %5d   (column-i expectation: %d %d %d %d)\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3],
        (l>>4)*16 + (l&15), RS + (l>>4)*16 + (l&15), 2*RS + (l>>4)*16 + (l&15), 3*RS + (l>>4)*16 + (l&15)); }
  }
  return 0;
}
