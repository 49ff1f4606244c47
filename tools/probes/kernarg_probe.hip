// kernarg_probe.hip — what does the kernel-argument fetch cost a short dependent launch on MI355X, and does gfx950's kernarg PRELOAD
// (user SGPRs filled by the dispatcher: -mllvm -amdgpu-kernarg-preload-count=N) remove it?
// A decode step is 98 dependent launches of 4-13 us; each starts with s_load of its argument block (a by-value struct: never preloaded),
// and only then can it issue its first weight load.  Chain of dependent GEMV-shaped launches (every launch reads the vector the previous
// one wrote, streams its own weight region, writes the next vector), captured in a hipGraph, three argument forms:
//   S  one by-value struct (the product's form)                         -> s_load, then everything else
//   F  flat leading arguments (W, x, y, n16, ...)                       -> preloaded into SGPRs when built with the flag, s_load otherwise
// Build twice: hipcc -O3 --offload-arch=gfx950 kernarg_probe.hip -o build/kernarg_probe
//              hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 kernarg_probe.hip -o build/kernarg_probe_pre
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct Args {
  const u32x4* W; const float* x; float* y; size_t n16_per_wg; int xn; int pad0;
  long long filler[24];      // the product's GemvArgs is ~400 bytes
};

// the body: x slice into registers (dependent input), weight slices streamed 8 x 16 B per lane in flight, one float out per workgroup
__device__ __forceinline__ void body(const u32x4* __restrict__ W, const float* __restrict__ x, float* __restrict__ y, size_t n16_per_wg, int xn) {
  const u32x4* q = W + (size_t)blockIdx.x * n16_per_wg;
  const f32x4 xv = reinterpret_cast<const f32x4*>(x)[threadIdx.x % (xn / 4)];
  float acc = 0.f;
  for (size_t i = threadIdx.x; i < n16_per_wg; i += 256 * 8) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = __builtin_nontemporal_load(q + (i + (size_t)j * 256 < n16_per_wg ? i + (size_t)j * 256 : i));
#pragma unroll
    for (int j = 0; j < 8; j++) acc += __uint_as_float(v[j][0] & 0x3f800000u) * xv[0] + __uint_as_float(v[j][1] & 0x3f800000u) * xv[1] + __uint_as_float(v[j][2] & 0x3f800000u) * xv[2] + __uint_as_float(v[j][3] & 0x3f800000u) * xv[3];
  }
  for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) y[blockIdx.x % xn] = (red[0] + red[1] + red[2] + red[3]) * 1e-9f + 1.0f;
}

__global__ __launch_bounds__(256) void k_struct(const Args a) { body(a.W, a.x, a.y, a.n16_per_wg, a.xn); }
__global__ __launch_bounds__(256) void k_flat(const u32x4* W, const float* x, float* y, size_t n16_per_wg, int xn, const Args rest) { body(W, x, y, n16_per_wg, xn); if (n16_per_wg == 0x7fffffffffffull) y[1] = (float)rest.filler[3]; }

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t pool_bytes = (size_t)2 << 30;
  unsigned char* pool; CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 0x3f, pool_bytes));
  float* vec[2]; for (auto& v : vec) { CK(hipMalloc(&v, 8192 * 4)); CK(hipMemset(v, 0, 8192 * 4)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int xn = 2048, launches = 96;
  printf("%-28s %10s %10s\n", "launch shape", "struct us", "flat us");
  for (int wg : {256, 1024}) for (size_t kb : {0, 1024, 8192, 12288, 32768, 65536}) {
    const size_t bytes = kb << 10, n16 = bytes / 16 / wg;
    float res[2];
    for (int form = 0; form < 2; form++) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int it = 0; it < launches; it++) {
        const u32x4* W = (const u32x4*)(pool + ((size_t)it * (bytes ? bytes : 4096)) % (pool_bytes - bytes - 4096));
        if (form == 0) { Args a{}; a.W = W; a.x = vec[it & 1]; a.y = vec[(it + 1) & 1]; a.n16_per_wg = n16; a.xn = xn; hipLaunchKernelGGL(k_struct, dim3(wg), dim3(256), 0, st, a); }
        else { Args r{}; hipLaunchKernelGGL(k_flat, dim3(wg), dim3(256), 0, st, W, (const float*)vec[it & 1], vec[(it + 1) & 1], n16, xn, r); }
      }
      CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      float best = 1e9f;
      for (int rep = 0; rep < 12; rep++) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best) best = ms;
      }
      res[form] = best * 1000.f / launches;
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    char nm[64]; snprintf(nm, sizeof nm, "%4d WGs x %6zu KB", wg, kb);
    printf("%-28s %10.2f %10.2f\n", nm, res[0], res[1]);
  }
  return 0;
}
