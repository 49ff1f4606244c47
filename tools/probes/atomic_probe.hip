// atomic_probe.hip — what do agent-scope 64-bit integer atomics cost as the cross-workgroup reduction of a K-sliced GEMV on MI355X?
// Idea under test (round 4): split o_proj's K range over workgroups by kv group (a workgroup then needs only ITS heads' attention records, so
// the attention combine can ride in o_proj's prologue at 18 KB per workgroup instead of 148 KB) and reduce the partial outputs with
// order-independent FIXED-POINT atomics (integer adds commute: deterministic).  The price is n_addr x per_addr no-return atomics per launch.
// Chain in a hipGraph: [producer: P atomics per address on n_addr addresses] -> [consumer: every workgroup reads all n_addr sums] x 48, against
// the same chain with plain stores (one writer per address).
// Build: hipcc -O3 --offload-arch=gfx950 atomic_probe.hip -o build/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// grid = n_addr * per_addr / 256 workgroups; thread t of workgroup b adds into address (b * 256 + t) % n_addr
template <int MODE>   // 0: plain store (only the first writer of an address), 1: int64 atomic add (no return), 2: fp32 atomic add (no return)
__global__ __launch_bounds__(256) void producer(long long* acc, float* accf, const float* in, int n_addr) {
  const int g = blockIdx.x * 256 + threadIdx.x, a = g % n_addr;
  const float v = in[a] * 0.5f + 1.0f;
  if (MODE == 0) { if (g < n_addr) accf[a] = v; }
  else if (MODE == 1) __hip_atomic_fetch_add(acc + a, (long long)(v * 4294967296.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_fetch_add(accf + a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE>
__global__ __launch_bounds__(256) void consumer(const long long* acc, const float* accf, float* out, int n_addr) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n_addr; i += 256) s += MODE == 1 ? (float)acc[i] * (1.0f / 4294967296.0f) : accf[i];
  for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0 && blockIdx.x < n_addr) out[(blockIdx.x * 4 + (threadIdx.x >> 6)) % n_addr] = s * 1e-6f;
}
__global__ void zero_acc(long long* acc, float* accf, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) { acc[i] = 0; accf[i] = 0.f; } }

int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  long long* acc; float *accf, *vec; CK(hipMalloc(&acc, 8192 * 8)); CK(hipMalloc(&accf, 8192 * 4)); CK(hipMalloc(&vec, 8192 * 4));
  CK(hipMemset(acc, 0, 8192 * 8)); CK(hipMemset(accf, 0, 8192 * 4)); CK(hipMemset(vec, 0, 8192 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int pairs = 48;
  printf("%-36s %10s %10s %10s   (us per producer + consumer pair; consumer = 256 workgroups reading every sum)\n", "addresses x adds per address", "store", "i64 atomic", "f32 atomic");
  for (int n_addr : {2048, 8192}) for (int per : {1, 4, 8, 32, 128}) {
    float res[3];
    for (int mode = 0; mode < 3; mode++) {
      hipGraph_t g; hipGraphExec_t ge;
      const int wgs = n_addr * per / 256;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int it = 0; it < pairs; it++) {
        if (mode == 0) { hipLaunchKernelGGL(producer<0>, dim3(wgs), dim3(256), 0, st, acc, accf, (const float*)vec, n_addr); hipLaunchKernelGGL(consumer<0>, dim3(256), dim3(256), 0, st, (const long long*)acc, (const float*)accf, vec, n_addr); }
        if (mode == 1) { hipLaunchKernelGGL(producer<1>, dim3(wgs), dim3(256), 0, st, acc, accf, (const float*)vec, n_addr); hipLaunchKernelGGL(consumer<1>, dim3(256), dim3(256), 0, st, (const long long*)acc, (const float*)accf, vec, n_addr); }
        if (mode == 2) { hipLaunchKernelGGL(producer<2>, dim3(wgs), dim3(256), 0, st, acc, accf, (const float*)vec, n_addr); hipLaunchKernelGGL(consumer<2>, dim3(256), dim3(256), 0, st, (const long long*)acc, (const float*)accf, vec, n_addr); }
      }
      CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      float best = 1e9f;
      for (int rep = 0; rep < 10; rep++) {
        hipLaunchKernelGGL(zero_acc, dim3(32), dim3(256), 0, st, acc, accf, 8192);
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best) best = ms;
      }
      res[mode] = best * 1000.f / pairs;
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    char nm[64]; snprintf(nm, sizeof nm, "%5d x %3d (%6d adds, %4d WGs)", n_addr, per, n_addr * per, n_addr * per / 256);
    printf("%-36s %10.2f %10.2f %10.2f\n", nm, res[0], res[1], res[2]);
  }
  return 0;
}
