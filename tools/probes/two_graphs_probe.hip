// Probe: two hipGraphs (each a linear chain) launched on two different streams — do they run concurrently?
// Graph A: waiter(flag). Graph B: setter(flag).  A is launched first.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void waiter(unsigned* flag, unsigned* result, long long max_spins) {
  if (threadIdx.x == 0) {
    long long n = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && n < max_spins) { __builtin_amdgcn_s_sleep(10); n++; }
    result[blockIdx.x] = (n < max_spins) ? 1u : 2u;
  }
}
__global__ void setter(unsigned* flag, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void nop() {}
int main() {
  unsigned *flag, *res; unsigned h[512];
  CK(hipMalloc(&flag, 4)); CK(hipMalloc(&res, 512 * 4));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const long long spins = 100000;
  hipGraph_t gA, gB; hipGraphExec_t eA, eB;
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s1);
  hipLaunchKernelGGL(waiter, dim3(256), dim3(256), 0, s1, flag, res, spins);
  hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s1);
  CK(hipStreamEndCapture(s1, &gA)); CK(hipGraphInstantiate(&eA, gA, nullptr, nullptr, 0));
  CK(hipStreamBeginCapture(s2, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s2);
  hipLaunchKernelGGL(setter, dim3(1), dim3(64), 0, s2, flag, 1u);
  hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s2);
  CK(hipStreamEndCapture(s2, &gB)); CK(hipGraphInstantiate(&eB, gB, nullptr, nullptr, 0));
  for (int rep = 0; rep < 4; rep++) {
    CK(hipMemset(flag, 0, 4)); CK(hipMemset(res, 0, 512 * 4)); CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(eA, s1)); CK(hipGraphLaunch(eB, s2));
    CK(hipDeviceSynchronize());
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    CK(hipMemcpy(h, res, 256 * 4, hipMemcpyDeviceToHost));
    int ok = 0, to = 0; for (int i = 0; i < 256; i++) { ok += h[i] == 1; to += h[i] == 2; }
    printf("two graphs on two streams (waiter graph launched first), rep %d: saw flag %d, timed out %d, %.2f ms\n", rep, ok, to, ms);
  }
  return 0;
}
