// Probe: do two kernels on parallel branches of a hipGraph (or on two streams) run CONCURRENTLY on MI355X?
// Kernel W spins (bounded) until kernel S sets a flag.  If they are serialized W-before-S, W times out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void waiter(unsigned* flag, unsigned* result, long long max_spins) {
  if (threadIdx.x == 0) {
    long long n = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && n < max_spins) { __builtin_amdgcn_s_sleep(10); n++; }
    result[blockIdx.x] = (n < max_spins) ? 1u : 2u;   // 1 = saw the flag, 2 = timed out
  }
}
__global__ void setter(unsigned* flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  unsigned *flag, *res; unsigned h[512];
  CK(hipMalloc(&flag, 4)); CK(hipMalloc(&res, 512 * 4));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const long long spins = 200000;   // ~ 200000 * 640 cycles = 53 ms at 2.4 GHz
  for (int order = 0; order < 2; order++) {
    // eager, two streams
    CK(hipMemset(flag, 0, 4)); CK(hipMemset(res, 0, 512 * 4)); CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    if (order == 0) { hipLaunchKernelGGL(waiter, dim3(512), dim3(256), 0, s1, flag, res, spins); hipLaunchKernelGGL(setter, dim3(1), dim3(64), 0, s2, flag); }
    else { hipLaunchKernelGGL(setter, dim3(1), dim3(64), 0, s2, flag); hipLaunchKernelGGL(waiter, dim3(512), dim3(256), 0, s1, flag, res, spins); }
    CK(hipDeviceSynchronize());
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    CK(hipMemcpy(h, res, 512 * 4, hipMemcpyDeviceToHost));
    int ok = 0, to = 0; for (int i = 0; i < 512; i++) { ok += h[i] == 1; to += h[i] == 2; }
    printf("eager 2 streams, %s first: saw flag %d, timed out %d, %.2f ms\n", order == 0 ? "waiter" : "setter", ok, to, ms);
  }
  // graph with two parallel branches: fork from s1 to s2 during capture
  for (int order = 0; order < 2; order++) {
    hipEvent_t fork, join; CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(flag, 0, 4, s1));
    CK(hipEventRecord(fork, s1)); CK(hipStreamWaitEvent(s2, fork, 0));
    if (order == 0) { hipLaunchKernelGGL(waiter, dim3(512), dim3(256), 0, s1, flag, res, spins); hipLaunchKernelGGL(setter, dim3(1), dim3(64), 0, s2, flag); }
    else { hipLaunchKernelGGL(setter, dim3(1), dim3(64), 0, s2, flag); hipLaunchKernelGGL(waiter, dim3(512), dim3(256), 0, s1, flag, res, spins); }
    CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s1, join, 0));
    CK(hipStreamEndCapture(s1, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
      CK(hipMemset(res, 0, 512 * 4)); CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      CK(hipGraphLaunch(ge, s1)); CK(hipStreamSynchronize(s1));
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      CK(hipMemcpy(h, res, 512 * 4, hipMemcpyDeviceToHost));
      int ok = 0, to = 0; for (int i = 0; i < 512; i++) { ok += h[i] == 1; to += h[i] == 2; }
      printf("graph 2 branches, %s captured first, rep %d: saw flag %d, timed out %d, %.2f ms\n", order == 0 ? "waiter" : "setter", rep, ok, to, ms);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
