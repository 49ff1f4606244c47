// skinny_probe.hip — where does the time of the skinny GEMM (kernels/skinny.h) go at the gate_up shape of a decode batch?  The product kernel
// compiled with parts switched off (-DTGX_SKINNY_DIS=bits, see skinny.h), Llama-3.2-1B gate_up: N = 16384, K = 2048, RMSNorm on the way (ASRC 2),
// siluMul epilogue; M = 8 (one 16-row block) and M = 32 (two).  Results are garbage for DIS != 0: timing only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels/common.h"
#include "kernels/skinny.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int MB>
static void run(int M, int N, int K, const tgx::bf16_t* W, const float* X, const tgx::bf16_t* nw, const float* ssq, tgx::bf16_t* oh, tgx::bf16_t* ol) {
  tgx::GemmArgs g{};
  g.A_f32 = X; g.lda = K; g.norm_w = nw; g.ssq_part = ssq; g.ssq_ncb = tgx::SK_NCB; g.eps = 1e-5f;
  g.inter = N / 2; g.out_hi = oh; g.out_lo = ol; g.B = W; g.M = M; g.N = N; g.K = K; g.ldc = N;
#ifndef PROBE_ASRC
#define PROBE_ASRC 2      // 2: fp32 rows, RMSNorm + split while staging (the product's gate_up call); 0: the 16-bit terms precomputed in memory
#endif
  g.A_hi = oh + (size_t)32 * N / 2; g.A_lo = oh + (size_t)32 * N / 2 + 32 * K;      // any initialised memory will do for timing
  auto kern = tgx::skinny_gemm_kernel<tgx::DT_BF16, tgx::GEMM_SILU, MB, 2, 0, PROBE_ASRC>;
  const size_t lds = tgx::skinny_lds_bytes(MB, 2, 0);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 20; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(N / 64), dim3(256), lds, 0, g);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 3 && ms < best) best = ms;
  }
  printf("ASRC=%d DIS=%2d  M %2d  N %d K %d: %.1f us  (%.2f TB/s of weights)\n", PROBE_ASRC, TGX_SKINNY_DIS, M, N, K, best * 1e3, (double)N * K * 2 / (best * 1e-3) / 1e12);
}
int main() {
  const int N = 16384, K = 2048;
  std::vector<unsigned short> hw((size_t)N * K);
  unsigned s = 12345;
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((s >> 16) & 0x1ff) - 0x100 + ((s >> 30) << 15)); }
  std::vector<float> hx((size_t)32 * K), hs(32 * tgx::SK_NCB, (float)K / tgx::SK_NCB);
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  // 16 rotating weight copies (1 GB): nothing repeats out of the Infinity Cache
  constexpr int NC = 16;
  tgx::bf16_t* W[NC]; float *X, *ssq; tgx::bf16_t *nw, *oh, *ol;
  for (int i = 0; i < NC; i++) { CK(hipMalloc(&W[i], (size_t)N * K * 2)); CK(hipMemcpy(W[i], hw.data(), (size_t)N * K * 2, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&X, hx.size() * 4)); CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&ssq, hs.size() * 4)); CK(hipMemcpy(ssq, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&nw, K * 2)); CK(hipMemcpy(nw, hw.data(), K * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&oh, (size_t)32 * N * 2)); CK(hipMemset(oh, 0, (size_t)32 * N * 2)); CK(hipMalloc(&ol, (size_t)32 * N));
  static int rot = 0;
  for (int rep = 0; rep < 2; rep++) {
    run<1>(8, N, K, W[(rot++) % NC], X, nw, ssq, oh, ol);
    run<2>(32, N, K, W[(rot++) % NC], X, nw, ssq, oh, ol);
  }
  return 0;
}
