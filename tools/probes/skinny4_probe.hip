// skinny4_probe.hip — where does the time of the FOUR-block skinny GEMM (kernels/skinny.h MB = 4, stored terms: the 33-64-row batched step of round 3) go?
// The product kernel compiled with parts switched off (-DTGX_SKINNY_DIS=bits: 1 activation panel staged once, 2 no fragment reads / MFMAs, 8 no W refills,
// 16 no panel barriers), Llama-3.2-1B gate_up (N = 16384, K = 2048, siluMul epilogue) and down (N = 2048, K = 8192 as 16 K-split slabs), M = 64, against MB = 2 at
// M = 32.  Results are garbage for DIS != 0: timing only.     build: tools/probes/build_skinny4_probe.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels/common.h"
#include "kernels/skinny.h"
#include "kernels/skinny_dma.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int MB, int CFG, int EPI>
static void run(const char* what, int M, int N, int K, int nsplit, const tgx::bf16_t* W, const tgx::bf16_t* ah, const tgx::bf16_t* al, tgx::bf16_t* oh, tgx::bf16_t* ol, float* part) {
  tgx::GemmArgs g{};
  g.A_hi = ah; g.A_lo = al; g.inter = N / 2; g.out_hi = oh; g.out_lo = ol; g.B = W; g.M = M; g.N = N; g.K = K; g.ldc = N;
  if (EPI == tgx::GEMM_PARTIAL) { g.part = part; g.nsplit = nsplit; g.k_per = K / nsplit; }
  auto kern = tgx::skinny_gemm_kernel<tgx::DT_BF16, EPI, MB, 2, CFG, 0>;
  const size_t lds = tgx::skinny_lds_bytes(MB, 2, CFG);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 20; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3((N + tgx::skinny_rows(CFG) - 1) / tgx::skinny_rows(CFG), nsplit), dim3(256), lds, 0, g);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 3 && ms < best) best = ms;
  }
  printf("DIS=%2d %-8s MB %d cfg %d M %2d N %5d K %4d x%2d: %6.1f us  (%.2f TB/s of weights)\n", TGX_SKINNY_DIS, what, MB, CFG, M, N, K, nsplit, best * 1e3, (double)N * K * 2 / (best * 1e-3) / 1e12);
}
template <int MB, int NBW, int EPI>
static void run_dma(const char* what, int M, int N, int K, int nsplit, const tgx::bf16_t* W, const tgx::bf16_t* ah, const tgx::bf16_t* al, tgx::bf16_t* oh, tgx::bf16_t* ol, float* part) {
  tgx::GemmArgs g{};
  g.A_hi = ah; g.A_lo = al; g.inter = N / 2; g.out_hi = oh; g.out_lo = ol; g.B = W; g.M = M; g.N = N; g.K = K; g.ldc = N;
  if (EPI == tgx::GEMM_PARTIAL) { g.part = part; g.nsplit = nsplit; g.k_per = K / nsplit; }
  auto kern = tgx::skinny_dma_kernel<tgx::DT_BF16, EPI, MB, NBW>;
  const size_t lds = tgx::skd_lds_bytes(MB, NBW);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 20; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3((N + 64 * NBW - 1) / (64 * NBW), nsplit), dim3(256), lds, 0, g);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 3 && ms < best) best = ms;
  }
  printf("DMA ring %-8s MB %d nbw %d M %2d N %5d K %4d x%2d: %6.1f us  (%.2f TB/s of weights)  depth %d, %zu KB of LDS\n", what, MB, NBW, M, N, K, nsplit, best * 1e3, (double)N * K * 2 / (best * 1e-3) / 1e12, tgx::skd_depth(MB, NBW), lds / 1024);
}
int main() {
  const size_t NK = (size_t)16384 * 2048;
  std::vector<unsigned short> hw(NK);
  unsigned s = 12345;
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((s >> 16) & 0x1ff) - 0x100 + ((s >> 30) << 15)); }
  constexpr int NC = 12;      // rotating weight copies: nothing repeats out of the Infinity Cache
  tgx::bf16_t* W[NC]; tgx::bf16_t *ah, *al, *oh, *ol; float* part;
  for (int i = 0; i < NC; i++) { CK(hipMalloc(&W[i], NK * 2)); CK(hipMemcpy(W[i], hw.data(), NK * 2, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&ah, (size_t)128 * 8192 * 2)); CK(hipMalloc(&al, (size_t)128 * 8192 * 2));
  CK(hipMemcpy(ah, hw.data(), (size_t)64 * 8192 * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(al, hw.data() + 64 * 8192, (size_t)64 * 8192 * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&oh, (size_t)128 * 8192 * 2)); CK(hipMalloc(&ol, (size_t)128 * 8192 * 2)); CK(hipMalloc(&part, (size_t)16 * 128 * 2048 * 4));
  int rot = 0;
  for (int rep = 0; rep < 2; rep++) {
    run<2, 0, tgx::GEMM_SILU>("gate_up", 32, 16384, 2048, 1, W[(rot++) % NC], ah, al, oh, ol, part);
    run<4, 0, tgx::GEMM_SILU>("gate_up", 64, 16384, 2048, 1, W[(rot++) % NC], ah, al, oh, ol, part);
    run<4, 2, tgx::GEMM_SILU>("gate_up", 64, 16384, 2048, 1, W[(rot++) % NC], ah, al, oh, ol, part);
    run<2, 0, tgx::GEMM_PARTIAL>("down", 32, 2048, 8192, 16, W[(rot++) % NC], ah, al, oh, ol, part);
    run<4, 0, tgx::GEMM_PARTIAL>("down", 64, 2048, 8192, 16, W[(rot++) % NC], ah, al, oh, ol, part);
#if TGX_SKINNY_DIS == 0
    run_dma<1, 1, tgx::GEMM_SILU>("gate_up", 16, 16384, 2048, 1, W[(rot++) % NC], ah, al, oh, ol, part);
    run_dma<2, 1, tgx::GEMM_SILU>("gate_up", 32, 16384, 2048, 1, W[(rot++) % NC], ah, al, oh, ol, part);
    run_dma<4, 1, tgx::GEMM_SILU>("gate_up", 64, 16384, 2048, 1, W[(rot++) % NC], ah, al, oh, ol, part);
    run_dma<4, 2, tgx::GEMM_SILU>("gate_up", 64, 16384, 2048, 1, W[(rot++) % NC], ah, al, oh, ol, part);
    run_dma<2, 1, tgx::GEMM_PARTIAL>("down", 32, 2048, 8192, 16, W[(rot++) % NC], ah, al, oh, ol, part);
    run_dma<4, 1, tgx::GEMM_PARTIAL>("down", 64, 2048, 8192, 16, W[(rot++) % NC], ah, al, oh, ol, part);
    run_dma<4, 1, tgx::GEMM_PARTIAL>("down", 64, 2048, 8192, 8, W[(rot++) % NC], ah, al, oh, ol, part);
    run_dma<8, 1, tgx::GEMM_SILU>("gate_up", 128, 16384, 2048, 1, W[(rot++) % NC], ah, al, oh, ol, part);      // eight blocks (128 rows): experiment
    run_dma<8, 2, tgx::GEMM_SILU>("gate_up", 128, 16384, 2048, 1, W[(rot++) % NC], ah, al, oh, ol, part);
    run_dma<8, 1, tgx::GEMM_PARTIAL>("down", 128, 2048, 8192, 8, W[(rot++) % NC], ah, al, oh, ol, part);
    run_dma<8, 1, tgx::GEMM_PARTIAL>("down", 128, 2048, 8192, 16, W[(rot++) % NC], ah, al, oh, ol, part);
#endif
  }
  return 0;
}
