// attn_prefill_probe.hip — where does the time of attn_prefill_kernel go?  The product kernel (kernels/prefill.h) compiled with parts switched off
// at compile time (-DTGX_ATTN_DIS=bits: 1 no LDS staging, 2 no softmax arithmetic, 4 no PV, 8 no QK^T, 16 no tile fetch), timed on the Llama-3.2-1B
// prefill shape (S = 2048, 32 query heads, 8 kv heads, head_dim 64) with random data.  Results are garbage for DIS != 0: timing only.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I tinygpt_amd/csrc -I include -DTGX_ATTN_DIS=<bits> attn_prefill_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "kernels/common.h"
#include "kernels/prefill.h"
#include "kernels/attn_prefill_dma.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#ifndef PROBE_LA
#define PROBE_LA 2
#endif
#ifndef PROBE_HD
#define PROBE_HD 64
#endif
int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 2048, heads = 32, kvh = 8, HD = PROBE_HD, max_ctx = S;
  const size_t nq = (size_t)S * heads * HD, nkv = (size_t)kvh * max_ctx * HD;
  std::vector<unsigned short> h(nq > nkv ? nq : nkv);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3f00 + ((s >> 16) & 0xff) - 0x80 + ((s >> 30) << 15)); }   // ~ +-0.5..1
  tgx::bf16_t *qh, *ql, *k, *v, *oh, *ol;
  CK(hipMalloc(&qh, nq * 2)); CK(hipMalloc(&ql, nq * 2)); CK(hipMalloc(&oh, nq * 2)); CK(hipMalloc(&ol, nq * 2));
  CK(hipMalloc(&k, nkv * 2)); CK(hipMalloc(&v, nkv * 2));
  CK(hipMemcpy(qh, h.data(), nq * 2, hipMemcpyHostToDevice)); CK(hipMemset(ql, 0, nq * 2));
  CK(hipMemcpy(k, h.data(), nkv * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(v, h.data(), nkv * 2, hipMemcpyHostToDevice));
  tgx::AttnPrefillArgs a{};
  a.q_hi = qh; a.q_lo = ql; a.k_cache = k; a.v_cache = v; a.o_hi = oh; a.o_lo = ol;
  a.S = S; a.heads = heads; a.kv_heads = kvh; a.max_ctx = max_ctx; a.past = 0; a.scale = 0.125f; a.qblk_mirror = 1;
#ifndef PROBE_KP
#define PROBE_KP 1
#endif
  a.heavy_first = PROBE_KP == 2;
  const dim3 grid = PROBE_KP == 2 ? dim3(heads, (S + 127) / 128) : dim3((S + 127) / 128, heads), blk(256 * PROBE_KP);
  const size_t lds = (size_t)PROBE_KP * (64 * (PROBE_HD + 8) + 64 * (PROBE_HD + 32)) * 2;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_prefill_kernel<tgx::DT_BF16, PROBE_HD, PROBE_LA, PROBE_KP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
#ifdef PROBE_DMA
    tgx::AttnPrefillArgs b = a; b.heavy_first = 1;
    hipLaunchKernelGGL((tgx::attn_prefill_dma_kernel<tgx::DT_BF16>), dim3(heads, (S + 127) / 128), dim3(256), (size_t)2 * 3 * 64 * 64 * 2, 0, b);
#else
    hipLaunchKernelGGL((tgx::attn_prefill_kernel<tgx::DT_BF16, PROBE_HD, PROBE_LA, PROBE_KP>), grid, blk, lds, 0, a);
#endif
  };
  float best = 1e30f;
  for (int r = 0; r < 12; r++) {
    CK(hipEventRecord(e0, 0));
    launch();
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2 && ms < best) best = ms;
  }
#ifdef PROBE_DMA
  {   // value check against attn_prefill_kernel (LA 2, KP 1) on the same data: both hi words and the reconstructed hi + lo
    std::vector<unsigned short> x(nq), xl(nq), y(nq), yl(nq);
    CK(hipMemcpy(x.data(), oh, nq * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(xl.data(), ol, nq * 2, hipMemcpyDeviceToHost));
    a.heavy_first = 0;
    hipLaunchKernelGGL((tgx::attn_prefill_kernel<tgx::DT_BF16, 64, 2, 1>), dim3((S + 127) / 128, heads), dim3(256), (size_t)(64 * 72 + 64 * 96) * 2, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y.data(), oh, nq * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(yl.data(), ol, nq * 2, hipMemcpyDeviceToHost));
    auto f = [](unsigned short b) { unsigned u = (unsigned)b << 16; float v; memcpy(&v, &u, 4); return v; };
    double md = 0, mr = 0; size_t nd = 0;
    for (size_t i = 0; i < nq; i++) { const double p = (double)f(x[i]) + f(xl[i]), q = (double)f(y[i]) + f(yl[i]); md = fmax(md, fabs(p - q)); mr = fmax(mr, fabs(q)); nd += x[i] != y[i]; }
    printf("  dma vs attn_prefill_kernel: max |d| / max |ref| = %.2e, %zu of %zu hi words differ\n", md / mr, nd, nq);
  }
  printf("DMA  hd %3d  S %d: %.1f us\n", HD, S, best * 1e3);
#else
  printf("DIS=%2d  LA %d  KP %d  hd %3d  S %d: %.1f us\n", TGX_ATTN_DIS, PROBE_LA, PROBE_KP, HD, S, best * 1e3);
#endif
  return 0;
}
