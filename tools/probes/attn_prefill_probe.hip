// attn_prefill_probe.hip — where does the time of attn_prefill_kernel go?  The product kernel (kernels/prefill.h) compiled with parts switched off
// at compile time (-DTGX_ATTN_DIS=bits: 1 no LDS staging, 2 no softmax arithmetic, 4 no PV, 8 no QK^T, 16 no tile fetch), timed on the Llama-3.2-1B
// prefill shape (S = 2048, 32 query heads, 8 kv heads, head_dim 64) with random data.  Results are garbage for DIS != 0: timing only.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I tinygpt_amd/csrc -I include -DTGX_ATTN_DIS=<bits> attn_prefill_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels/common.h"
#include "kernels/prefill.h"
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
  const dim3 grid((S + 127) / 128, heads), blk(256);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 12; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((tgx::attn_prefill_kernel<tgx::DT_BF16, PROBE_HD, PROBE_LA>), grid, blk, 0, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2 && ms < best) best = ms;
  }
  printf("DIS=%2d  LA %d  hd %3d  S %d: %.1f us\n", TGX_ATTN_DIS, PROBE_LA, HD, S, best * 1e3);
  return 0;
}
