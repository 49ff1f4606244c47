// lab_variants.h — kernel variants under test in layer_lab.hip (build with -DLAB_VARIANTS).
#pragma once
#include "kernels/oproj_sliced.h"

static int g_ops_dbg = 0;
static void v_oproj_sliced(Lab& b, int l, long long* acc, const float* resid) {
  const LayerBuf& w = b.lb[(size_t)l];
  OprojSlicedArgs a{};
  a.W = w.wo; a.ldw = b.qd; a.part = b.part; a.nsplit = b.nsplit; a.x = resid; a.acc = acc; a.H = b.H; a.dbg = g_ops_dbg;
  if (b.g.hd == 64 && b.qd % 256 == 0) { const dim3 grid(b.H / oproj_sliced_rows<32>(), b.qd / 256); hipLaunchKernelGGL((oproj_sliced_kernel<DT_BF16, 64, 32>), grid, dim3(256), 0, b.st, a); }
  else if (b.g.hd == 64) { const dim3 grid(b.H / oproj_sliced_rows<16>(), b.qd / 128); hipLaunchKernelGGL((oproj_sliced_kernel<DT_BF16, 64, 16>), grid, dim3(256), 0, b.st, a); }
  else if (b.qd % 512 == 0) { const dim3 grid(b.H / oproj_sliced_rows<64>(), b.qd / 512); hipLaunchKernelGGL((oproj_sliced_kernel<DT_BF16, 128, 64>), grid, dim3(256), 0, b.st, a); }
  else { const dim3 grid(b.H / oproj_sliced_rows<32>(), b.qd / 256); hipLaunchKernelGGL((oproj_sliced_kernel<DT_BF16, 128, 32>), grid, dim3(256), 0, b.st, a); }
}

static void v_gateup_acc(Lab& b, int l, long long* acc) {
  const LayerBuf& w = b.lb[(size_t)l];
  GemvArgs u{};
  u.W = w.wgu; u.x = b.x; u.x_acc = acc; u.norm_w = w.post_norm; u.eps = b.eps; u.N = 2 * b.I; u.K = b.H; u.ldw = b.H; u.units = b.I; u.ks = 1; u.out = b.h; u.hd = 2;
  const int nx = b.nx_of(b.H, 1), gr = b.grid_of(u.units, 1);
  switch (nx) { case 2: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO_RMSNORM, EPI_SILU_MUL, 2, 1, true>), dim3(gr), dim3(256), (size_t)b.H * 4, b.st, u); break;
                case 4: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO_RMSNORM, EPI_SILU_MUL, 4, 1, true>), dim3(gr), dim3(256), (size_t)b.H * 4, b.st, u); break;
                default: printf("lab: nx %d not instantiated\n", nx); }
}
static void v_down_acc(Lab& b, int l, long long* acc, float* resid) {
  const LayerBuf& w = b.lb[(size_t)l];
  GemvArgs d{};
  d.W = w.wdown; d.x = b.h; d.N = b.H; d.K = b.I; d.ldw = b.I; d.units = b.H / 2; d.ks = 4; d.out = resid; d.res_acc = acc; d.hd = 2;
  const int nx = b.nx_of(b.I, 4), gr = b.grid_of(d.units, 4);
  switch (nx) { case 3: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO_PLAIN, EPI_RESIDUAL, 3, 1, true>), dim3(gr), dim3(256), 0, b.st, d); break;
                case 4: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO_PLAIN, EPI_RESIDUAL, 4, 1, true>), dim3(gr), dim3(256), 0, b.st, d); break;
                default: printf("lab: nx %d not instantiated\n", nx); }
}

__global__ void acc_to_f32(const long long* acc, float* out, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = fix_to_f32(acc[i]); }

static void lab_variants_main(Lab& b) {
  long long* acc; CK(hipMalloc(&acc, (size_t)b.H * 8));
  float *x_a, *x_b; CK(hipMalloc(&x_a, (size_t)b.H * 4)); CK(hipMalloc(&x_b, (size_t)b.H * 4));
  // numerics: one layer, product chain vs sliced chain from the same residual
  CK(hipMemcpyAsync(x_a, b.x, (size_t)b.H * 4, hipMemcpyDeviceToDevice, b.st));
  CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
  p_attn_only(b, 3, nullptr); p_combine(b, 3, nullptr); p_oproj(b, 3, x_a);
  v_oproj_sliced(b, 3, acc, b.x);
  hipLaunchKernelGGL(acc_to_f32, dim3((b.H + 255) / 256), dim3(256), 0, b.st, (const long long*)acc, x_b, b.H);
  std::vector<float> ha((size_t)b.H), hb((size_t)b.H);
  CK(hipMemcpyAsync(ha.data(), x_a, (size_t)b.H * 4, hipMemcpyDeviceToHost, b.st)); CK(hipMemcpyAsync(hb.data(), x_b, (size_t)b.H * 4, hipMemcpyDeviceToHost, b.st));
  CK(hipStreamSynchronize(b.st));
  double mx = 0, ref = 0;
  for (int i = 0; i < b.H; i++) { mx = std::max(mx, (double)fabsf(ha[(size_t)i] - hb[(size_t)i])); ref = std::max(ref, (double)fabsf(ha[(size_t)i])); }
  printf("o_proj sliced vs {combine, o_proj}: max |x| %.4g, rel diff %.3g\n", ref, mx / ref);
  const float ta = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_attn_only(b, l, nullptr); p_combine(b, l, nullptr); p_oproj(b, l, b.scratch_x); } }, b.L);
  const float tb = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_attn_only(b, l, nullptr); v_oproj_sliced(b, l, acc, b.x); } }, b.L);
  const float tc = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_oproj_sliced(b, l, acc, b.x); }, b.L);
  printf("{attn, combine, o_proj} %.2f us   {attn, o_proj sliced + merge + fixed-point atomics} %.2f us   (o_proj sliced alone, back to back: %.2f)\n", ta, tb, tc);
  CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
  const float tg = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_gateup_acc(b, l, acc); }, b.L);
  const float td = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_down_acc(b, l, acc, b.scratch_x); }, b.L);
  CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
  const float tl = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_qkv(b, l, nullptr); p_attn_only(b, l, nullptr); v_oproj_sliced(b, l, acc, b.x); v_gateup_acc(b, l, acc); v_down_acc(b, l, acc, b.scratch_x); } }, b.L);
  printf("gate_up reading the fixed-point residual %.2f us, down adding to it %.2f us; the layer as 5 launches {qkv, attn, o_proj sliced, gate_up, down}: %.2f us per layer\n", tg, td, tl);
#ifdef LAB_DISSECT
  for (int d : {1, 2, 3, 4, 5, 6, 7}) {
    g_ops_dbg = d;
    const float t = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_oproj_sliced(b, l, acc, b.x); }, b.L);
    printf("  o_proj sliced alone, dissect %d (1 no atomics, 2 no merge, 4 no weight loads): %.2f us\n", d, t);
  }
  g_ops_dbg = 0;
#endif
}
