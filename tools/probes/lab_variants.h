// lab_variants.h — kernel variants under test in layer_lab.hip (build with -DLAB_VARIANTS).
#pragma once
#include <algorithm>
#include "kernels/oproj_sliced.h"
#include "qkv_attn.h"

// experiment: a plain K-sliced GEMV tile (no merge): workgroup = RB rows x SW columns, partial sums into fixed-point accumulators
struct SlicedArgs { const void* W; int ldw; const float* x; long long* acc; int N; };
template <int LPR, int NL>
__global__ __launch_bounds__(256) void gemv_sliced_kernel(const SlicedArgs a) {
  constexpr int SW = LPR * 8, RPL = 64 / LPR, RPW = NL * RPL, RB = 4 * RPW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int slice = blockIdx.y, row0 = blockIdx.x * RB + wv * RPW;
  const int rl = lane / LPR, cl = lane % LPR;
  Slice8<DT_BF16> w[NL];
  const unsigned short* Wp = static_cast<const unsigned short*>(a.W) + (size_t)slice * SW;
#pragma unroll
  for (int i = 0; i < NL; i++) w[i] = load_slice_nt<DT_BF16>(Wp + (size_t)min(row0 + i * RPL + rl, a.N - 1) * a.ldw, cl);
  const f32x4* xp = reinterpret_cast<const f32x4*>(a.x + (size_t)slice * SW + cl * 8);
  const f32x4 xa = xp[0], xb = xp[1];
  float mine = 0.f;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    float s = dot8<DT_BF16>(0.f, w[i], xa, xb);
    if constexpr (LPR == 64) s = wave_sum(s);
    else { s = row_group_sum<16>(s); if constexpr (LPR == 32) s += __shfl_xor(s, 16, 64); }
#pragma unroll
    for (int r = 0; r < RPL; r++) { const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), r * LPR)); if (lane == i * RPL + r) mine = t; }
  }
  if (lane < RPW && row0 + lane < a.N) __hip_atomic_fetch_add(a.acc + row0 + lane, f32_to_fix(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int LPR, int NL>
static void v_sliced(Lab& b, const void* W, int N, int K, const float* x, long long* acc) {
  SlicedArgs a{W, K, x, acc, N};
  constexpr int RB = 4 * NL * (64 / LPR);
  hipLaunchKernelGGL((gemv_sliced_kernel<LPR, NL>), dim3((N + RB - 1) / RB, K / (LPR * 8)), dim3(256), 0, b.st, a);
}

static int g_ops_dbg = 0;
static unsigned* g_epoch = nullptr;       // set: the o_proj launch advances the granule tag of kernels/qkv_attn.h
static void v_oproj_sliced(Lab& b, int l, long long* acc, const float* resid) {
  const LayerBuf& w = b.lb[(size_t)l];
  OprojSlicedArgs a{};
  a.W = w.wo; a.ldw = b.qd; a.part = b.part; a.nsplit = b.nsplit; a.x = resid; a.acc = acc; a.H = b.H; a.dbg = g_ops_dbg; a.epoch = g_epoch;
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
                case 6: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO_RMSNORM, EPI_SILU_MUL, 6, 1, true>), dim3(gr), dim3(256), (size_t)b.H * 4, b.st, u); break;
                case 8: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO_RMSNORM, EPI_SILU_MUL, 8, 1, true>), dim3(gr), dim3(256), (size_t)b.H * 4, b.st, u); break;
                default: printf("lab: nx %d not instantiated\n", nx); }
}
static void v_down_acc(Lab& b, int l, long long* acc, float* resid) {
  const LayerBuf& w = b.lb[(size_t)l];
  GemvArgs d{};
  d.W = w.wdown; d.x = b.h; d.N = b.H; d.K = b.I; d.ldw = b.I; d.units = b.H / 2; d.ks = 4; d.out = resid; d.res_acc = acc; d.hd = 2;
  const int nx = b.nx_of(b.I, 4), gr = b.grid_of(d.units, 4);
  switch (nx) { case 3: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO_PLAIN, EPI_RESIDUAL, 3, 1, true>), dim3(gr), dim3(256), 0, b.st, d); break;
                case 4: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO_PLAIN, EPI_RESIDUAL, 4, 1, true>), dim3(gr), dim3(256), 0, b.st, d); break;
                case 7: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO_PLAIN, EPI_RESIDUAL, 7, 1, true>), dim3(gr), dim3(256), 0, b.st, d); break;
                default: printf("lab: nx %d not instantiated\n", nx); }
}

__global__ void acc_to_f32(const long long* acc, float* out, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = fix_to_f32(acc[i]); }

template <int G, int NW, int UNR>
static void v_attn(Lab& b, int l) {
  AttnArgs a = attn_args(b, l);
  const int ngroups = (a.gfull + G - 1) / G;
  const dim3 grid(a.kv_heads * a.nsplit, 1, ngroups), blk(64 * NW);
  hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 64, G, NW, false, false, UNR>), grid, blk, 0, b.st, a);
}

// direct forms (short contexts): one workgroup per (kv head, query head), normalised output, no records
template <int NW>
static void v_attn_direct(Lab& b, int l) {
  AttnArgs a = attn_args(b, l);
  a.direct = 1;
  const dim3 grid(a.kv_heads, 1, a.gfull), blk(64 * NW);
  if (b.g.hd == 64) hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 64, 1, NW>), grid, blk, 0, b.st, a);
  else hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 128, 1, NW>), grid, blk, 0, b.st, a);
}

// direct form with the o_proj product in its epilogue (AttnArgs.oj_*): {attention, o_proj} as ONE launch at short contexts
template <int NW, int UNR = 4>
static void v_attn_oproj(Lab& b, int l, long long* acc, const float* resid, int rsplit) {
  AttnArgs a = attn_args(b, l);
  a.direct = 1;
  a.oj_w = b.lb[(size_t)l].wo; a.oj_x = resid; a.oj_acc = acc; a.oj_H = b.H; a.oj_ldw = b.qd; a.oj_rsplit = rsplit;
  const dim3 grid(a.kv_heads, 1, a.gfull * rsplit), blk(64 * NW);
  if (b.g.hd == 64) hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 64, 1, NW, false, false, UNR, true>), grid, blk, 0, b.st, a);
  else hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 128, 1, NW, false, false, 4, true>), grid, blk, 0, b.st, a);
}

static void lab_fused_short(Lab& b) {
  long long* acc; CK(hipMalloc(&acc, (size_t)b.H * 8));
  float *x_a, *x_b; CK(hipMalloc(&x_a, (size_t)b.H * 4)); CK(hipMalloc(&x_b, (size_t)b.H * 4));
  std::vector<float> ha((size_t)b.H), hb((size_t)b.H);
  const float tp = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { v_attn_direct<16>(b, l); p_oproj(b, l, b.scratch_x); } }, b.L);
  const float tp4 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { v_attn_direct<4>(b, l); p_oproj(b, l, b.scratch_x); } }, b.L);
  printf("context %d: {attn direct, o_proj} as two launches: 16 waves %.2f us, 4 waves %.2f\n", b.pos_h, tp, tp4);
  const float l5 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_qkv(b, l, nullptr); v_attn_direct<16>(b, l); p_oproj(b, l, b.scratch_x); p_gateup(b, l, nullptr); p_down(b, l, b.scratch_x); } }, b.L);
  const float l5b = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_qkv(b, l, nullptr); v_attn_direct<4>(b, l); p_oproj(b, l, b.scratch_x); p_gateup(b, l, nullptr); p_down(b, l, b.scratch_x); } }, b.L);
  printf("  layer as 5 launches {qkv, attn direct, o_proj, gate_up, down}: 16 waves %.2f us, 4 waves %.2f\n", l5, l5b);
  {
    CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
    const float ls = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_qkv(b, l, nullptr); p_attn_only(b, l, nullptr); v_oproj_sliced(b, l, acc, b.x); v_gateup_acc(b, l, acc); v_down_acc(b, l, acc, b.scratch_x); } }, b.L);
    printf("  layer as 5 launches {qkv, attn split form, o_proj sliced, gate_up, down}: %.2f us\n", ls);
  }
  for (int rsplit : {2, 4, 8}) {
    if ((b.H % rsplit) || (b.H / rsplit) % 8) continue;
    CK(hipMemcpyAsync(x_a, b.x, (size_t)b.H * 4, hipMemcpyDeviceToDevice, b.st));
    CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
    v_attn_direct<16>(b, 3); p_oproj(b, 3, x_a);
    v_attn_oproj<4>(b, 3, acc, b.x, rsplit);
    hipLaunchKernelGGL(acc_to_f32, dim3((b.H + 255) / 256), dim3(256), 0, b.st, (const long long*)acc, x_b, b.H);
    CK(hipMemcpyAsync(ha.data(), x_a, (size_t)b.H * 4, hipMemcpyDeviceToHost, b.st)); CK(hipMemcpyAsync(hb.data(), x_b, (size_t)b.H * 4, hipMemcpyDeviceToHost, b.st));
    CK(hipStreamSynchronize(b.st));
    double mx = 0, ref = 0;
    for (int i = 0; i < b.H; i++) { mx = std::max(mx, (double)fabsf(ha[(size_t)i] - hb[(size_t)i])); ref = std::max(ref, (double)fabsf(ha[(size_t)i])); }
    CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
    const float t16 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_attn_oproj<16>(b, l, acc, b.x, rsplit); }, b.L);
    const float t8 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_attn_oproj<8>(b, l, acc, b.x, rsplit); }, b.L);
    const float t4 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_attn_oproj<4>(b, l, acc, b.x, rsplit); }, b.L);
    const float t4u8 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_attn_oproj<4, 8>(b, l, acc, b.x, rsplit); }, b.L);
    const float t4u2 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_attn_oproj<4, 2>(b, l, acc, b.x, rsplit); }, b.L);
    const float t8u2 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_attn_oproj<8, 2>(b, l, acc, b.x, rsplit); }, b.L);
    printf("    (4 waves x 8 wave-loads per block %.2f, x 2: %.2f; 8 waves x 2: %.2f)\n", t4u8, t4u2, t8u2);
    CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
    const float l8 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_qkv(b, l, nullptr); v_attn_oproj<8>(b, l, acc, b.x, rsplit); v_gateup_acc(b, l, acc); v_down_acc(b, l, acc, b.scratch_x); } }, b.L);
    CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
    const float l4 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_qkv(b, l, nullptr); v_attn_oproj<4>(b, l, acc, b.x, rsplit); v_gateup_acc(b, l, acc); v_down_acc(b, l, acc, b.scratch_x); } }, b.L);
    printf("  attn + o_proj in one launch, %2d workgroups per head: 16 waves %.2f us, 8 waves %.2f, 4 waves %.2f (rel diff %.2g); layer as 4 launches: 8 waves %.2f, 4 waves %.2f\n", rsplit, t16, t8, t4, mx / ref, l8, l4);
  }
  CK(hipFree(acc)); CK(hipFree(x_a)); CK(hipFree(x_b));
}


// ---- experiment: gate_up + down as ONE launch (small models).  Workgroup w owns the intermediate elements i in [w IPW, (w + 1) IPW): it computes
// h_i = silu(gate_i . x') * (up_i . x') (rows i and I + i of W_gu) and adds h_i * W_down[:, i] into its H column sums — W_down is read TRANSPOSED
// ([I][H]: column i is a contiguous row), so both products stream.  No cross-workgroup dependency: the H partial sums per workgroup go to the fixed-point
// residual accumulators with atomics.
struct MlpFusedArgs { const unsigned short *wgu, *wdT, *norm_w; const float* x; long long* acc; int H, I, ipw; float eps; };
template <int NX, int NWV>     // NX: 16-byte slices of an H-vector per lane (H <= 512 NX); NWV: waves per workgroup
__global__ __launch_bounds__(64 * NWV) void mlp_fused_kernel(const MlpFusedArgs a) {
  extern __shared__ float red_[];
  float (*red)[NX * 512] = reinterpret_cast<float (*)[NX * 512]>(red_);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nsl = a.H / 8;                                      // slices per row
  int cidx[NX]; bool cok[NX];
#pragma unroll
  for (int j = 0; j < NX; j++) { cidx[j] = min(lane + 64 * j, nsl - 1); cok[j] = lane + 64 * j < nsl; }
  const int i0 = blockIdx.x * a.ipw, i1 = min(i0 + a.ipw, a.I);
  // first unit's weights leave before the norm
  Slice8<DT_BF16> wg[NX], wu[NX], wd[NX], ng[NX], nu[NX], nd[NX];
  auto load_unit = [&](int i, Slice8<DT_BF16>* g, Slice8<DT_BF16>* u, Slice8<DT_BF16>* d) {
    const int ic = min(i, a.I - 1);
#pragma unroll
    for (int j = 0; j < NX; j++) {
      g[j] = load_slice_nt<DT_BF16>(a.wgu + (size_t)ic * a.H, cidx[j]);
      u[j] = load_slice_nt<DT_BF16>(a.wgu + (size_t)(a.I + ic) * a.H, cidx[j]);
      d[j] = load_slice_nt<DT_BF16>(a.wdT + (size_t)ic * a.H, cidx[j]);
    }
  };
  load_unit(i0 + wv, wg, wu, wd);
  float xr[NX][8];
  Slice8<DT_BF16> nw[NX];
#pragma unroll
  for (int j = 0; j < NX; j++) {
    const f32x4* xp = reinterpret_cast<const f32x4*>(a.x + cidx[j] * 8);
    const f32x4 x0 = xp[0], x1 = xp[1];
#pragma unroll
    for (int t = 0; t < 4; t++) { xr[j][t] = cok[j] ? x0[t] : 0.f; xr[j][4 + t] = cok[j] ? x1[t] : 0.f; }
    nw[j] = load_slice<DT_BF16>(a.norm_w, cidx[j]);
  }
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NX; j++)
#pragma unroll
    for (int t = 0; t < 8; t++) ss = fmaf(xr[j][t], xr[j][t], ss);
  ss = wave_sum(ss);
  const float inv = 1.0f / sqrtf(ss / (float)a.H + a.eps);
#pragma unroll
  for (int j = 0; j < NX; j++) {
    float w[8];
    slice_unpack<DT_BF16>(nw[j], w);
#pragma unroll
    for (int t = 0; t < 8; t++) xr[j][t] = w[t] * (xr[j][t] * inv);
  }
  float out[NX][8];
#pragma unroll
  for (int j = 0; j < NX; j++)
#pragma unroll
    for (int t = 0; t < 8; t++) out[j][t] = 0.f;
  for (int i = i0 + wv; i < i1; i += NWV) {
    const bool more = i + NWV < i1;
    if (more) load_unit(i + NWV, ng, nu, nd);
    float g = 0.f, u = 0.f;
#pragma unroll
    for (int j = 0; j < NX; j++) {
      float fg[8], fu[8];
      slice_unpack<DT_BF16>(wg[j], fg); slice_unpack<DT_BF16>(wu[j], fu);
#pragma unroll
      for (int t = 0; t < 8; t++) { g = fmaf(fg[t], xr[j][t], g); u = fmaf(fu[t], xr[j][t], u); }
    }
    g = wave_sum(g); u = wave_sum(u);
    const float h = (g / (1.0f + expf(-g))) * u;
#pragma unroll
    for (int j = 0; j < NX; j++) {
      float fd[8];
      slice_unpack<DT_BF16>(wd[j], fd);
#pragma unroll
      for (int t = 0; t < 8; t++) out[j][t] = fmaf(h, fd[t], out[j][t]);
    }
    if (more) {
#pragma unroll
      for (int j = 0; j < NX; j++) { wg[j] = ng[j]; wu[j] = nu[j]; wd[j] = nd[j]; }
    }
  }
  // the four waves' column sums meet in LDS, then one fixed-point add per column
#pragma unroll
  for (int j = 0; j < NX; j++) {
    f32x4* dst = reinterpret_cast<f32x4*>(&red[wv][(lane + 64 * j) * 8]);
    dst[0] = f32x4{out[j][0], out[j][1], out[j][2], out[j][3]}; dst[1] = f32x4{out[j][4], out[j][5], out[j][6], out[j][7]};
  }
  __syncthreads();
  for (int c = threadIdx.x; c < a.H; c += 64 * NWV) {
    float v = red[0][c];
#pragma unroll
    for (int w = 1; w < NWV; w++) v += red[w][c];
    long long f = f32_to_fix(v);
    if (blockIdx.x == 0) f += f32_to_fix(a.x[c]);          // the residual, once
    __hip_atomic_fetch_add(a.acc + c, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void transpose_bf16(const unsigned short* src, unsigned short* dst, int rows, int cols) {      // dst[c][r] = src[r][c]
  const size_t n = (size_t)rows * cols;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t r = idx / cols, c = idx - r * cols;
    dst[c * rows + r] = src[idx];
  }
}
template <int NWV>
static void v_mlp_fused(Lab& b, int l, const unsigned short* wdT, long long* acc, int nwg) {
  const LayerBuf& w = b.lb[(size_t)l];
  MlpFusedArgs a{w.wgu, wdT, w.post_norm, b.x, acc, b.H, b.I, (b.I + nwg - 1) / nwg, b.eps};
  if (b.H <= 1024) {
    const size_t lds = (size_t)NWV * 2 * 512 * 4;
    static bool set = false; if (!set) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<2, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set = true; }
    hipLaunchKernelGGL((mlp_fused_kernel<2, NWV>), dim3(nwg), dim3(64 * NWV), lds, b.st, a);
  } else {
    constexpr int W = NWV > 8 ? 8 : NWV;
    const size_t lds = (size_t)W * 4 * 512 * 4;
    static bool set = false; if (!set) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_kernel<4, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set = true; }
    hipLaunchKernelGGL((mlp_fused_kernel<4, W>), dim3(nwg), dim3(64 * W), lds, b.st, a);
  }
}
static void lab_mlp_fused(Lab& b) {
  if (b.H > 2048) return;
  std::vector<unsigned short*> wdT((size_t)b.L);
  for (int l = 0; l < b.L; l++) {
    CK(hipMalloc(&wdT[(size_t)l], (size_t)b.H * b.I * 2));
    hipLaunchKernelGGL(transpose_bf16, dim3(1024), dim3(256), 0, b.st, (const unsigned short*)b.lb[(size_t)l].wdown, wdT[(size_t)l], b.H, b.I);
  }
  long long* acc; CK(hipMalloc(&acc, (size_t)b.H * 8));
  float *x_a, *x_b; CK(hipMalloc(&x_a, (size_t)b.H * 4)); CK(hipMalloc(&x_b, (size_t)b.H * 4));
  std::vector<float> ha((size_t)b.H), hb((size_t)b.H);
  CK(hipMemcpyAsync(x_a, b.x, (size_t)b.H * 4, hipMemcpyDeviceToDevice, b.st));
  p_gateup(b, 3, nullptr); p_down(b, 3, x_a);
  const float tp = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_gateup(b, l, nullptr); p_down(b, l, b.scratch_x); } }, b.L);
  printf("{gate_up, down} as two launches: %.2f us\n", tp);
  for (int nwg : {64, 128, 192, 256, 512}) {
    CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
    v_mlp_fused<4>(b, 3, wdT[3], acc, nwg);
    hipLaunchKernelGGL(acc_to_f32, dim3((b.H + 255) / 256), dim3(256), 0, b.st, (const long long*)acc, x_b, b.H);
    CK(hipMemcpyAsync(ha.data(), x_a, (size_t)b.H * 4, hipMemcpyDeviceToHost, b.st)); CK(hipMemcpyAsync(hb.data(), x_b, (size_t)b.H * 4, hipMemcpyDeviceToHost, b.st));
    CK(hipStreamSynchronize(b.st));
    double mx = 0, ref = 0;
    for (int i = 0; i < b.H; i++) { mx = std::max(mx, (double)fabsf(ha[(size_t)i] - hb[(size_t)i])); ref = std::max(ref, (double)fabsf(ha[(size_t)i])); }
    const float t = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_mlp_fused<4>(b, l, wdT[(size_t)l], acc, nwg); }, b.L);
    const float t8 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_mlp_fused<8>(b, l, wdT[(size_t)l], acc, nwg); }, b.L);
    const float t16 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_mlp_fused<16>(b, l, wdT[(size_t)l], acc, nwg); }, b.L);
    printf("  gate_up + down in one launch (W_down transposed, fixed-point column sums), %3d workgroups: 4 waves %.2f us, 8 waves %.2f, 16 waves (8 at hidden > 1024) %.2f   (rel diff vs two launches %.2g)\n", nwg, t, t8, t16, mx / ref);
  }
  for (int l = 0; l < b.L; l++) CK(hipFree(wdT[(size_t)l]));
  CK(hipFree(acc)); CK(hipFree(x_a)); CK(hipFree(x_b));
}

static void v_oproj_sliced16(Lab& b, int l, long long* acc, const float* resid) {
  const LayerBuf& w = b.lb[(size_t)l];
  OprojSlicedArgs a{};
  a.W = w.wo; a.ldw = b.qd; a.part = b.part; a.nsplit = b.nsplit; a.x = resid; a.acc = acc; a.H = b.H;
  const dim3 grid(b.H / oproj_sliced_rows<16>(), b.qd / 128); hipLaunchKernelGGL((oproj_sliced_kernel<DT_BF16, 64, 16>), grid, dim3(256), 0, b.st, a);
}
static void lab_oproj_shapes(Lab& b) {
  long long* acc; CK(hipMalloc(&acc, (size_t)b.H * 8)); CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
  const float t32 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_oproj_sliced(b, l, acc, b.x); }, b.L);
  const float t16 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_oproj_sliced16(b, l, acc, b.x); }, b.L);
  const float p32 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_attn_only(b, l, nullptr); v_oproj_sliced(b, l, acc, b.x); } }, b.L);
  const float p16 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_attn_only(b, l, nullptr); v_oproj_sliced16(b, l, acc, b.x); } }, b.L);
  printf("o_proj sliced alone: 256-column slices (4 heads) %.2f us, 128-column slices (2 heads) %.2f; with the attention launch in front: %.2f / %.2f\n", t32, t16, p32, p16);
  CK(hipFree(acc));
}
// ---- round 6: the QKV product and the split-form attention as ONE launch (kernels/qkv_attn.h) -----------------------------------------------
struct FuseBufs { unsigned long long *gq, *gkv; unsigned* epoch; float* part2; unsigned long long* stamps; };
template <int DEPTH = 4, bool TIMING = false, bool COMMUTE = false>
static void v_qkv_attn(Lab& b, int l, const FuseBufs& f, float* part, int n_prod) {
  const LayerBuf& w = b.lb[(size_t)l];
  QkvAttnArgs A{};
  const int ks = b.H >= 2048 ? 4 : 1;
  GemvArgs& k = A.g;
  k.W = w.wqkv; k.x = b.x; k.norm_w = w.in_norm; k.eps = b.eps; k.N = b.NQ; k.K = b.H; k.ldw = b.H; k.units = b.NQ / 2; k.ks = ks;
  k.k_cache = w.kc; k.v_cache = w.vc; k.rope_cos = b.rc; k.rope_sin = b.rs; k.pos = b.pos;
  k.heads = b.g.heads; k.kv_heads = b.g.kv; k.hd = b.g.hd; k.max_ctx = b.max_ctx;
  A.a = attn_args(b, l); A.a.part = part;
  A.gran_q = f.gq; A.gran_kv = f.gkv; A.epoch = f.epoch; A.n_prod = n_prod; A.stamps = f.stamps;
  const int grid = n_prod + b.g.heads * b.nsplit;
  switch (b.nx_of(b.H, ks)) {
    case 1: hipLaunchKernelGGL((qkv_attn_kernel<DT_BF16, 64, 1, DEPTH, TIMING, COMMUTE>), dim3(grid), dim3(256), 0, b.st, A); break;
    case 2: hipLaunchKernelGGL((qkv_attn_kernel<DT_BF16, 64, 2, DEPTH, TIMING, COMMUTE>), dim3(grid), dim3(256), 0, b.st, A); break;
    default: printf("lab: qkv_attn nx not instantiated\n");
  }
}
static void lab_qkv_attn(Lab& b) {
  if (b.g.hd != 64) { printf("qkv || attention: head_dim 64 only\n"); return; }
  FuseBufs f{};
  CK(hipMalloc(&f.stamps, (size_t)4096 * 64)); CK(hipMemsetAsync(f.stamps, 0, (size_t)4096 * 64, b.st));
  CK(hipMalloc(&f.gq, (size_t)b.qd * 8)); CK(hipMalloc(&f.gkv, (size_t)2 * b.kvd * 8)); CK(hipMalloc(&f.epoch, 4)); CK(hipMalloc(&f.part2, b.part_row * 4));
  CK(hipMemsetAsync(f.gq, 0, (size_t)b.qd * 8, b.st)); CK(hipMemsetAsync(f.gkv, 0, (size_t)2 * b.kvd * 8, b.st)); CK(hipMemsetAsync(f.part2, 0, b.part_row * 4, b.st));
  const unsigned one = 1; CK(hipMemcpyAsync(f.epoch, &one, 4, hipMemcpyHostToDevice, b.st));
  long long* acc; CK(hipMalloc(&acc, (size_t)b.H * 8)); CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
  // ---- values: the two product launches vs the fused launch, records compared bit for bit (layer 3)
  const int cache_row = b.g.kv * b.g.hd;   // one position of every kv head is rewritten by either path with the same values
  (void)cache_row;
  p_qkv(b, 3, nullptr); v_attn<1, 4, 4>(b, 3);
  CK(hipStreamSynchronize(b.st));
  std::vector<float> ra(b.part_row), rb(b.part_row);
  CK(hipMemcpy(ra.data(), b.part, b.part_row * 4, hipMemcpyDeviceToHost));
  for (int np : {256, 512, 768}) {
    CK(hipMemsetAsync(f.part2, 0, b.part_row * 4, b.st));
    v_qkv_attn(b, 3, f, f.part2, np);
    hipLaunchKernelGGL(bump_epoch_kernel, dim3(1), dim3(64), 0, b.st, f.epoch);
    CK(hipStreamSynchronize(b.st));
    CK(hipMemcpy(rb.data(), f.part2, b.part_row * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; const int active = (b.pos_h + 1 + 127) / 128;
    for (int h = 0; h < b.g.heads; h++) for (int s = 0; s < b.nsplit; s++) {
      const float* x = &ra[((size_t)h * b.nsplit + s) * 68]; const float* y = &rb[((size_t)h * b.nsplit + s) * 68];
      if (s >= active) { if (!(y[64] == -INFINITY)) bad++; continue; }
      for (int d = 0; d < 66; d++) if (memcmp(&x[d], &y[d], 4)) bad++;
    }
    printf("qkv || attention, %d producer workgroups: %zu record words differ from {qkv, attention} (0 = bit-identical)\n", np, bad);
  }
  // ---- timeline of one fused launch in the middle of a chain of layers (100 MHz stamps; us from the first workgroup's start)
  {   // the norm scale commuted into the epilogue: same records within rounding
    CK(hipMemsetAsync(f.part2, 0, b.part_row * 4, b.st));
    v_qkv_attn<3, false, true>(b, 3, f, f.part2, 512);
    hipLaunchKernelGGL(bump_epoch_kernel, dim3(1), dim3(64), 0, b.st, f.epoch);
    CK(hipStreamSynchronize(b.st));
    CK(hipMemcpy(rb.data(), f.part2, b.part_row * 4, hipMemcpyDeviceToHost));
    double mx = 0, ref = 0; const int active = (b.pos_h + 1 + 127) / 128;
    for (int h = 0; h < b.g.heads; h++) for (int s = 0; s < active; s++) for (int d = 0; d < 64; d++) {
      const size_t i = ((size_t)h * b.nsplit + s) * 68 + d;
      mx = std::max(mx, (double)fabsf(ra[i] - rb[i])); ref = std::max(ref, (double)fabsf(ra[i]));
    }
    printf("qkv || attention with the RMSNorm scale commuted into the epilogue: records differ by %.3g of max |o| %.3g\n", mx / ref, ref);
  }
  auto timeline = [&](int np, int depth) {
    for (int l = 0; l < 8; l++) {
      if (l == 6) { switch (depth) { case 2: v_qkv_attn<2, true>(b, l, f, f.part2, np); break; case 3: v_qkv_attn<3, true>(b, l, f, f.part2, np); break; default: v_qkv_attn<4, true>(b, l, f, f.part2, np); } }
      else v_qkv_attn<2>(b, l, f, f.part2, np);
      g_epoch = f.epoch; v_oproj_sliced(b, l, acc, b.x); g_epoch = nullptr;
      v_gateup_acc(b, l, acc); v_down_acc(b, l, acc, b.scratch_x);
    }
    CK(hipStreamSynchronize(b.st));
    const int nwg = np + b.g.heads * b.nsplit;
    std::vector<unsigned long long> st((size_t)nwg * 8);
    CK(hipMemcpy(st.data(), f.stamps, st.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemset(f.stamps, 0, (size_t)4096 * 64));
    unsigned long long t0 = ~0ull; for (int w = 0; w < nwg; w++) t0 = std::min(t0, st[(size_t)w * 8]);
    auto stat = [&](int lo, int hi, int idx, const char* what) {
      std::vector<double> v; for (int w = lo; w < hi; w++) if (st[(size_t)w * 8 + idx]) v.push_back((double)(st[(size_t)w * 8 + idx] - t0) / 100.0);
      if (v.empty()) return; std::sort(v.begin(), v.end());
      printf("    %-34s n %4zu  min %5.2f  median %5.2f  p90 %5.2f  max %5.2f us\n", what, v.size(), v[0], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    };
    const int active = ((b.pos_h + 1 + 127) / 128) * b.g.heads;
    printf("  timeline, %d producers (depth %d), %d consumers holding keys:\n", np, depth, active);
    stat(0, np, 0, "producer start"); stat(0, np, 1, "producer x normalised"); stat(0, np, 2, "producer dots done (last batch)"); stat(0, np, 3, "producer published (last batch)"); stat(0, np, 7, "producer end");
    stat(np, np + active, 0, "consumer start"); stat(np, np + active, 1, "consumer position read"); stat(np, np + active, 2, "consumer q arrived"); stat(np, np + active, 3, "consumer keys done"); stat(np, np + active, 7, "consumer end");
    stat(np + active, nwg, 7, "empty-split consumer end");
  };
  timeline(512, 3);
  // ---- time: the pair, and the layer of five / four launches
  const float t_q = time_graph(b, [&] { for (int l = 0; l < b.L; l++) p_qkv(b, l, nullptr); }, b.L);
  const float t_a = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_attn<1, 4, 4>(b, l); }, b.L);
  const float t_pair = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_qkv(b, l, nullptr); v_attn<1, 4, 4>(b, l); } }, b.L);
  printf("product: qkv %.2f us, attention (1 head per workgroup) %.2f us, the pair back to back %.2f us\n", t_q, t_a, t_pair);
  const float t_bump = time_graph(b, [&] { for (int l = 0; l < b.L; l++) hipLaunchKernelGGL(bump_epoch_kernel, dim3(1), dim3(64), 0, b.st, f.epoch); }, b.L);
  for (int np : {256, 384, 448, 512, 768, 1024}) {
    const float t2 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { v_qkv_attn<2>(b, l, f, f.part2, np); hipLaunchKernelGGL(bump_epoch_kernel, dim3(1), dim3(64), 0, b.st, f.epoch); } }, b.L);
    const float t4 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { v_qkv_attn<4>(b, l, f, f.part2, np); hipLaunchKernelGGL(bump_epoch_kernel, dim3(1), dim3(64), 0, b.st, f.epoch); } }, b.L);
    const float t6 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { v_qkv_attn<6>(b, l, f, f.part2, np); hipLaunchKernelGGL(bump_epoch_kernel, dim3(1), dim3(64), 0, b.st, f.epoch); } }, b.L);
    printf("  fused launch (minus the %.2f us epoch-bump launch behind it), %4d producers: depth 2 %.2f us, depth 4 %.2f, depth 6 %.2f\n", t_bump, np, t2 - t_bump, t4 - t_bump, t6 - t_bump);
  }
  g_epoch = nullptr;
  CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
  const float l5 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { p_qkv(b, l, nullptr); v_attn<1, 4, 4>(b, l); v_oproj_sliced(b, l, acc, b.x); v_gateup_acc(b, l, acc); v_down_acc(b, l, acc, b.scratch_x); } }, b.L);
  g_epoch = f.epoch;
  for (int np : {384, 448, 512, 640}) {
    CK(hipMemsetAsync(acc, 0, (size_t)b.H * 8, b.st));
    const float l3 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { v_qkv_attn<3>(b, l, f, b.part, np); v_oproj_sliced(b, l, acc, b.x); v_gateup_acc(b, l, acc); v_down_acc(b, l, acc, b.scratch_x); } }, b.L);
    const float l4 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { v_qkv_attn<4>(b, l, f, b.part, np); v_oproj_sliced(b, l, acc, b.x); v_gateup_acc(b, l, acc); v_down_acc(b, l, acc, b.scratch_x); } }, b.L);
    const float c3 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { v_qkv_attn<3, false, true>(b, l, f, b.part, np); v_oproj_sliced(b, l, acc, b.x); v_gateup_acc(b, l, acc); v_down_acc(b, l, acc, b.scratch_x); } }, b.L);
    const float c4 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) { v_qkv_attn<4, false, true>(b, l, f, b.part, np); v_oproj_sliced(b, l, acc, b.x); v_gateup_acc(b, l, acc); v_down_acc(b, l, acc, b.scratch_x); } }, b.L);
    printf("layer: {qkv, attention, o_proj sliced, gate_up, down} %.2f us   {qkv || attention (%d producers), o_proj sliced, gate_up, down} depth 3 %.2f, depth 4 %.2f; norm scale commuted: %.2f, %.2f us\n", l5, np, l3, l4, c3, c4);
  }
  g_epoch = nullptr;
  CK(hipFree(f.gq)); CK(hipFree(f.gkv)); CK(hipFree(f.epoch)); CK(hipFree(f.part2)); CK(hipFree(acc));
}

static void lab_variants_main(Lab& b) {
  if (getenv("LAB_FUSE")) { lab_qkv_attn(b); return; }
  if (getenv("LAB_MLP")) { lab_mlp_fused(b); return; }
  if (getenv("LAB_OPROJ")) { lab_oproj_shapes(b); return; }
  if (b.pos_h < 1024 || getenv("LAB_SHORT_ONLY")) { lab_fused_short(b); if (getenv("LAB_SHORT_ONLY")) return; }
  if (b.g.hd == 64 && b.pos_h < 1024) {
    const float d4 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_attn_direct<4>(b, l); }, b.L);
    const float d8 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_attn_direct<8>(b, l); }, b.L);
    const float d16 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_attn_direct<16>(b, l); }, b.L);
    // numerics: 16 waves vs 4 waves on layer 2
    std::vector<float> oa((size_t)b.qd), ob((size_t)b.qd);
    v_attn_direct<4>(b, 2); CK(hipStreamSynchronize(b.st)); CK(hipMemcpy(oa.data(), b.attn, (size_t)b.qd * 4, hipMemcpyDeviceToHost));
    v_attn_direct<16>(b, 2); CK(hipStreamSynchronize(b.st)); CK(hipMemcpy(ob.data(), b.attn, (size_t)b.qd * 4, hipMemcpyDeviceToHost));
    double mx = 0, ref = 0; for (int i = 0; i < b.qd; i++) { mx = std::max(mx, (double)fabsf(oa[(size_t)i] - ob[(size_t)i])); ref = std::max(ref, (double)fabsf(oa[(size_t)i])); }
    printf("attention direct form at context %d: 4 waves %.2f us, 8 waves %.2f, 16 waves %.2f   (16 vs 4 waves: rel diff %.2g)\n", b.pos_h, d4, d8, d16, mx / ref);
  }
  if (b.g.hd == 64) {
#define TA(G_, NW_, U_) { const float t = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_attn<G_, NW_, U_>(b, l); }, b.L); printf("  attn split form: %d heads per workgroup, %2d waves x %d wave-loads (%3d keys per workgroup): %.2f us\n", G_, NW_, U_, NW_ * 8 * U_, t); }
    TA(2, 4, 4) TA(1, 4, 4) TA(4, 4, 4) TA(3, 4, 4) TA(2, 4, 2) TA(1, 4, 2) TA(4, 4, 2) TA(2, 4, 8)
#undef TA
  }
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
  {
    const float t1 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_sliced<32, 8>(b, b.lb[(size_t)l].wdown, b.H, b.I, b.h, acc); }, b.L);
    const float t2 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_sliced<64, 8>(b, b.lb[(size_t)l].wdown, b.H, b.I, b.h, acc); }, b.L);
    const float t3 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_sliced<32, 16>(b, b.lb[(size_t)l].wdown, b.H, b.I, b.h, acc); }, b.L);
    const float t4 = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_sliced<64, 16>(b, b.lb[(size_t)l].wdown, b.H, b.I, b.h, acc); }, b.L);
    printf("down as K-sliced tiles + fixed-point atomics: 64 rows x 256 cols %.2f us, 32 x 512 %.2f, 128 x 256 %.2f, 64 x 512 %.2f   (product down %s)\n", t1, t2, t3, t4, "above");
  }
#ifdef LAB_DISSECT
  for (int d : {1, 2, 3, 4, 5, 6, 7}) {
    g_ops_dbg = d;
    const float t = time_graph(b, [&] { for (int l = 0; l < b.L; l++) v_oproj_sliced(b, l, acc, b.x); }, b.L);
    printf("  o_proj sliced alone, dissect %d (1 no atomics, 2 no merge, 4 no weight loads): %.2f us\n", d, t);
  }
  g_ops_dbg = 0;
#endif
}
