// engine_probe.hip — the product's persistent decode engine (kernels/engine.h) against the product's GEMV launches (kernels/gemv.h) on the
// Llama-3.2-1B layer geometry: same weights, same inputs; compares x / q / cache rows and times both as hipGraphs over L layers.
//   per layer, launches:  o_proj(+res) -> gate_up(norm, siluMul) -> down(+res) -> qkv of the next layer (norm, RoPE, cache append)
//   per layer, engine:    ONE launch with the same four ops
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../tinygpt_amd/csrc engine_probe.hip -o build/engine_probe
// Run:   engine_probe [layers=16] [ns=7] [thin=0] [stats=0] [geom=1b|3b|7b] [depth=3]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels/engine.h"
#include "kernels/gemv.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace tgx;

static unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    const float f = ((float)(x & 0xffff) / 32768.0f - 1.0f) * scale;
    p[i] = f32_to_bf16(f);
  }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale, float bias) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    p[i] = ((float)(x & 0xffff) / 32768.0f - 1.0f) * scale + bias;
  }
}
__global__ void reduce8_test(float* out) {
  const int lane = threadIdx.x;
  float v[8];
  for (int r = 0; r < 8; r++) v[r] = (float)((lane * 7 + r * 13) % 31) + 0.25f * r;
  out[lane] = eng_reduce8(v, lane);
  out[64 + lane] = (float)eng_reduce8_index(lane);
}

struct Geom { int H, I, heads, kv, hd; };

struct LayerBuf { unsigned short *wo, *wgu, *wdown, *wqkv, *post_norm, *in_norm; unsigned short *kc_ref, *vc_ref, *kc_eng, *vc_eng; };

static double maxrel(const std::vector<float>& a, const std::vector<float>& b, double* maxabs) {
  double mx = 0, ref = 0;
  for (size_t i = 0; i < a.size(); i++) { mx = std::max(mx, (double)fabsf(a[i] - b[i])); ref = std::max(ref, (double)fabsf(a[i])); }
  if (maxabs) *maxabs = ref;
  return ref > 0 ? mx / ref : mx;
}

int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 16;
  const int ns = argc > 2 ? atoi(argv[2]) : 7;
  const int thin = argc > 3 ? atoi(argv[3]) : 0;
  const int stats = argc > 4 ? atoi(argv[4]) : 0;
  const char* gname = argc > 5 ? argv[5] : "1b";
  const int depth = argc > 6 ? atoi(argv[6]) : 3;
  const int mode = argc > 7 ? atoi(argv[7]) : 0;     // 0: engine = o_proj, gate_up, down, qkv;  1: engine = gate_up, down (o_proj and qkv stay launches)
  Geom g{2048, 8192, 32, 8, 64};
  if (!strcmp(gname, "3b")) g = Geom{3072, 8192, 24, 8, 128};
  if (!strcmp(gname, "7b")) g = Geom{4096, 14336, 32, 8, 128};
  const int H = g.H, I = g.I, qd = g.heads * g.hd, kvd = g.kv * g.hd, NQ = qd + 2 * kvd, half = g.hd / 2;
  const int max_ctx = 4096, pos_h = 2064;
  const float eps = 1e-5f;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount;
  printf("device %s, %d CUs; geometry %s H=%d I=%d heads=%d kv=%d hd=%d; layers %d, ring %d slots, thin %d, depth %d\n", prop.name, G, gname, H, I, g.heads, g.kv, g.hd, L, ns, thin, depth);
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

  {   // unit check of the 8-value reduction
    float* d; CK(hipMalloc(&d, 128 * 4));
    hipLaunchKernelGGL(reduce8_test, dim3(1), dim3(64), 0, st, d);
    std::vector<float> o(128); CK(hipMemcpyAsync(o.data(), d, 512, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    int bad = 0;
    for (int lane = 0; lane < 64; lane++) {
      const int idx = (int)o[64 + lane];
      float ref = 0; for (int l2 = 0; l2 < 64; l2++) ref += (float)((l2 * 7 + idx * 13) % 31) + 0.25f * idx;
      if (fabsf(ref - o[lane]) > 1e-3f) { if (bad < 4) printf("reduce8 lane %d idx %d got %f want %f\n", lane, idx, o[lane], ref); bad++; }
    }
    printf("reduce8 unit check: %s\n", bad ? "FAILED" : "ok");
    if (bad) return 1;
    CK(hipFree(d));
  }

  std::vector<LayerBuf> lb((size_t)L + 1);
  const size_t cache_elems = (size_t)g.kv * max_ctx * g.hd;
  for (int l = 0; l <= L; l++) {
    LayerBuf& b = lb[(size_t)l];
    CK(hipMalloc(&b.wo, (size_t)H * qd * 2)); CK(hipMalloc(&b.wgu, (size_t)2 * I * H * 2)); CK(hipMalloc(&b.wdown, (size_t)H * I * 2));
    CK(hipMalloc(&b.wqkv, (size_t)NQ * H * 2)); CK(hipMalloc(&b.post_norm, (size_t)H * 2)); CK(hipMalloc(&b.in_norm, (size_t)H * 2));
    CK(hipMalloc(&b.kc_ref, cache_elems * 2)); CK(hipMalloc(&b.vc_ref, cache_elems * 2)); CK(hipMalloc(&b.kc_eng, cache_elems * 2)); CK(hipMalloc(&b.vc_eng, cache_elems * 2));
    CK(hipMemsetAsync(b.kc_ref, 0, cache_elems * 2, st)); CK(hipMemsetAsync(b.vc_ref, 0, cache_elems * 2, st));
    CK(hipMemsetAsync(b.kc_eng, 0, cache_elems * 2, st)); CK(hipMemsetAsync(b.vc_eng, 0, cache_elems * 2, st));
    const unsigned s0 = 1000u * (unsigned)l;
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, b.wo, (size_t)H * qd, s0 + 1, 0.02f);
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, b.wgu, (size_t)2 * I * H, s0 + 2, 0.03f);
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, b.wdown, (size_t)H * I, s0 + 3, 0.02f);
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, b.wqkv, (size_t)NQ * H, s0 + 4, 0.03f);
    hipLaunchKernelGGL(fill_bf16, dim3(8), dim3(256), 0, st, b.post_norm, (size_t)H, s0 + 5, 0.5f);
    hipLaunchKernelGGL(fill_bf16, dim3(8), dim3(256), 0, st, b.in_norm, (size_t)H, s0 + 6, 0.5f);
  }
  float *attn, *x0, *x_ref, *x_eng, *h_ref, *q_ref, *q_eng, *rc, *rs;
  int* pos; unsigned *epoch, *err; u64 *g_x1, *g_h, *g_x2; unsigned long long* stats_d;
  CK(hipMalloc(&attn, (size_t)qd * 4)); CK(hipMalloc(&x0, (size_t)H * 4)); CK(hipMalloc(&x_ref, (size_t)H * 4)); CK(hipMalloc(&x_eng, (size_t)H * 4));
  CK(hipMalloc(&h_ref, (size_t)I * 4)); CK(hipMalloc(&q_ref, (size_t)qd * 4)); CK(hipMalloc(&q_eng, (size_t)qd * 4));
  CK(hipMalloc(&rc, (size_t)max_ctx * half * 4)); CK(hipMalloc(&rs, (size_t)max_ctx * half * 4));
  CK(hipMalloc(&pos, 4)); CK(hipMalloc(&epoch, 4)); CK(hipMalloc(&err, 4));
  CK(hipMalloc(&g_x1, (size_t)H * 8)); CK(hipMalloc(&g_h, (size_t)I * 8)); CK(hipMalloc(&g_x2, (size_t)H * 8)); CK(hipMalloc(&stats_d, (size_t)G * ENG_NSTAT * 8));
  CK(hipMemsetAsync(g_x1, 0, (size_t)H * 8, st)); CK(hipMemsetAsync(g_h, 0, (size_t)I * 8, st)); CK(hipMemsetAsync(g_x2, 0, (size_t)H * 8, st));
  CK(hipMemsetAsync(stats_d, 0, (size_t)G * ENG_NSTAT * 8, st));
  hipLaunchKernelGGL(fill_f32, dim3(8), dim3(256), 0, st, attn, (size_t)qd, 77u, 1.0f, 0.f);
  hipLaunchKernelGGL(fill_f32, dim3(8), dim3(256), 0, st, x0, (size_t)H, 78u, 1.0f, 0.f);
  hipLaunchKernelGGL(fill_f32, dim3(256), dim3(256), 0, st, rc, (size_t)max_ctx * half, 79u, 1.0f, 0.f);
  hipLaunchKernelGGL(fill_f32, dim3(256), dim3(256), 0, st, rs, (size_t)max_ctx * half, 80u, 1.0f, 0.f);
  { const unsigned one = 1, zero = 0; CK(hipMemcpyAsync(pos, &pos_h, 4, hipMemcpyHostToDevice, st)); CK(hipMemcpyAsync(epoch, &one, 4, hipMemcpyHostToDevice, st)); CK(hipMemcpyAsync(err, &zero, 4, hipMemcpyHostToDevice, st)); }
  CK(hipStreamSynchronize(st));

  // ---- the launch chain (product GEMV kernels with the product's tuning: o_proj ks 1, gate_up ks 1, down ks 4, qkv ks 4; grids capped at 4 per CU)
  auto nx_of = [](int K, int ks) { return ((K / 8) + ks * 64 - 1) / (ks * 64); };
  auto grid_of = [&](int units, int ks) { const int upb = 4 / ks, want = (units + upb - 1) / upb, cap = G * 4; return want < cap ? want : cap; };
  auto launch_ref_layer = [&](int l, float* x, float* q) {
    const LayerBuf& b = lb[(size_t)l]; const LayerBuf& nb = lb[(size_t)l + 1];
    GemvArgs a{};
    a.W = b.wo; a.x = attn; a.N = H; a.K = qd; a.ldw = qd; a.units = H / 2; a.ks = 1; a.out = x; a.hd = 2;
#define LAUNCH_GEMV(PRO, EPI, KK, KS, ARGS) do { const int nx_ = nx_of(KK, KS), gr_ = grid_of((ARGS).units, KS); \
    switch (nx_) { case 1: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 1, 1>), dim3(gr_), dim3(256), 0, st, ARGS); break; \
                   case 2: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 2, 1>), dim3(gr_), dim3(256), 0, st, ARGS); break; \
                   case 3: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 3, 1>), dim3(gr_), dim3(256), 0, st, ARGS); break; \
                   case 4: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 4, 1>), dim3(gr_), dim3(256), 0, st, ARGS); break; \
                   case 6: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 6, 1>), dim3(gr_), dim3(256), 0, st, ARGS); break; \
                   case 7: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 7, 1>), dim3(gr_), dim3(256), 0, st, ARGS); break; \
                   default: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 8, 1>), dim3(gr_), dim3(256), 0, st, ARGS); break; } } while (0)
    { const int ks = nx_of(qd, 1) > 8 ? 2 : 1; a.ks = ks; LAUNCH_GEMV(PRO_PLAIN, EPI_RESIDUAL, qd, ks, a); }
    GemvArgs u{};
    u.W = b.wgu; u.x = x; u.norm_w = b.post_norm; u.eps = eps; u.N = 2 * I; u.K = H; u.ldw = H; u.units = I; u.ks = 1; u.out = h_ref; u.hd = 2;
    LAUNCH_GEMV(PRO_RMSNORM, EPI_SILU_MUL, H, 1, u);
    GemvArgs d{};
    d.W = b.wdown; d.x = h_ref; d.N = H; d.K = I; d.ldw = I; d.units = H / 2; d.ks = 4; d.out = x; d.hd = 2;
    LAUNCH_GEMV(PRO_PLAIN, EPI_RESIDUAL, I, 4, d);
    GemvArgs k{};
    k.W = nb.wqkv; k.x = x; k.norm_w = nb.in_norm; k.eps = eps; k.N = NQ; k.K = H; k.ldw = H; k.units = NQ / 2; k.ks = 4;
    k.q_out = q; k.k_cache = nb.kc_ref; k.v_cache = nb.vc_ref; k.rope_cos = rc; k.rope_sin = rs; k.pos = pos;
    k.heads = g.heads; k.kv_heads = g.kv; k.hd = g.hd; k.max_ctx = max_ctx;
    LAUNCH_GEMV(PRO_RMSNORM, EPI_QKV_ROPE, H, 4, k);
  };

  auto launch_oproj = [&](int l, float* x) {
    const LayerBuf& b = lb[(size_t)l];
    GemvArgs a{};
    a.W = b.wo; a.x = attn; a.N = H; a.K = qd; a.ldw = qd; a.units = H / 2; a.ks = 1; a.out = x; a.hd = 2;
    { const int ks = nx_of(qd, 1) > 8 ? 2 : 1; a.ks = ks; LAUNCH_GEMV(PRO_PLAIN, EPI_RESIDUAL, qd, ks, a); }
  };
  auto launch_qkv = [&](int l, float* x, float* q, bool eng_cache) {
    const LayerBuf& nb = lb[(size_t)l + 1];
    GemvArgs k{};
    k.W = nb.wqkv; k.x = x; k.norm_w = nb.in_norm; k.eps = eps; k.N = NQ; k.K = H; k.ldw = H; k.units = NQ / 2; k.ks = 4;
    k.q_out = q; k.k_cache = eng_cache ? nb.kc_eng : nb.kc_ref; k.v_cache = eng_cache ? nb.vc_eng : nb.vc_ref; k.rope_cos = rc; k.rope_sin = rs; k.pos = pos;
    k.heads = g.heads; k.kv_heads = g.kv; k.hd = g.hd; k.max_ctx = max_ctx;
    LAUNCH_GEMV(PRO_RMSNORM, EPI_QKV_ROPE, H, 4, k);
  };
  // ---- the engine
  const int xb0 = std::max(qd, I) * 4, xb1 = H * 4;
  const size_t lds = eng_lds_bytes(ns, xb0, xb1);
  printf("engine LDS %zu bytes\n", lds);
  if (lds > 160 * 1024) { printf("LDS over budget\n"); return 1; }
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&engine_kernel<DT_BF16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&engine_kernel<DT_BF16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto launch_eng_mlp = [&](int l, float* x) {
    const LayerBuf& b = lb[(size_t)l];
    EngArgs a{};
    a.nops = 2; a.ns = ns; a.xb_bytes[0] = H * 4; a.xb_bytes[1] = I * 4; a.thin = thin; a.depth = depth;
    a.x_in = x; a.pos = pos; a.heads = g.heads; a.kv_heads = g.kv; a.hd = g.hd; a.max_ctx = max_ctx; a.eps = eps; a.epoch = epoch; a.err = err; a.stats = stats ? stats_d : nullptr;
    EngOp& o0 = a.op[0]; o0.W = b.wgu; o0.norm_w = b.post_norm; o0.N = 2 * I; o0.K = H; o0.epi = EOP_SILU; o0.in_plain = x; o0.out_gran = g_h; o0.out_tag = 0;
    EngOp& o1 = a.op[1]; o1.W = b.wdown; o1.N = H; o1.K = I; o1.epi = EOP_RESID; o1.in_gran = g_h; o1.in_tag = 0; o1.out_plain = x;
    for (int k = 0; k < 2; k++) eng_plan_op(a.op[k], G);
    const size_t l2 = eng_lds_bytes(ns, H * 4, I * 4);
    if (stats) hipLaunchKernelGGL((engine_kernel<DT_BF16, true>), dim3(G), dim3(ENG_THREADS), l2, st, a);
    else hipLaunchKernelGGL((engine_kernel<DT_BF16, false>), dim3(G), dim3(ENG_THREADS), l2, st, a);
  };
  auto launch_eng_layer = [&](int l, float* x, float* q, int nops) {
    if (mode == 1) { launch_oproj(l, x); launch_eng_mlp(l, x); launch_qkv(l, x, q, true); return; }
    const LayerBuf& b = lb[(size_t)l]; const LayerBuf& nb = lb[(size_t)l + 1];
    EngArgs a{};
    a.nops = nops; a.ns = ns; a.xb_bytes[0] = xb0; a.xb_bytes[1] = xb1; a.thin = thin; a.depth = depth;
    a.x_in = x; a.q_out = q; a.k_cache = nb.kc_eng; a.v_cache = nb.vc_eng; a.rope_cos = rc; a.rope_sin = rs; a.pos = pos;
    a.heads = g.heads; a.kv_heads = g.kv; a.hd = g.hd; a.max_ctx = max_ctx; a.eps = eps; a.epoch = epoch; a.err = err; a.stats = stats ? stats_d : nullptr;
    EngOp& o0 = a.op[0]; o0.W = b.wo; o0.N = H; o0.K = qd; o0.epi = EOP_RESID; o0.in_plain = attn; o0.out_gran = g_x1; o0.out_tag = 0;
    EngOp& o1 = a.op[1]; o1.W = b.wgu; o1.norm_w = b.post_norm; o1.N = 2 * I; o1.K = H; o1.epi = EOP_SILU; o1.in_gran = g_x1; o1.in_tag = 0; o1.out_gran = g_h; o1.out_tag = 1;
    EngOp& o2 = a.op[2]; o2.W = b.wdown; o2.N = H; o2.K = I; o2.epi = EOP_RESID; o2.in_gran = g_h; o2.in_tag = 1; o2.out_plain = x;
    if (nops > 3) { o2.out_gran = g_x2; o2.out_tag = 2; }
    EngOp& o3 = a.op[3]; o3.W = nb.wqkv; o3.norm_w = nb.in_norm; o3.N = NQ; o3.K = H; o3.epi = EOP_QKV; o3.in_gran = g_x2; o3.in_tag = 2;
    for (int k = 0; k < nops; k++) eng_plan_op(a.op[k], G);
    if (stats) hipLaunchKernelGGL((engine_kernel<DT_BF16, true>), dim3(G), dim3(ENG_THREADS), lds, st, a);
    else hipLaunchKernelGGL((engine_kernel<DT_BF16, false>), dim3(G), dim3(ENG_THREADS), lds, st, a);
  };

  // ---- correctness: L layers through both chains from the same start
  CK(hipMemcpyAsync(x_ref, x0, (size_t)H * 4, hipMemcpyDeviceToDevice, st)); CK(hipMemcpyAsync(x_eng, x0, (size_t)H * 4, hipMemcpyDeviceToDevice, st));
  for (int l = 0; l < L; l++) launch_ref_layer(l, x_ref, q_ref);
  CK(hipStreamSynchronize(st));
  for (int l = 0; l < L; l++) launch_eng_layer(l, x_eng, q_eng, 4);
  CK(hipStreamSynchronize(st));
  {
    unsigned e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    printf("engine give-up code: 0x%x\n", e);
    std::vector<float> a((size_t)H), b((size_t)H), qa((size_t)qd), qb((size_t)qd);
    CK(hipMemcpy(a.data(), x_ref, (size_t)H * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), x_eng, (size_t)H * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(qa.data(), q_ref, (size_t)qd * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(qb.data(), q_eng, (size_t)qd * 4, hipMemcpyDeviceToHost));
    double ma, mq;
    const double rx = maxrel(a, b, &ma), rq = maxrel(qa, qb, &mq);
    printf("x after %d layers: max|ref| %.4g  rel diff %.3g ; q: max|ref| %.4g rel diff %.3g\n", L, ma, rx, mq, rq);
    // cache rows at pos of the last layer's qkv
    std::vector<unsigned short> ka((size_t)g.hd), kb((size_t)g.hd);
    int kbad = 0;
    for (int l = 1; l <= L; l++)
      for (int hh = 0; hh < g.kv; hh++)
        for (int kv = 0; kv < 2; kv++) {
          const unsigned short* ra = (kv ? lb[(size_t)l].vc_ref : lb[(size_t)l].kc_ref) + ((size_t)hh * max_ctx + pos_h) * g.hd;
          const unsigned short* rb = (kv ? lb[(size_t)l].vc_eng : lb[(size_t)l].kc_eng) + ((size_t)hh * max_ctx + pos_h) * g.hd;
          CK(hipMemcpy(ka.data(), ra, (size_t)g.hd * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(kb.data(), rb, (size_t)g.hd * 2, hipMemcpyDeviceToHost));
          for (int t = 0; t < g.hd; t++) { const int da = (int)ka[(size_t)t] - (int)kb[(size_t)t]; if (da > 1 || da < -1) kbad++; }
        }
    printf("cache rows differing by more than one bf16 ulp: %d of %d\n", kbad, L * g.kv * 2 * g.hd);
    if (e || rx > 1e-4 || rq > 1e-4) printf("PARITY FAILED\n"); else printf("parity ok\n");
  }

  // ---- timing: graphs over L layers
  const int NREP = 100;
  auto time_graph = [&](bool eng, const char* what) {
    hipGraph_t gr; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < L; l++) { if (eng) launch_eng_layer(l, x_eng, q_eng, 4); else launch_ref_layer(l, x_ref, q_ref); }
    CK(hipStreamEndCapture(st, &gr));
    CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 50; i++) CK(hipGraphLaunch(ex, st));
    CK(hipStreamSynchronize(st));
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < NREP; i++) CK(hipGraphLaunch(ex, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    printf("%-28s %8.2f us per layer  (%d layers, best of 5 x 100 replays)\n", what, best * 1000.0f / (float)NREP / L, L);
    CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(gr));
  };
  time_graph(false, "launches (4 GEMVs / layer)");
  time_graph(true, "engine (1 launch / layer)");
  time_graph(false, "launches (4 GEMVs / layer)");
  time_graph(true, "engine (1 launch / layer)");
  {
    unsigned e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    printf("engine give-up code after timing: 0x%x\n", e);
  }
  if (stats) {
    std::vector<unsigned long long> s((size_t)G * ENG_NSTAT);
    CK(hipMemcpy(s.data(), stats_d, (size_t)G * ENG_NSTAT * 8, hipMemcpyDeviceToHost));
    double avg[ENG_NSTAT] = {0}; unsigned long long mx[ENG_NSTAT] = {0};
    for (int c = 0; c < G; c++) for (int k = 0; k < ENG_NSTAT; k++) { avg[k] += (double)s[(size_t)c * ENG_NSTAT + k] / G; mx[k] = std::max(mx[k], s[(size_t)c * ENG_NSTAT + k]); }
    const char* nm[ENG_NSTAT] = {"loader op0 issued", "loader op1 issued", "loader op2 issued", "loader op3 issued", "loader all landed", "loader blocked (ring full)",
      "gather op0 staged", "gather op1 staged", "gather op2 staged", "gather op3 staged", "gather ticks in sweeps", "",
      "cons0 op0 done", "cons0 op1 done", "cons0 op2 done", "cons0 op3 done", "cons1 op0 done", "cons1 op1 done", "cons1 op2 done", "cons1 op3 done",
      "cons2 op0 done", "cons2 op1 done", "cons2 op2 done", "cons2 op3 done", "cons0 wait input", "cons1 wait input", "cons2 wait input",
      "cons0 wait tiles", "cons1 wait tiles", "cons2 wait tiles", "cons0 tile work", ""};
    {   // skew by XCD (workgroup c runs on XCD c % 8): when did each CU's loader finish issuing gate_up, when did its consumers finish it
      printf("per XCD (c %% 8): loader op1 issued avg/min/max | slowest consumer op1 done avg/min/max  [us]\n");
      for (int x = 0; x < 8; x++) {
        double a1 = 0, a2 = 0, mn1 = 1e9, mx1 = 0, mn2 = 1e9, mx2 = 0; int n = 0;
        for (int c = x; c < G; c += 8, n++) {
          const double l1 = (double)s[(size_t)c * ENG_NSTAT + 1] / 100.0;
          const double c1 = (double)std::max(std::max(s[(size_t)c * ENG_NSTAT + 13], s[(size_t)c * ENG_NSTAT + 17]), s[(size_t)c * ENG_NSTAT + 21]) / 100.0;
          a1 += l1; a2 += c1; mn1 = std::min(mn1, l1); mx1 = std::max(mx1, l1); mn2 = std::min(mn2, c1); mx2 = std::max(mx2, c1);
        }
        printf("  xcd %d: %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f\n", x, a1 / n, mn1, mx1, a2 / n, mn2, mx2);
      }
    }
    printf("last launch (last layer of the last replay), us since kernel start (avg over CUs / max):\n");
    for (int k = 0; k < ENG_NSTAT; k++) if (nm[k][0]) printf("  %-28s %7.2f / %7.2f\n", nm[k], avg[k] / 100.0, (double)mx[k] / 100.0);
  }
  return 0;
}
