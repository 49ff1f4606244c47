// layer_lab.hip — the batch-1 decode layer of the product (kernels/gemv.h, kernels/attn_decode.h) as a standalone harness: the six launches of a layer
// {qkv, attention, combine, o_proj, gate_up, down} over L layers of distinct weights and caches, as one hipGraph (the product's structure) and class by
// class (one class back-to-back over all layers, what tgx_profile_decode measures).  Compiles in seconds — the place where kernel variants are tried
// against the product kernels on identical data before they enter csrc/decode.hip.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -DLAB_VARIANTS -DLAB_EPOCH -I../../tinygpt_amd/csrc layer_lab.hip -o build/layer_lab
// Run:   layer_lab [geom=1b|0.5b|3b|7b] [pos=2064] [layers=16] [nsplit=CUs / kv heads, <= 32]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include "kernels/gemv.h"
#include "kernels/attn_decode.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
using namespace tgx;

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    p[i] = f32_to_bf16(((float)(x & 0xffff) / 32768.0f - 1.0f) * scale);
  }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale, float bias) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    p[i] = ((float)(x & 0xffff) / 32768.0f - 1.0f) * scale + bias;
  }
}

struct Geom { int H, I, heads, kv, hd; };
struct LayerBuf { unsigned short *wo, *wgu, *wdown, *wqkv, *post_norm, *in_norm, *kc, *vc; };

struct Lab {
  Geom g; int L, G, max_ctx, pos_h; float eps = 1e-5f;
  int H, I, qd, kvd, NQ, half, nsplit; size_t part_row;
  hipStream_t st;
  std::vector<LayerBuf> lb;
  float *attn, *x, *h, *q, *rc, *rs, *part, *scratch_x; int* pos;
  int nx_of(int K, int ks) const { return ((K / 8) + ks * 64 - 1) / (ks * 64); }
  int grid_of(int units, int ks, int bpc = 4) const { const int upb = 4 / ks, want = (units + upb - 1) / upb, cap = G * bpc; return want < cap ? want : cap; }
};

#define LAUNCH_GEMV(lab, PRO, EPI, KK, KS, ARGS) do { const int nx_ = (lab).nx_of(KK, KS), gr_ = (lab).grid_of((ARGS).units, KS); \
    switch (nx_) { case 1: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 1, 1>), dim3(gr_), dim3(256), 0, (lab).st, ARGS); break; \
                   case 2: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 2, 1>), dim3(gr_), dim3(256), 0, (lab).st, ARGS); break; \
                   case 3: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 3, 1>), dim3(gr_), dim3(256), 0, (lab).st, ARGS); break; \
                   case 4: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 4, 1>), dim3(gr_), dim3(256), 0, (lab).st, ARGS); break; \
                   case 5: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 5, 1>), dim3(gr_), dim3(256), 0, (lab).st, ARGS); break; \
                   case 6: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 6, 1>), dim3(gr_), dim3(256), 0, (lab).st, ARGS); break; \
                   case 7: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 7, 1>), dim3(gr_), dim3(256), 0, (lab).st, ARGS); break; \
                   default: hipLaunchKernelGGL((gemv_kernel<DT_BF16, PRO, EPI, 8, 1>), dim3(gr_), dim3(256), 0, (lab).st, ARGS); break; } } while (0)

// ---- the product's launches (csrc/decode.hip launch_layer_kernel, batch 1, split-form attention) ----
static void p_qkv(Lab& b, int l, float* resid) {
  (void)resid;
  const LayerBuf& w = b.lb[(size_t)l];
  GemvArgs k{};
  const int ks = b.H >= 2048 ? 4 : 1;
  k.W = w.wqkv; k.x = b.x; k.norm_w = w.in_norm; k.eps = b.eps; k.N = b.NQ; k.K = b.H; k.ldw = b.H; k.units = b.NQ / 2; k.ks = ks;
  k.q_out = b.q; k.k_cache = w.kc; k.v_cache = w.vc; k.rope_cos = b.rc; k.rope_sin = b.rs; k.pos = b.pos;
  k.heads = b.g.heads; k.kv_heads = b.g.kv; k.hd = b.g.hd; k.max_ctx = b.max_ctx;
  LAUNCH_GEMV(b, PRO_RMSNORM, EPI_QKV_ROPE, b.H, ks, k);
}
static int g_attn_dbg = 0;
static AttnArgs attn_args(Lab& b, int l) {
  const LayerBuf& w = b.lb[(size_t)l];
  AttnArgs a{};
  a.q = b.q; a.k_cache = w.kc; a.v_cache = w.vc; a.pos = b.pos; a.part = b.part; a.out = b.attn;
  a.heads = b.g.heads; a.kv_heads = b.g.kv; a.max_ctx = b.max_ctx; a.nsplit = b.nsplit; a.scale = 1.0f / sqrtf((float)b.g.hd);
  a.q_stride = b.qd; a.kv_stride = 0; a.part_stride = (long long)b.part_row;
  a.gfull = b.g.heads / b.g.kv; a.dbg = g_attn_dbg;
  return a;
}
static void p_attn_only(Lab& b, int l, float*) {
  AttnArgs a = attn_args(b, l);
  const int gfull = a.gfull, gmax = 2, ngroups = gfull > gmax ? (gfull + gmax - 1) / gmax : 1, Gh = (gfull + ngroups - 1) / ngroups;
  const dim3 grid(a.kv_heads * a.nsplit, 1, ngroups), blk(256);
  if (b.g.hd == 64) {
    switch (Gh) { case 1: hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 64, 1, 4>), grid, blk, 0, b.st, a); break;
                  case 2: hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 64, 2, 4>), grid, blk, 0, b.st, a); break;
                  case 3: hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 64, 3, 4>), grid, blk, 0, b.st, a); break;
                  default: hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 64, 4, 4>), grid, blk, 0, b.st, a); break; }
  } else {
    switch (Gh) { case 1: hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 128, 1, 4>), grid, blk, 0, b.st, a); break;
                  case 2: hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 128, 2, 4>), grid, blk, 0, b.st, a); break;
                  case 3: hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 128, 3, 4>), grid, blk, 0, b.st, a); break;
                  default: hipLaunchKernelGGL((attn_decode_kernel<DT_BF16, 128, 4, 4>), grid, blk, 0, b.st, a); break; }
  }
}
static void p_combine(Lab& b, int l, float*) {
  AttnArgs a = attn_args(b, l);
  if (b.g.hd == 64) hipLaunchKernelGGL((attn_combine_kernel<64>), dim3(a.heads, 1), dim3(256), 0, b.st, a);
  else hipLaunchKernelGGL((attn_combine_kernel<128>), dim3(a.heads, 1), dim3(256), 0, b.st, a);
}
static void p_attn(Lab& b, int l, float* r) { p_attn_only(b, l, r); p_combine(b, l, r); }
static void p_oproj(Lab& b, int l, float* resid) {
  const LayerBuf& w = b.lb[(size_t)l];
  GemvArgs a{};
  a.W = w.wo; a.x = b.attn; a.N = b.H; a.K = b.qd; a.ldw = b.qd; a.units = b.H / 2; a.ks = 1; a.out = resid; a.hd = 2;
  LAUNCH_GEMV(b, PRO_PLAIN, EPI_RESIDUAL, b.qd, 1, a);
}
static void p_gateup(Lab& b, int l, float*) {
  const LayerBuf& w = b.lb[(size_t)l];
  GemvArgs u{};
  u.W = w.wgu; u.x = b.x; u.norm_w = w.post_norm; u.eps = b.eps; u.N = 2 * b.I; u.K = b.H; u.ldw = b.H; u.units = b.I; u.ks = 1; u.out = b.h; u.hd = 2;
  LAUNCH_GEMV(b, PRO_RMSNORM, EPI_SILU_MUL, b.H, 1, u);
}
static void p_down(Lab& b, int l, float* resid) {
  const LayerBuf& w = b.lb[(size_t)l];
  GemvArgs d{};
  int ks = 4; while (ks < 4 && b.nx_of(b.I, ks) > 8) ks *= 2;
  d.W = w.wdown; d.x = b.h; d.N = b.H; d.K = b.I; d.ldw = b.I; d.units = b.H / 2; d.ks = ks; d.out = resid; d.hd = 2;
  LAUNCH_GEMV(b, PRO_PLAIN, EPI_RESIDUAL, b.I, ks, d);
}

typedef void (*LaunchFn)(Lab&, int, float*);
struct Class { const char* name; LaunchFn fn; };

static float time_graph(Lab& b, const std::function<void()>& body, int per) {
  hipGraph_t gr; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(b.st, hipStreamCaptureModeThreadLocal));
  body();
  CK(hipStreamEndCapture(b.st, &gr));
  CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; i++) CK(hipGraphLaunch(ex, b.st));
  CK(hipStreamSynchronize(b.st));
  float best = 1e9f;
  const int NREP = 20;
  for (int rep = 0; rep < 5; rep++) {
    CK(hipEventRecord(e0, b.st));
    for (int i = 0; i < NREP; i++) CK(hipGraphLaunch(ex, b.st));
    CK(hipEventRecord(e1, b.st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(gr));
  return best * 1000.0f / NREP / per;
}

#ifdef LAB_VARIANTS
#include "lab_variants.h"
#endif

int main(int argc, char** argv) {
  Lab b{};
  const char* gname = argc > 1 ? argv[1] : "1b";
  b.pos_h = argc > 2 ? atoi(argv[2]) : 2064;
  b.L = argc > 3 ? atoi(argv[3]) : 16;
  b.g = Geom{2048, 8192, 32, 8, 64};
  if (!strcmp(gname, "0.5b")) b.g = Geom{896, 4864, 14, 2, 64};
  if (!strcmp(gname, "3b")) b.g = Geom{3072, 8192, 24, 8, 128};
  if (!strcmp(gname, "7b")) b.g = Geom{4096, 14336, 32, 8, 128};
  b.H = b.g.H; b.I = b.g.I; b.qd = b.g.heads * b.g.hd; b.kvd = b.g.kv * b.g.hd; b.NQ = b.qd + 2 * b.kvd; b.half = b.g.hd / 2;
  b.max_ctx = 4096;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  b.G = prop.multiProcessorCount;
  b.nsplit = argc > 4 ? atoi(argv[4]) : std::min(32, std::max(1, b.G / b.g.kv));
  b.part_row = (size_t)b.g.heads * b.nsplit * (b.g.hd + 4);
  printf("device %s, %d CUs; geometry %s H=%d I=%d heads=%d kv=%d hd=%d; %d layers, context %d, nsplit %d\n", prop.name, b.G, gname, b.H, b.I, b.g.heads, b.g.kv, b.g.hd, b.L, b.pos_h, b.nsplit);
  CK(hipStreamCreateWithFlags(&b.st, hipStreamNonBlocking));
  b.lb.resize((size_t)b.L);
  const size_t cache_elems = (size_t)b.g.kv * b.max_ctx * b.g.hd;
  for (int l = 0; l < b.L; l++) {
    LayerBuf& w = b.lb[(size_t)l];
    CK(hipMalloc(&w.wo, (size_t)b.H * b.qd * 2)); CK(hipMalloc(&w.wgu, (size_t)2 * b.I * b.H * 2)); CK(hipMalloc(&w.wdown, (size_t)b.H * b.I * 2));
    CK(hipMalloc(&w.wqkv, (size_t)b.NQ * b.H * 2)); CK(hipMalloc(&w.post_norm, (size_t)b.H * 2)); CK(hipMalloc(&w.in_norm, (size_t)b.H * 2));
    CK(hipMalloc(&w.kc, cache_elems * 2)); CK(hipMalloc(&w.vc, cache_elems * 2));
    const unsigned s0 = 1000u * (unsigned)l;
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, b.st, w.wo, (size_t)b.H * b.qd, s0 + 1, 0.02f);
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, b.st, w.wgu, (size_t)2 * b.I * b.H, s0 + 2, 0.03f);
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, b.st, w.wdown, (size_t)b.H * b.I, s0 + 3, 0.02f);
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, b.st, w.wqkv, (size_t)b.NQ * b.H, s0 + 4, 0.03f);
    hipLaunchKernelGGL(fill_bf16, dim3(8), dim3(256), 0, b.st, w.post_norm, (size_t)b.H, s0 + 5, 0.5f);
    hipLaunchKernelGGL(fill_bf16, dim3(8), dim3(256), 0, b.st, w.in_norm, (size_t)b.H, s0 + 6, 0.5f);
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, b.st, w.kc, cache_elems, s0 + 7, 1.0f);
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, b.st, w.vc, cache_elems, s0 + 8, 1.0f);
  }
  CK(hipMalloc(&b.attn, (size_t)b.qd * 4)); CK(hipMalloc(&b.x, (size_t)b.H * 4)); CK(hipMalloc(&b.scratch_x, (size_t)b.H * 4)); CK(hipMalloc(&b.h, (size_t)b.I * 4)); CK(hipMalloc(&b.q, (size_t)b.qd * 4));
  CK(hipMalloc(&b.rc, (size_t)b.max_ctx * b.half * 4)); CK(hipMalloc(&b.rs, (size_t)b.max_ctx * b.half * 4)); CK(hipMalloc(&b.part, b.part_row * 4)); CK(hipMalloc(&b.pos, 4));
  hipLaunchKernelGGL(fill_f32, dim3(8), dim3(256), 0, b.st, b.attn, (size_t)b.qd, 77u, 1.0f, 0.f);
  hipLaunchKernelGGL(fill_f32, dim3(8), dim3(256), 0, b.st, b.x, (size_t)b.H, 78u, 1.0f, 0.f);
  hipLaunchKernelGGL(fill_f32, dim3(8), dim3(256), 0, b.st, b.scratch_x, (size_t)b.H, 78u, 1.0f, 0.f);
  hipLaunchKernelGGL(fill_f32, dim3(32), dim3(256), 0, b.st, b.h, (size_t)b.I, 81u, 1.0f, 0.f);
  hipLaunchKernelGGL(fill_f32, dim3(8), dim3(256), 0, b.st, b.q, (size_t)b.qd, 82u, 1.0f, 0.f);
  hipLaunchKernelGGL(fill_f32, dim3(256), dim3(256), 0, b.st, b.rc, (size_t)b.max_ctx * b.half, 79u, 1.0f, 0.f);
  hipLaunchKernelGGL(fill_f32, dim3(256), dim3(256), 0, b.st, b.rs, (size_t)b.max_ctx * b.half, 80u, 1.0f, 0.f);
  CK(hipMemsetAsync(b.part, 0, b.part_row * 4, b.st));
  CK(hipMemcpyAsync(b.pos, &b.pos_h, 4, hipMemcpyHostToDevice, b.st));
  CK(hipStreamSynchronize(b.st));

  const Class cls[] = {{"qkv", p_qkv}, {"attn", p_attn_only}, {"combine", p_combine}, {"o_proj", p_oproj}, {"gate_up", p_gateup}, {"down", p_down}};
  float sum = 0.f;
  printf("product kernels, one class back-to-back over %d layers (us per launch):\n", b.L);
  for (const Class& c : cls) {
    const float t = time_graph(b, [&] { for (int l = 0; l < b.L; l++) c.fn(b, l, b.scratch_x); }, b.L);
    printf("  %-10s %7.2f\n", c.name, t); sum += t;
  }
  printf("  %-10s %7.2f\n", "sum", sum);
  const float whole = time_graph(b, [&] { for (int l = 0; l < b.L; l++) for (const Class& c : cls) c.fn(b, l, b.scratch_x); }, b.L);
  printf("product layer as one graph (6 launches per layer, residual into a scratch vector): %.2f us per layer\n", whole);
#if TGX_DISSECT
  for (int d : {1, 2, 3, 4}) {
    g_attn_dbg = d;
    const float t = time_graph(b, [&] { for (int l = 0; l < b.L; l++) p_attn_only(b, l, nullptr); }, b.L);
    printf("  attn dissect %d (1 no K/V loop, 2 no merge, 4 exit at once): %.2f us\n", d, t);
  }
  g_attn_dbg = 0;
#endif
#ifdef LAB_VARIANTS
  lab_variants_main(b);
#endif
  return 0;
}
