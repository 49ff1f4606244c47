// abi.hip — the extern "C" shim of include/tgx.h for MI355X (gfx950): context, weight upload by HF name, KV cache, decode-step hipGraphs and the
// decode loop.  The kernels and their launch sequences live in the sibling translation units (ctx.h lists them).
//
// One context = one GPU = one HIP stream; every call comes from one host thread (the reference's engine is
// entered by one thread only: examples/inference/main.cpp, server/HttpServer.cpp:118-163).
// There is NO CPU path in this library: every entry point either runs on the GPU or returns an error.
#include <climits>
#include "ctx.h"

static std::string g_create_err;

int set_err(tgx_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_err = buf;
  return code;
}

namespace {

inline uint16_t host_f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float host_bf16_to_f32(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline float host_half_to_f32(uint16_t h) {
  uint32_t s = (h >> 15) & 1, e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
  if (e == 0) {
    if (m == 0) u = s << 31;
    else { e = 127 - 15 + 1; while (!(m & 0x400)) { m <<= 1; e--; } m &= 0x3ff; u = (s << 31) | (e << 23) | (m << 13); }
  } else if (e == 31) u = (s << 31) | 0x7f800000u | (m << 13);
  else u = (s << 31) | ((e + 127 - 15) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}

inline uint16_t host_f32_to_half(float f) {   // round-to-nearest-even, subnormals kept (== torch .to(float16))
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u, abs = u & 0x7fffffffu;
  if (abs > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
  if (abs >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);                // >= 65536 (or inf) -> inf; 65520..65536 handled below
  if (abs < 0x33000000u) return (uint16_t)sign;                            // < 2^-25 -> 0
  int e = (int)(abs >> 23) - 127;
  uint32_t m = (abs & 0x7fffffu) | 0x800000u;                              // 24-bit significand
  int shift = e >= -14 ? 13 : 13 + (-14 - e);                              // bits dropped (subnormal: more)
  uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (q & 1u))) q++;
  uint32_t h = e >= -14 ? ((uint32_t)(e + 15) << 10) + (q - 0x400u) : q;  // carries propagate into the exponent
  if (h >= 0x7c00u) h = 0x7c00u;
  return (uint16_t)(sign | h);
}

// host -> device copy with conversion to the compute dtype (== model().to(dtype), ModelLoader.cpp:84):
// widening is exact, narrowing rounds to nearest even once.
int upload_param(tgx_ctx* c, ebyte* dst, const void* host, int64_t n, int src_dtype) {
  const int want = c->d.compute_dtype;
  if (src_dtype != TGX_BF16 && src_dtype != TGX_F32 && src_dtype != TGX_F16) return set_err(c, TGX_ERR_INVALID, "unknown source dtype %d", src_dtype);
  if (src_dtype == want) {
    HIP_OK(c, hipMemcpy(dst, host, (size_t)n * c->esz, hipMemcpyHostToDevice));
    return TGX_OK;
  }
  auto src = [&](int64_t i) -> float {
    if (src_dtype == TGX_F32) return ((const float*)host)[i];
    if (src_dtype == TGX_BF16) return host_bf16_to_f32(((const uint16_t*)host)[i]);
    return host_half_to_f32(((const uint16_t*)host)[i]);
  };
  if (want == TGX_F32) {
    std::vector<float> tmp((size_t)n);
    for (int64_t i = 0; i < n; i++) tmp[(size_t)i] = src(i);
    HIP_OK(c, hipMemcpy(dst, tmp.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  } else {
    std::vector<uint16_t> tmp((size_t)n);
    if (want == TGX_BF16) for (int64_t i = 0; i < n; i++) tmp[(size_t)i] = host_f32_to_bf16(src(i));
    else for (int64_t i = 0; i < n; i++) tmp[(size_t)i] = host_f32_to_half(src(i));
    HIP_OK(c, hipMemcpy(dst, tmp.data(), (size_t)n * 2, hipMemcpyHostToDevice));
  }
  return TGX_OK;
}

bool shape_is(const int64_t* s, int nd, int64_t a, int64_t b) {
  if (b < 0) return nd == 1 && s[0] == a;
  return nd == 2 && s[0] == a && s[1] == b;
}

// GPT-2 checkpoints (hub layout without the "transformer." prefix, ModelGPT2.h:226; the prefixed form is accepted too).
// Conv1D weights are stored [in][out] (ModelGPT2.h:26): transposed on the host into the [out][in] rows the GEMV streams.
int upload_gpt2(tgx_ctx* c, const char* name, const void* host, const int64_t* shape, int nd, int src_dtype) {
  const tgx_model_desc& d = c->d;
  const int64_t H = d.hidden, I = d.inter, V = d.vocab;
  auto bad_shape = [&]() { return set_err(c, TGX_ERR_SHAPE, "shape not equal for tensor: %s", name); };
  if (!strncmp(name, "transformer.", 12)) name += 12;
  if (!strcmp(name, "wte.weight")) { if (!shape_is(shape, nd, V, H)) return bad_shape(); c->embed_ok = true; return upload_param(c, c->embed, host, V * H, src_dtype); }
  if (!strcmp(name, "wpe.weight")) { if (!shape_is(shape, nd, d.n_positions, H)) return bad_shape(); c->wpe_ok = true; return upload_param(c, c->wpe, host, (int64_t)d.n_positions * H, src_dtype); }
  if (!strcmp(name, "ln_f.weight")) { if (!shape_is(shape, nd, H, -1)) return bad_shape(); c->final_norm_ok = true; return upload_param(c, c->final_norm, host, H, src_dtype); }
  if (!strcmp(name, "ln_f.bias")) { if (!shape_is(shape, nd, H, -1)) return bad_shape(); c->final_norm_b_ok = true; return upload_param(c, c->final_norm_b, host, H, src_dtype); }
  if (!strcmp(name, "lm_head.weight")) { if (!shape_is(shape, nd, V, H)) return bad_shape(); return TGX_OK; }   // aliases wte
  int l = -1;
  char rest[128] = {0};
  if (sscanf(name, "h.%d.%127s", &l, rest) == 2 && l >= 0 && l < d.layers) {
    LayerW& w = c->L[(size_t)l];
    struct Vec { const char* n; ebyte* p; int64_t len; };
    const Vec vecs[] = {{"ln_1.weight", w.in_norm, H}, {"ln_1.bias", w.in_norm_b, H}, {"ln_2.weight", w.post_norm, H}, {"ln_2.bias", w.post_norm_b, H},
                        {"attn.c_attn.bias", w.bqkv, 3 * H}, {"attn.c_proj.bias", w.bo, H}, {"mlp.c_fc.bias", w.bfc, I}, {"mlp.c_proj.bias", w.bdown, H}};
    for (int i = 0; i < 8; i++)
      if (!strcmp(rest, vecs[i].n)) {
        if (!shape_is(shape, nd, vecs[i].len, -1)) return bad_shape();
        w.gpt2_filled |= 1 << i;
        return upload_param(c, vecs[i].p, host, vecs[i].len, src_dtype);
      }
    struct Mat { const char* n; ebyte* p; int64_t in, out; };
    const Mat mats[] = {{"attn.c_attn.weight", w.wqkv, H, 3 * H}, {"attn.c_proj.weight", w.wo, H, H}, {"mlp.c_fc.weight", w.wgu, H, I}, {"mlp.c_proj.weight", w.wdown, I, H}};
    for (int i = 0; i < 4; i++)
      if (!strcmp(rest, mats[i].n)) {
        if (!shape_is(shape, nd, mats[i].in, mats[i].out)) return bad_shape();
        if (src_dtype != TGX_BF16 && src_dtype != TGX_F32 && src_dtype != TGX_F16) return set_err(c, TGX_ERR_INVALID, "unknown source dtype %d", src_dtype);
        const size_t es = src_dtype == TGX_F32 ? 4 : 2, n_in = (size_t)mats[i].in, n_out = (size_t)mats[i].out;
        std::vector<unsigned char> t(n_in * n_out * es);
        const unsigned char* src = static_cast<const unsigned char*>(host);
        for (size_t k = 0; k < n_in; k++)
          for (size_t n = 0; n < n_out; n++) memcpy(&t[(n * n_in + k) * es], src + (k * n_out + n) * es, es);
        w.gpt2_filled |= 1 << (8 + i);
        return upload_param(c, mats[i].p, t.data(), (int64_t)(n_in * n_out), src_dtype);
      }
    if (!strcmp(rest, "attn.bias") || !strcmp(rest, "attn.masked_bias")) return TGX_OK;   // causal-mask buffers of old hub checkpoints: not parameters
  }
  return set_err(c, TGX_ERR_NAME, "Unexpected key: %s", name);
}

// nn::RoPE tables (ctor at ModelLlama.h:41-42): HF LlamaRotaryEmbedding incl. llama3 scaling, fp32.
void build_rope_host(const tgx_model_desc& d, std::vector<float>& cs, std::vector<float>& sn) {
  const int half = d.head_dim / 2;
  std::vector<float> inv((size_t)half);
  for (int i = 0; i < half; i++) {
    const float e = (float)(2 * i) / (float)d.head_dim;
    const float p = (float)std::pow((double)d.rope_theta, (double)e);
    inv[(size_t)i] = 1.0f / p;
  }
  if (d.family == TGX_FAMILY_LLAMA && d.rope_factor > 0.f) {
    const float factor = d.rope_factor, lo = d.rope_low_freq, hi = d.rope_high_freq, old = (float)d.rope_orig_ctx;
    const float low_wl = old / lo, high_wl = old / hi;
    for (int i = 0; i < half; i++) {
      const float wl = 2.0f * (float)M_PI / inv[(size_t)i];
      const float v = wl > low_wl ? inv[(size_t)i] / factor : inv[(size_t)i];
      const float smooth = (old / wl - lo) / (hi - lo);
      const float sm = (1.0f - smooth) * v / factor + smooth * v;
      const bool medium = !(wl < high_wl) && !(wl > low_wl);
      inv[(size_t)i] = medium ? sm : v;
    }
  }
  cs.resize((size_t)d.max_ctx * half);
  sn.resize((size_t)d.max_ctx * half);
  for (int p = 0; p < d.max_ctx; p++)
    for (int i = 0; i < half; i++) {
      const float a = inv[(size_t)i] * (float)p;
      cs[(size_t)p * half + i] = cosf(a);
      sn[(size_t)p * half + i] = sinf(a);
    }
}

}  // namespace

bool is_greedy(const tgx_sampler_cfg* s) {   // Sampler.cpp:15-21
  return !(s->temperature > 0.f || s->top_k > 0 || s->top_p < 1.f || s->min_p > 0.f);
}

// Context from which the MFMA decode attention (kernels/attn_decode_mfma.h) beats the VALU kernel, measured per geometry class
// (profiles/r02_attn_long.txt): head_dim 64 with 8 kv heads from ~6k keys (Llama-3.2-1B: 11.0 -> 9.9 µs per layer at 6k, 23.3 -> 18.3 at 30k);
// two kv heads (Qwen2.5-0.5B: few workgroups) and head_dim 128 (Mistral-7B, Llama-3.2-3B) from ~14k (Mistral-7B: 25.3 -> 21.4 at 16k, 39.5 -> 31.4 at 30k).
static int attn_mfma_threshold(const tgx_ctx* c) {
  if (c->attn_mfma_min >= 0) return c->attn_mfma_min;
  return (c->d.head_dim == 64 && c->d.kv_heads >= 8) ? 6000 : 14000;
}
// The attention form of the launches about to be issued / captured, from the context the call ends at: direct (one workgroup per head, no
// combine) for short contexts, the MFMA decode attention for long ones, the VALU split form in between.  One place for all callers
// (the prefill-by-steps branch used to leave attn_mfma at whatever the previous decode call had chosen).
// contexts up to which a decode call of this batch runs the direct form / its four-wave variant: batch-1 steps whose o_proj rides in the attention
// launch (decode.hip oproj_fused_capable) keep the direct form longer and on fewer waves
static long long direct_limit(const tgx_ctx* c, int rpl, bool step) {
  if (step && rpl == 1 && c->batch == 1 && oproj_fused_capable(c)) return c->attn_fused_max;
  // (2-3 rows, round 4: the direct form stays ahead to ~2x / ~3x the batch-1 limit — Llama-3.2-1B B = 3 at context 600 0.888 -> 0.820 ms per step, at 1200 0.959 -> 0.856;
  //  crossovers measured at ~2000 / ~2800 keys there, ~900 / ~1600 on Mistral-7B, ~1400 / > 1500 on Qwen2.5-0.5B)
  return (long long)c->attn_direct_max * (rpl >= 2 ? rpl : 1);
}
static long long nw4_limit(const tgx_ctx* c, int rpl, bool step) {
  if (rpl >= 4) return 0;
  if (step && rpl == 1 && c->batch == 1 && oproj_fused_capable(c)) return c->attn_fused_nw4;
  return c->attn_direct_nw4;
}
static void update_attn_modes(tgx_ctx* c, int n_positions, int rows_per_launch = -1, bool step = true) {   // rows_per_launch: batch rows that share an attention launch (-1: the batch); step: a decode step (not the chunk rows of a prefill-by-steps pass)
  // batches (round 3): the rows themselves fill the chip, so the one-workgroup-per-(kv head, row) form stays ahead of the split form far beyond the
  // batch-1 crossover — Llama-3.2-1B at context 2k: B = 8 1.250 -> 1.105 ms/step, B = 32 2.360 -> 1.680; Mistral-7B at 600: B = 32 6.91 -> 5.47
  const int rpl = rows_per_launch < 0 ? c->batch : rows_per_launch;
  const long long direct_lim = direct_limit(c, rpl, step), nw4_lim = nw4_limit(c, rpl, step);
  c->attn_direct = c->past + n_positions <= direct_lim;
  c->attn_nw4 = c->attn_direct && nw4_lim > 0 && c->past + n_positions <= nw4_lim;
  c->attn_mfma = !c->attn_direct && c->past >= attn_mfma_threshold(c) && c->dt != tgx::DT_F32 && !(c->d.qk_norm && c->d.head_dim == 128 && c->qk_fuse);
}

static void launch_decode_step(tgx_ctx* c, const tgx_sampler_cfg& cfg) {
  if (decode_mfma_ok(c)) {   // more than 4 rows: every Linear is one pass over its weights for up to 32 rows (kernels/skinny.h)
    // rows per weight pass (option decode.step_rows: 32 or 64): batches beyond 32 rows take four activation blocks per skinny product (one pass over
    // the weights for up to 64 rows) instead of two passes of two blocks
    // (128 rows = eight blocks exist on the LDS-DMA ring kernel only: every product must then take stored terms, i.e. the attention a direct form that writes them)
    const bool wide_ok = c->skinny_dma && c->skinny_dma_oproj >= 2 && c->attn_direct && c->dt != tgx::DT_F32;
    const int per = c->decode_step_rows > 64 && !wide_ok ? 64 : c->decode_step_rows;
    for (int row0 = 0; row0 < c->batch; row0 += per) launch_decode_step_mfma(c, row0, std::min(per, c->batch - row0), cfg);
    return;
  }
  // batch rows share each pass over the weights in groups of 4 / 2 / 1 (the batched GEMV's R template)
  for (int row0 = 0; row0 < c->batch;) {
    const int rem = c->batch - row0, R = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
    launch_layers(c, row0, R);
    launch_lm_head(c, row0, R);
    launch_sample(c, row0, R, cfg, /*advance_pos=*/true, /*log_step=*/true);
    row0 += R;
  }
}

static bool same_cfg(const tgx_sampler_cfg& a, const tgx_sampler_cfg& b) {
  return a.temperature == b.temperature && a.top_k == b.top_k && a.top_p == b.top_p && a.min_p == b.min_p;
}

// The decode step as a hipGraph, captured once per (batch, sampler config): `steps` consecutive steps per graph — token,
// position and step counter live on the device, so a multi-step graph is the same launch sequence repeated.
static int capture_steps(tgx_ctx* c, const tgx_sampler_cfg& cfg, int steps, hipGraphExec_t* out) {
  hipGraph_t g = nullptr;
  // multi-step graphs serve tgx_decode, which reads the ids from the device log afterwards: no per-step store over PCIe
  c->mirror_to_host = steps == 1;
  HIP_OK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < steps; i++) launch_decode_step(c, cfg);
  const hipError_t cap = hipStreamEndCapture(c->stream, &g);
  c->mirror_to_host = true;
  HIP_OK(c, cap);
  if (c->launch_fault) (void)hipGraphDestroy(g);
  LAUNCH_OK(c);
  HIP_OK(c, hipGraphInstantiate(out, g, nullptr, nullptr, 0));
  (void)hipGraphDestroy(g);
  return TGX_OK;
}

void drop_step_graphs(tgx_ctx* c) {
  bool any = false;
  for (auto& g : c->graph_cache) any |= g.step != nullptr || g.multi != nullptr;
  c->step_graph = nullptr; c->multi_graph = nullptr; c->graph_cur = -1;
  if (!any) return;
  (void)hipStreamSynchronize(c->stream);
  for (auto& g : c->graph_cache) {
    if (g.step) (void)hipGraphExecDestroy(g.step);
    if (g.multi) (void)hipGraphExecDestroy(g.multi);
    g = tgx_ctx::GraphSet{};
  }
}

static int ensure_step_graph(tgx_ctx* c, const tgx_sampler_cfg& cfg, bool want_multi) {
  if (!c->use_graph) return TGX_OK;
  int hit = -1, victim = 0;
  for (int i = 0; i < 6; i++) {
    const tgx_ctx::GraphSet& g = c->graph_cache[i];
    if (g.step && g.batch == c->batch && same_cfg(g.cfg, cfg) && g.direct == c->attn_direct && g.mfma == c->attn_mfma && g.nw4 == c->attn_nw4) { hit = i; break; }
    if (!g.step) victim = i;
    else if (c->graph_cache[victim].step && g.used < c->graph_cache[victim].used) victim = i;
  }
  if (hit < 0) {
    tgx_ctx::GraphSet& g = c->graph_cache[victim];
    c->step_graph = nullptr; c->multi_graph = nullptr; c->graph_cur = -1;     // never left pointing at an evicted or half-captured set
    if (g.step || g.multi) {      // evict the least recently used set (its replays may still be in flight)
      HIP_OK(c, hipStreamSynchronize(c->stream));
      if (g.step) (void)hipGraphExecDestroy(g.step);
      if (g.multi) (void)hipGraphExecDestroy(g.multi);
      g = tgx_ctx::GraphSet{};
    }
    int rc = capture_steps(c, cfg, 1, &g.step);
    if (rc) { g = tgx_ctx::GraphSet{}; return rc; }
    g.batch = c->batch; g.cfg = cfg; g.direct = c->attn_direct; g.mfma = c->attn_mfma; g.nw4 = c->attn_nw4;
    g.used = ++c->graph_clock;
    hit = victim;
  }
  tgx_ctx::GraphSet& g = c->graph_cache[hit];
  if (want_multi && !g.multi && c->graph_steps > 1) { int rc = capture_steps(c, cfg, c->graph_steps, &g.multi); if (rc) { g.multi = nullptr; return rc; } }
  g.used = ++c->graph_clock;
  c->graph_cur = hit; c->step_graph = g.step; c->multi_graph = g.multi;
  return TGX_OK;
}

// ---- paged KV: block assignment (include/tgx.h "kv.budget_tokens").  Host-side free list; the device tables are updated by a small launch that carries the
// new entries BY VALUE (stream-ordered behind the launches that still read the old ones; no host buffer has to outlive the call).
struct KvTblUpdate { int n; int idx[16]; int val[16]; };
static __global__ void kv_tbl_set_kernel(int* tbl, KvTblUpdate u) { if ((int)threadIdx.x < u.n) tbl[u.idx[threadIdx.x]] = u.val[threadIdx.x]; }

static void kv_tbl_push(tgx_ctx* c, const std::vector<std::pair<int, int>>& changes) {      // (flat table index, value)
  for (size_t i = 0; i < changes.size(); i += 16) {
    KvTblUpdate u{};
    u.n = (int)std::min<size_t>(16, changes.size() - i);
    for (int k = 0; k < u.n; k++) { u.idx[k] = changes[i + (size_t)k].first; u.val[k] = changes[i + (size_t)k].second; c->kv_tbl_host[(size_t)u.idx[k]] = u.val[k]; }
    hipLaunchKernelGGL(kv_tbl_set_kernel, dim3(1), dim3(64), 0, c->stream, c->kv_tbl, u);
  }
}

int kv_ensure_blocks(tgx_ctx* c, int row, long long tokens) {
  if (!c->kv_paged) return TGX_OK;
  const int need = (int)((tokens + tgx::KV_BLOCK - 1) / tgx::KV_BLOCK);
  int& have = c->kv_row_nblk[(size_t)row];
  if (need <= have) return TGX_OK;
  if (need > c->kv_tbl_stride) return set_err(c, TGX_ERR_CONTEXT, "context size exceeded: row %d wants %lld tokens (contextSize %d)", row, tokens, c->d.max_ctx);
  if ((size_t)(need - have) > c->kv_free.size())
    return set_err(c, TGX_ERR_CONTEXT, "KV budget exhausted: row %d needs %d more blocks of %d tokens, %zu free of %d (option kv.budget_tokens = %d)", row, need - have,
                   tgx::KV_BLOCK, c->kv_free.size(), c->kv_nblocks - 1, c->kv_budget_tokens);
  std::vector<std::pair<int, int>> ch;
  for (; have < need; have++) { ch.emplace_back(row * c->kv_tbl_stride + have, c->kv_free.back()); c->kv_free.pop_back(); }
  kv_tbl_push(c, ch);
  return TGX_OK;
}

static void kv_release_row(tgx_ctx* c, int row) {       // the row's blocks back to the free list; its table entries back to the scratch block
  if (!c->kv_paged) return;
  int& have = c->kv_row_nblk[(size_t)row];
  std::vector<std::pair<int, int>> ch;
  for (int b = 0; b < have; b++) { const int i = row * c->kv_tbl_stride + b; c->kv_free.push_back(c->kv_tbl_host[(size_t)i]); ch.emplace_back(i, 0); }
  have = 0;
  kv_tbl_push(c, ch);
}

// tgx_read_probs evaluates a row's final probabilities on demand: remember what its last sampled step was configured with
static void note_sampled(tgx_ctx* c, int row0, int R, const tgx_sampler_cfg& cfg) {
  if (c->row_probs_cfg.size() < (size_t)c->d.max_batch) { c->row_probs_cfg.resize((size_t)c->d.max_batch); c->row_probs_ok.assign((size_t)c->d.max_batch, 0); }
  const bool sampled = !is_greedy(&cfg);
  for (int b = row0; b < row0 + R; b++) { c->row_probs_cfg[(size_t)b] = cfg; c->row_probs_ok[(size_t)b] = sampled ? 1 : 0; }
  c->have_probs = false;
  for (int b = 0; b < c->batch; b++) c->have_probs = c->have_probs || c->row_probs_ok[(size_t)b];
}

static int run_decode_steps(tgx_ctx* c, const tgx_sampler_cfg& cfg, uint64_t seed, int n) {
  if (c->kv_paged)       // every live row's next n positions have a block before the steps that write them are enqueued
    for (int b = 0; b < c->batch; b++)
      if (!c->row_idle[(size_t)b]) { int rc = kv_ensure_blocks(c, b, c->row_past[(size_t)b] + n); if (rc) return rc; }
  if (!is_greedy(&cfg)) {
    // the engine passes one seed for a whole generation (the draw mixes in position and row): only a CHANGED seed is copied — and that
    // copy must drain the stream, because steps already enqueued still read the old word.  With an unchanged seed tgx_step_async returns
    // without waiting for the previous step (the one-step lookahead of generateAsync, GPTEngine.cpp:196-217)
    if (!c->seed_valid || c->seed_on_dev != (unsigned long long)seed) {
      HIP_OK(c, hipStreamSynchronize(c->stream));
      const unsigned long long s = seed;
      HIP_OK(c, hipMemcpy(c->seed_dev, &s, 8, hipMemcpyHostToDevice));
      c->seed_on_dev = s; c->seed_valid = true;
    }
  }
  note_sampled(c, 0, c->batch, cfg);
  if (decode_mfma_ok(c)) {   // the batched step's workspace must exist before the step is captured
    int rc = ensure_skinny_ws(c, std::min(c->decode_step_rows, c->batch));
    if (rc) return rc;
  }
  // a retired row rides along from wherever its position word stands (row_past mirrors it); the attention form and the capacity check are chosen for the
  // longest LIVE row, so an idle row that has outrun it (the longer rows were retired since) restarts from position 0
  for (int b = 0; b < c->batch; b++)
    if (c->row_idle[(size_t)b] && c->row_past[(size_t)b] > c->past) {
      HIP_OK(c, hipMemsetAsync(c->rows[(size_t)b].pos, 0, 4, c->stream));
      c->row_past[(size_t)b] = 0;
    }
  // The attention form depends on the context (four-wave direct / sixteen-wave direct / split + combine / matrix cores): a call that crosses a limit is
  // issued in chunks, each on the form of its own contexts, from the cache of captured graphs
  int remaining = n;
  while (remaining > 0) {
    int m = remaining;
    const long long lims[2] = {nw4_limit(c, c->batch, true), direct_limit(c, c->batch, true)};
    for (long long lim : lims)
      if (lim > 0 && c->past + 1 <= lim && c->past + m > lim) m = (int)(lim - c->past);
    update_attn_modes(c, m);
    if (c->use_graph) {
      const int K = c->graph_steps;
      // any multi-step call captures the K-step graph as well (a short warm-up call then leaves nothing to capture inside a later, longer
      // call); one-step streaming calls never pay for it
      int rc = ensure_step_graph(c, cfg, /*want_multi=*/m >= 2);
      if (rc) return rc;
      int i = 0;
      if (c->multi_graph) for (; i + K <= m; i += K) HIP_OK(c, hipGraphLaunch(c->multi_graph, c->stream));
      for (; i < m; i++) HIP_OK(c, hipGraphLaunch(c->step_graph, c->stream));
    } else {
      for (int i = 0; i < m; i++) launch_decode_step(c, cfg);
      HIP_OK(c, hipGetLastError());
      if (c->launch_fault) { (void)hipStreamSynchronize(c->stream); c->poisoned = true; }      // the kernels issued before the fault advanced the device-side state
      LAUNCH_OK(c);
    }
    c->past += m;
    for (int b = 0; b < c->batch; b++) c->row_past[(size_t)b] += m;
    c->steps_issued += m;
    remaining -= m;
  }
  return TGX_OK;
}

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int tgx_abi_version(void) { return TGX_ABI_VERSION; }

int tgx_device_count(int* out_count) {
  if (!out_count) return TGX_ERR_INVALID;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *out_count = 0; return set_err(nullptr, TGX_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *out_count = n;
  return TGX_OK;
}

const char* tgx_last_error(const tgx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int tgx_create(const tgx_model_desc* desc, int device_ordinal, tgx_ctx** out_ctx) {
  if (!desc || !out_ctx) return set_err(nullptr, TGX_ERR_INVALID, "null argument");
  *out_ctx = nullptr;
  const tgx_model_desc& d = *desc;
  if (d.family != TGX_FAMILY_LLAMA && d.family != TGX_FAMILY_QWEN2 && d.family != TGX_FAMILY_MISTRAL && d.family != TGX_FAMILY_QWEN3 && d.family != TGX_FAMILY_GPT2)
    return set_err(nullptr, TGX_ERR_UNSUPPORTED, "family %d is not implemented on mi355x (gpt2/llama/qwen2/qwen3/mistral are)", d.family);
  const bool gpt2 = d.family == TGX_FAMILY_GPT2;
  if (gpt2 && (d.kv_heads != d.heads || d.heads * d.head_dim != d.hidden)) return set_err(nullptr, TGX_ERR_INVALID, "gpt2: n_head * head_dim must equal n_embd, no grouped heads");
  if (gpt2 && d.hidden > 2048) return set_err(nullptr, TGX_ERR_UNSUPPORTED, "gpt2: n_embd %d > 2048 (the LayerNorm-fused launches are built for up to 4 slices per lane)", d.hidden);
  if (gpt2 && d.n_positions < d.max_ctx) return set_err(nullptr, TGX_ERR_INVALID, "gpt2: n_positions %d < context size %d", d.n_positions, d.max_ctx);
  if (gpt2 && d.qk_norm) return set_err(nullptr, TGX_ERR_INVALID, "gpt2 has no q/k norm");
  if (d.compute_dtype != TGX_BF16 && d.compute_dtype != TGX_F16 && d.compute_dtype != TGX_F32) return set_err(nullptr, TGX_ERR_INVALID, "unknown compute dtype %d", d.compute_dtype);
  if (d.head_dim != 64 && d.head_dim != 128) return set_err(nullptr, TGX_ERR_UNSUPPORTED, "head_dim %d (64 and 128 are built)", d.head_dim);
  if (d.heads <= 0 || d.kv_heads <= 0 || d.heads % d.kv_heads) return set_err(nullptr, TGX_ERR_INVALID, "heads %% kv_heads != 0");
  if (d.heads / d.kv_heads > 16) return set_err(nullptr, TGX_ERR_UNSUPPORTED, "GQA group size %d > 16", d.heads / d.kv_heads);
  if (d.hidden % 8 || d.inter % 8 || (d.heads * d.head_dim) % 8) return set_err(nullptr, TGX_ERR_UNSUPPORTED, "hidden/intermediate sizes must be multiples of 8");
  if (d.hidden <= 0 || d.layers <= 0 || d.inter <= 0 || d.vocab <= 0 || d.max_ctx <= 0) return set_err(nullptr, TGX_ERR_INVALID, "non-positive model dimension");
  // a launch keeps its K range in registers: at most 8 slices of 8 elements per lane and 4 waves per row pair = 16384 elements
  if (d.hidden > 16384) return set_err(nullptr, TGX_ERR_UNSUPPORTED, "hidden_size %d > 16384", d.hidden);
  if (d.inter > 65536 || d.heads * d.head_dim > 16384) return set_err(nullptr, TGX_ERR_UNSUPPORTED, "projection input wider than 65536 (intermediate) / 16384 (heads * head_dim)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(nullptr, TGX_ERR_DEVICE, "no HIP device visible (--device mi355x needs a GPU; there is no CPU fallback)");
  if (device_ordinal < 0 || device_ordinal >= ndev) return set_err(nullptr, TGX_ERR_INVALID, "device ordinal %d out of range [0,%d)", device_ordinal, ndev);

  tgx_ctx* c = new (std::nothrow) tgx_ctx();
  if (!c) return set_err(nullptr, TGX_ERR_NOMEM, "host allocation failed");
  *out_ctx = c;
  c->d = d;
  c->gpt2 = gpt2;
  if (gpt2) { c->d.tied = 1; c->d.qkv_bias = 1; }     // the head is wte (ModelGPT2.h:170-176); every Conv1D has a bias
  if (c->d.max_batch < 1) c->d.max_batch = 1;
  c->device = device_ordinal;
  c->dt = d.compute_dtype == TGX_BF16 ? tgx::DT_BF16 : (d.compute_dtype == TGX_F16 ? tgx::DT_F16 : tgx::DT_F32);
  c->esz = d.compute_dtype == TGX_F32 ? 4 : 2;
  HIP_OK(c, hipSetDevice(device_ordinal));
  hipDeviceProp_t prop;
  HIP_OK(c, hipGetDeviceProperties(&prop, device_ordinal));
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_OK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  if (const char* e = getenv("TGX_NO_GRAPH")) c->use_graph = !(e[0] == '1');
  // measured crossover of the direct and the split attention (tools/sweep.py --grid attn.direct_max=0,100000): context ~850-1100 at
  // head_dim 64 (Qwen2.5-0.5B, Llama-3.2-1B), ~500 at 128 (Mistral-7B: half the tokens per wave-load)
  // round 4 (the split form lost its combine launch to the K-sliced o_proj, the 16-wave direct form its per-wave merge): Llama-3.2-1B 0.652 / 0.654 ms/token
  // split / direct at context 700, 0.654 / 0.667 at 1000; Qwen2.5-0.5B (2 kv heads: few split workgroups) 0.591 / 0.586 at 700, 0.591 / 0.600 at 1000
  c->attn_direct_max = d.head_dim == 64 ? (d.kv_heads >= 8 ? 576 : 832) : 384;
  // four waves per head up to 256 keys at head_dim 64 (Qwen2.5-0.5B 16-token prompt 0.579 -> 0.565 ms/token, Llama-3.2-1B 0.648 -> 0.634; from ~256 keys and at head_dim 128 the sixteen-wave form is ahead)
  c->attn_direct_nw4 = d.head_dim == 64 ? 192 : 0;      // (round 4: equal at ~220 keys, the sixteen-wave form ahead from ~300)
  // very short prompts: ONE pass through the batched decode kernels (4 positions) still beats the skinny MFMA prefill on small models
  // (Llama-3.2-1B: S = 4 1.00 vs 1.07 ms, S = 5 1.55 vs 1.07; Mistral-7B S = 4 4.65 vs 4.09) — tools/prefill_crossover.py, profiles/r02_prefill_short.txt
  c->prefill_min_rows = d.hidden > 2048 ? 4 : 5;
  c->tune[TGX_KERNEL_DOWN].ks = d.inter >= 8192 ? 4 : 2;   // K = intermediate_size: 4 waves split each row pair (2 below 8192 columns: Qwen2.5-0.5B 0.549 -> 0.542 ms/token, tools/sweep.py round 4)
  // qkv is the most latency-bound launch (few rows): 4 waves per row pair shorten every wave's load -> reduce chain; measured
  // ks 1 -> 4: Llama-3.2-1B 1395 -> 1411 tok/s, 3B 628 -> 637, Mistral-7B 341 -> 347; hidden 896 (Qwen2.5-0.5B) loses 2 %
  c->tune[TGX_KERNEL_QKV].ks = d.hidden >= 2048 ? 4 : 1;
  // fp32 storage (round 4, tools/sweep.py --dtype fp32): a row pair is twice the bytes — two waves per pair for gate_up / c_fc and o_proj: GPT-2 124M 0.3205 -> 0.3133 ms per
  // token (with qkv as well: 0.3104), Llama-3.2-1B 1.145 -> 1.124, Qwen2.5-0.5B 0.752 -> 0.739; 16-bit storage: no gain
  if (c->dt == tgx::DT_F32) {
    c->tune[TGX_KERNEL_GATEUP].ks = 2; c->tune[TGX_KERNEL_OPROJ].ks = 2;
    if (c->gpt2) c->tune[TGX_KERNEL_QKV].ks = 2;
  }

  const int H = d.hidden, I = d.inter, V = d.vocab, qd = d.heads * d.head_dim, kvd = d.kv_heads * d.head_dim;
  int rc;
  const size_t es = c->esz;
  if ((rc = dev_alloc(c, &c->embed, (size_t)V * H * es))) return rc;
  if (!c->d.tied && (rc = dev_alloc(c, &c->lm_head, (size_t)V * H * es))) return rc;
  if ((rc = dev_alloc(c, &c->final_norm, (size_t)H * es))) return rc;
  if (gpt2 && ((rc = dev_alloc(c, &c->wpe, (size_t)d.n_positions * H * es)) || (rc = dev_alloc(c, &c->final_norm_b, (size_t)H * es)))) return rc;
  c->L.resize((size_t)d.layers);
  for (auto& w : c->L) {
    if ((rc = dev_alloc(c, &w.in_norm, (size_t)H * es))) return rc;
    if ((rc = dev_alloc(c, &w.post_norm, (size_t)H * es))) return rc;
    if ((rc = dev_alloc(c, &w.wqkv, (size_t)(qd + 2 * kvd) * H * es))) return rc;
    if (c->d.qkv_bias && (rc = dev_alloc(c, &w.bqkv, (size_t)(qd + 2 * kvd) * es))) return rc;
    if ((rc = dev_alloc(c, &w.wo, (size_t)H * qd * es))) return rc;
    if (d.qk_norm && ((rc = dev_alloc(c, &w.q_norm, (size_t)d.head_dim * es)) || (rc = dev_alloc(c, &w.k_norm, (size_t)d.head_dim * es)))) return rc;
    if ((rc = dev_alloc(c, &w.wgu, (size_t)(gpt2 ? 1 : 2) * I * H * es))) return rc;
    if ((rc = dev_alloc(c, &w.wdown, (size_t)H * I * es))) return rc;
    if (gpt2 && ((rc = dev_alloc(c, &w.in_norm_b, (size_t)H * es)) || (rc = dev_alloc(c, &w.post_norm_b, (size_t)H * es)) || (rc = dev_alloc(c, &w.bo, (size_t)H * es)) ||
                 (rc = dev_alloc(c, &w.bfc, (size_t)I * es)) || (rc = dev_alloc(c, &w.bdown, (size_t)H * es)))) return rc;
  }
  return TGX_OK;
}

int tgx_upload(tgx_ctx* c, const char* name, const void* host, const int64_t* shape, int nd, int src_dtype) {
  if (!c || !name || !host || !shape) return c ? set_err(c, TGX_ERR_INVALID, "null argument") : TGX_ERR_INVALID;
  HIP_OK(c, hipSetDevice(c->device));
  const tgx_model_desc& d = c->d;
  const int64_t H = d.hidden, I = d.inter, V = d.vocab, qd = (int64_t)d.heads * d.head_dim, kvd = (int64_t)d.kv_heads * d.head_dim;
  auto bad_shape = [&]() { return set_err(c, TGX_ERR_SHAPE, "shape not equal for tensor: %s", name); };
  if (c->gpt2) return upload_gpt2(c, name, host, shape, nd, src_dtype);
  if (!strcmp(name, "model.embed_tokens.weight")) {
    if (!shape_is(shape, nd, V, H)) return bad_shape();
    c->embed_ok = true;
    return upload_param(c, c->embed, host, V * H, src_dtype);
  }
  if (!strcmp(name, "lm_head.weight")) {
    if (!shape_is(shape, nd, V, H)) return bad_shape();
    if (d.tied) return TGX_OK;   // aliased to embed_tokens (GPTModel.h:39-41)
    c->lm_head_ok = true;
    return upload_param(c, c->lm_head, host, V * H, src_dtype);
  }
  if (!strcmp(name, "model.norm.weight")) {
    if (!shape_is(shape, nd, H, -1)) return bad_shape();
    c->final_norm_ok = true;
    return upload_param(c, c->final_norm, host, H, src_dtype);
  }
  int l = -1;
  char rest[128] = {0};
  if (sscanf(name, "model.layers.%d.%127s", &l, rest) == 2 && l >= 0 && l < d.layers) {
    LayerW& w = c->L[(size_t)l];
    if (!strcmp(rest, "input_layernorm.weight")) {
      if (!shape_is(shape, nd, H, -1)) return bad_shape();
      w.in_norm_ok = true;
      return upload_param(c, w.in_norm, host, H, src_dtype);
    }
    if (d.qk_norm && (!strcmp(rest, "self_attn.q_norm.weight") || !strcmp(rest, "self_attn.k_norm.weight"))) {
      if (!shape_is(shape, nd, d.head_dim, -1)) return bad_shape();
      const bool isq = rest[10] == 'q';
      (isq ? w.q_norm_ok : w.k_norm_ok) = true;
      return upload_param(c, isq ? w.q_norm : w.k_norm, host, d.head_dim, src_dtype);
    }
    if (!strcmp(rest, "post_attention_layernorm.weight")) {
      if (!shape_is(shape, nd, H, -1)) return bad_shape();
      w.post_norm_ok = true;
      return upload_param(c, w.post_norm, host, H, src_dtype);
    }
    // MergedLinear row slices (Linear.h:64-79): [q | k | v] and [gate | up]
    struct Slot { const char* n; ebyte* base; ebyte* bias; int64_t row0, rows, cols; int kind; int bit; };
    const Slot slots[] = {
        {"self_attn.q_proj", w.wqkv, w.bqkv, 0, qd, H, 0, 0},        {"self_attn.k_proj", w.wqkv, w.bqkv, qd, kvd, H, 0, 1},
        {"self_attn.v_proj", w.wqkv, w.bqkv, qd + kvd, kvd, H, 0, 2}, {"self_attn.o_proj", w.wo, nullptr, 0, H, qd, 1, -1},
        {"mlp.gate_proj", w.wgu, nullptr, 0, I, H, 2, 3},             {"mlp.up_proj", w.wgu, nullptr, I, I, H, 2, 4},
        {"mlp.down_proj", w.wdown, nullptr, 0, H, I, 3, -1}};
    for (const Slot& s : slots) {
      const size_t ln = strlen(s.n);
      if (strncmp(rest, s.n, ln) || rest[ln] != '.') continue;
      if (!strcmp(rest + ln + 1, "weight")) {
        if (!shape_is(shape, nd, s.rows, s.cols)) return bad_shape();
        int rc = upload_param(c, s.base + (size_t)(s.row0 * s.cols) * c->esz, host, s.rows * s.cols, src_dtype);
        if (rc) return rc;
        if (s.bit >= 0) w.merged_filled |= 1 << s.bit;
        else if (s.kind == 1) w.wo_ok = true;
        else w.wdown_ok = true;
        return TGX_OK;
      }
      if (!strcmp(rest + ln + 1, "bias") && s.bias) {
        if (!shape_is(shape, nd, s.rows, -1)) return bad_shape();
        w.merged_filled |= 1 << (5 + s.bit);
        return upload_param(c, s.bias + (size_t)s.row0 * c->esz, host, s.rows, src_dtype);
      }
    }
  }
  return set_err(c, TGX_ERR_NAME, "Unexpected key: %s", name);
}

int tgx_finalize(tgx_ctx* c) {
  if (!c) return TGX_ERR_INVALID;
  HIP_OK(c, hipSetDevice(c->device));
  const tgx_model_desc& d = c->d;
  const int H = d.hidden, I = d.inter, V = d.vocab, hd = d.head_dim, qd = d.heads * hd, kvd = d.kv_heads * hd;
  if (c->gpt2) {
    if (!c->embed_ok) return set_err(c, TGX_ERR_STATE, "Missing key: wte.weight");
    if (!c->wpe_ok) return set_err(c, TGX_ERR_STATE, "Missing key: wpe.weight");
    if (!c->final_norm_ok || !c->final_norm_b_ok) return set_err(c, TGX_ERR_STATE, "Missing key: ln_f.%s", c->final_norm_ok ? "bias" : "weight");
    for (int l = 0; l < d.layers; l++)
      if (c->L[(size_t)l].gpt2_filled != 0xfff) return set_err(c, TGX_ERR_STATE, "Missing key in h.%d", l);
  }
  if (!c->gpt2 && !c->embed_ok) return set_err(c, TGX_ERR_STATE, "Missing key: model.embed_tokens.weight");
  if (!d.tied && !c->lm_head_ok) return set_err(c, TGX_ERR_STATE, "Missing key: lm_head.weight");
  if (!c->gpt2 && !c->final_norm_ok) return set_err(c, TGX_ERR_STATE, "Missing key: model.norm.weight");
  for (int l = 0; l < d.layers && !c->gpt2; l++) {
    const LayerW& w = c->L[(size_t)l];
    static const char* merged_names[8] = {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
                                          "self_attn.q_proj.bias", "self_attn.k_proj.bias", "self_attn.v_proj.bias"};
    for (int b = 0; b < (d.qkv_bias ? 8 : 5); b++)
      if (!(w.merged_filled & (1 << b))) return set_err(c, TGX_ERR_STATE, "Missing key: model.layers.%d.%s", l, merged_names[b]);
    if (!w.in_norm_ok || !w.post_norm_ok || !w.wo_ok || !w.wdown_ok)
      return set_err(c, TGX_ERR_STATE, "Missing key: model.layers.%d.%s", l, !w.in_norm_ok ? "input_layernorm.weight" : !w.post_norm_ok ? "post_attention_layernorm.weight" : !w.wo_ok ? "self_attn.o_proj.weight" : "mlp.down_proj.weight");
    if (d.qk_norm && (!w.q_norm_ok || !w.k_norm_ok)) return set_err(c, TGX_ERR_STATE, "Missing key: model.layers.%d.self_attn.{q,k}_norm.weight", l);
  }
  if (c->finalized) return TGX_OK;

  std::vector<float> cs, sn;
  if (c->gpt2) {      // no rotary embedding: the qkv epilogue's rotation becomes the identity
    cs.assign((size_t)d.max_ctx * (hd / 2), 1.0f);
    sn.assign((size_t)d.max_ctx * (hd / 2), 0.0f);
  } else build_rope_host(d, cs, sn);
  int rc;
  if ((rc = dev_alloc(c, &c->rope_cos, cs.size()))) return rc;
  if ((rc = dev_alloc(c, &c->rope_sin, sn.size()))) return rc;
  HIP_OK(c, hipMemcpy(c->rope_cos, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(c, hipMemcpy(c->rope_sin, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));

  c->lm_grid = gemv_grid(c, (V + 1) / 2, 1, c->tune[TGX_KERNEL_LMHEAD].bpc);
  int ns = c->num_cus / d.kv_heads;
  c->attn_nsplit = ns < 1 ? 1 : (ns > 32 ? 32 : ns);
  if (c->attn_nsplit_opt > 0) c->attn_nsplit = c->attn_nsplit_opt;

  c->rows.resize((size_t)d.max_batch);
  const size_t B = (size_t)d.max_batch;
  size_t kv_elems = (size_t)d.layers * d.kv_heads * d.max_ctx * hd;
  c->kv_row_elems = kv_elems;
  size_t kv_total = B * kv_elems;                     // elements of the K cache (and of the V cache)
  if (c->kv_budget_tokens > 0) {                      // paged KV: pools of KV_BLOCK-token blocks shared by the rows (kernels/common.h)
    if (c->dt == tgx::DT_F32) return set_err(c, TGX_ERR_UNSUPPORTED, "kv.budget_tokens: paged KV serves the 16-bit storage dtypes");
    c->kv_paged = true;
    c->kv_nblocks = (c->kv_budget_tokens + tgx::KV_BLOCK - 1) / tgx::KV_BLOCK + 1;       // + the scratch block 0
    c->kv_tbl_stride = (d.max_ctx + tgx::KV_BLOCK - 1) / tgx::KV_BLOCK;
    kv_total = (size_t)d.layers * c->kv_nblocks * d.kv_heads * tgx::KV_BLOCK * hd;
    c->kv_row_elems = 1;                              // "the rows are separate sequences" wherever a row stride is tested against 0; never used as a stride (decode.hip kvs)
    if ((rc = dev_alloc(c, &c->kv_tbl, B * (size_t)c->kv_tbl_stride))) return rc;
    HIP_OK(c, hipMemset(c->kv_tbl, 0, B * (size_t)c->kv_tbl_stride * 4));
    c->kv_tbl_host.assign(B * (size_t)c->kv_tbl_stride, 0);
    c->kv_row_nblk.assign(B, 0);
    c->kv_free.clear();
    for (int b = c->kv_nblocks - 1; b >= 1; b--) c->kv_free.push_back(b);
  }
  c->attn_part_row = (size_t)d.heads * c->attn_nsplit * (hd + 4);
  if ((rc = dev_alloc(c, &c->slab_x, B * H))) return rc;
  if ((rc = dev_alloc(c, &c->slab_q, B * qd))) return rc;
  if ((rc = dev_alloc(c, &c->slab_kraw, B * kvd))) return rc;
  if ((rc = dev_alloc(c, &c->slab_attn, B * qd))) return rc;
  if ((rc = dev_alloc(c, &c->slab_h, B * I))) return rc;
  if ((rc = dev_alloc(c, &c->slab_logits, B * V))) return rc;
  if ((rc = dev_alloc(c, &c->slab_probs, B * V))) return rc;
  if ((rc = dev_alloc(c, &c->slab_part_val, B * c->lm_grid))) return rc;
  if ((rc = dev_alloc(c, &c->slab_part_idx, B * c->lm_grid))) return rc;
  if ((rc = dev_alloc(c, &c->slab_attn_part, B * c->attn_part_row))) return rc;
  if ((rc = dev_alloc(c, &c->slab_tok, B))) return rc;
  if ((rc = dev_alloc(c, &c->slab_pos, B))) return rc;
  if ((rc = dev_alloc(c, &c->slab_prompt, B * d.max_ctx))) return rc;
  if ((rc = dev_alloc(c, &c->slab_k, kv_total * c->esz))) return rc;
  if ((rc = dev_alloc(c, &c->slab_v, kv_total * c->esz))) return rc;
  HIP_OK(c, hipMemset(c->slab_tok, 0, B * 4));
  HIP_OK(c, hipMemset(c->slab_pos, 0, B * 4));
  HIP_OK(c, hipMemset(c->slab_k, 0, kv_total * c->esz));
  HIP_OK(c, hipMemset(c->slab_v, 0, kv_total * c->esz));
  for (size_t b = 0; b < B; b++) {
    RowState& r = c->rows[b];
    r.x = c->slab_x + b * H; r.q = c->slab_q + b * qd; r.k_raw = c->slab_kraw + b * kvd; r.attn = c->slab_attn + b * qd;
    r.h = c->slab_h + b * I; r.logits = c->slab_logits + b * V; r.probs = c->slab_probs + b * V;
    r.part_val = c->slab_part_val + b * c->lm_grid; r.part_idx = c->slab_part_idx + b * c->lm_grid;
    r.attn_part = c->slab_attn_part + b * c->attn_part_row;
    r.tok = c->slab_tok + b; r.pos = c->slab_pos + b; r.prompt = c->slab_prompt + b * d.max_ctx;
    r.kcache = c->slab_k + b * kv_elems * c->esz; r.vcache = c->slab_v + b * kv_elems * c->esz;
    if (c->kv_paged) { r.kcache = c->slab_k; r.vcache = c->slab_v; r.tbl = c->kv_tbl + b * (size_t)c->kv_tbl_stride; }
  }
  if ((rc = dev_alloc(c, &c->ch_x, 4 * (size_t)H))) return rc;
  if ((rc = dev_alloc(c, &c->ch_q, 4 * (size_t)qd))) return rc;
  if ((rc = dev_alloc(c, &c->ch_kraw, 4 * (size_t)kvd))) return rc;
  if ((rc = dev_alloc(c, &c->ch_attn, 4 * (size_t)qd))) return rc;
  if ((rc = dev_alloc(c, &c->ch_h, 4 * (size_t)I))) return rc;
  if ((rc = dev_alloc(c, &c->ch_part, 4 * c->attn_part_row))) return rc;
  if ((rc = dev_alloc(c, &c->ch_pos, 4))) return rc;
  for (size_t k = 0; k < 4; k++) {
    RowState& r = c->chunk[k];
    r.x = c->ch_x + k * H; r.q = c->ch_q + k * qd; r.k_raw = c->ch_kraw + k * kvd; r.attn = c->ch_attn + k * qd; r.h = c->ch_h + k * I;
    r.attn_part = c->ch_part + k * c->attn_part_row; r.pos = c->ch_pos + k;
  }
  c->log_cap = d.max_ctx > 1024 ? d.max_ctx : 1024;
  if ((rc = dev_alloc(c, &c->step, 1)) || (rc = dev_alloc(c, &c->step_done, 1))) return rc;
  if ((rc = dev_alloc(c, &c->seed_dev, 1))) return rc;
  if ((rc = sampler_alloc(c))) return rc;
  if ((rc = dev_alloc(c, &c->scratch_x, (size_t)H))) return rc;
  HIP_OK(c, hipMemset(c->scratch_x, 0, (size_t)H * 4));
  if ((rc = dev_alloc(c, &c->tok_log, (size_t)c->log_cap * d.max_batch))) return rc;
  HIP_OK(c, hipMemset(c->step, 0, 4));
  HIP_OK(c, hipMemset(c->step_done, 0, 4));
  HIP_OK(c, hipHostMalloc((void**)&c->host_ring, (size_t)HOST_RING * d.max_batch * 4, hipHostMallocMapped));
  HIP_OK(c, hipHostGetDevicePointer((void**)&c->host_ring_dev, c->host_ring, 0));
  for (int i = 0; i < MAX_TICKET_EVENTS; i++) HIP_OK(c, hipEventCreateWithFlags(&c->ticket_ev[i], hipEventDisableTiming));
  for (int i = 0; i < 2; i++) HIP_OK(c, hipEventCreate(&c->prof.ev[i]));
  if ((rc = prefill_set_attrs(c)) || (rc = skinny_set_attrs(c)) || (rc = attn_set_attrs(c))) return rc;
  // fixed-point accumulators of the K-sliced o_proj (kernels/oproj_sliced.h): zero between layers
  if ((rc = dev_alloc(c, &c->slab_acc, B * (size_t)H))) return rc;
  HIP_OK(c, hipMemset(c->slab_acc, 0, B * (size_t)H * 8));
  c->past = 0;
  c->row_past.assign(B, 0);
  c->row_tok.assign(B, 0);
  c->row_idle.assign(B, 0);
  c->finalized = true;
  return TGX_OK;
}

void tgx_destroy(tgx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  drop_step_graphs(c);
  auto fr = [](void* p) { if (p) (void)hipFree(p); };
  fr(c->embed); fr(c->lm_head); fr(c->final_norm); fr(c->wpe); fr(c->final_norm_b); fr(c->rope_cos); fr(c->rope_sin); fr(c->step); fr(c->step_done); fr(c->tok_log); fr(c->scratch_x); fr(c->seed_dev); fr(c->samp_scratch); fr(c->samp_list_comp); fr(c->samp_list_v);
  fr(c->slab_acc); fr(c->kv_tbl);
  fr(c->ch_x); fr(c->ch_q); fr(c->ch_kraw); fr(c->ch_attn); fr(c->ch_h); fr(c->ch_part); fr(c->ch_pos);
  fr(c->ws_x); fr(c->ws_out); fr(c->ws_ah); fr(c->ws_al); fr(c->ws_al2); fr(c->ws_qh); fr(c->ws_ql); fr(c->ws_hh); fr(c->ws_hl); fr(c->ws_part); fr(c->ws_ssq); fr(c->ws_pos);
  for (auto& w : c->L) { fr(w.in_norm); fr(w.post_norm); fr(w.wqkv); fr(w.bqkv); fr(w.wo); fr(w.q_norm); fr(w.k_norm); fr(w.wgu); fr(w.wdown); fr(w.in_norm_b); fr(w.post_norm_b); fr(w.bo); fr(w.bfc); fr(w.bdown); }
  fr(c->slab_x); fr(c->slab_q); fr(c->slab_kraw); fr(c->slab_attn); fr(c->slab_h); fr(c->slab_logits); fr(c->slab_probs);
  fr(c->slab_part_val); fr(c->slab_part_idx); fr(c->slab_attn_part); fr(c->slab_tok); fr(c->slab_pos); fr(c->slab_prompt); fr(c->slab_k); fr(c->slab_v);
  if (c->host_ring) (void)hipHostFree(c->host_ring);
  for (auto& e : c->ticket_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : c->prof.ev) if (e) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int tgx_forward(tgx_ctx* c, const int64_t* ids, int batch, int seq) {
  if (!c || !ids) return c ? set_err(c, TGX_ERR_INVALID, "null argument") : TGX_ERR_INVALID;
  if (!c->finalized) return set_err(c, TGX_ERR_STATE, "forward before finalize");
  if (batch < 1 || batch > c->d.max_batch || seq < 1) return set_err(c, TGX_ERR_INVALID, "batch/seq out of range");
  if (c->poisoned) return set_err(c, TGX_ERR_STATE, "an earlier pass failed half-way: call tgx_reset_cache first");
  if (seq > 1 && c->past > 0) return set_err(c, TGX_ERR_INVALID, "seq>1 with pastLength>0");
  if (c->past + seq > c->d.max_ctx) return set_err(c, TGX_ERR_CONTEXT, "context size exceeded: %lld + %d > %d", (long long)c->past, seq, c->d.max_ctx);
  for (int b = 0; b < batch; b++)
    if (c->row_idle[(size_t)b] || c->row_past[(size_t)b] != c->past) return set_err(c, TGX_ERR_STATE, "tgx_forward on a batch whose rows differ in length (row %d: %lld, longest %lld): use tgx_decode / tgx_forward_row, or tgx_reset_cache", b, (long long)c->row_past[(size_t)b], (long long)c->past);
  for (int64_t i = 0; i < (int64_t)batch * seq; i++)
    if (ids[i] < 0 || ids[i] >= c->d.vocab) return set_err(c, TGX_ERR_INVALID, "token id out of range");
  HIP_OK(c, hipSetDevice(c->device));
  c->batch = batch;
  // matrix-core prefill: 16-bit storage through the split-term GEMMs (every family incl. GPT-2), fp32 storage through the f32-input MFMA
  const bool f32_path = c->dt == tgx::DT_F32 && seq >= c->prefill_f32_min_rows && c->prefill_mfma;
  const bool mfma_path = (f32_path || (seq >= c->prefill_min_rows && seq >= 4 && c->prefill_mfma && c->dt != tgx::DT_F32 && prefill_shapes_ok(c->d)));
  for (int b = 0; b < batch; b++) { int rc = kv_ensure_blocks(c, b, c->past + seq); if (rc) return rc; }
  for (int b = 0; b < batch; b++) HIP_OK(c, hipMemcpyAsync(c->rows[(size_t)b].prompt, ids + (size_t)b * seq, (size_t)seq * 8, hipMemcpyHostToDevice, c->stream));
  if (mfma_path) {
    // batched prefill on the matrix cores; logits for the last position only (== forward + narrow, GPTEngine.cpp:96-97).  Batch rows are
    // stacked into one row block while that stays within 8192 workspace rows (the CLI's 4 short prompts cost one pass over the weights)
    const int per = std::max(1, std::min(batch, 8192 / seq));
    for (int row0 = 0; row0 < batch; row0 += per) {
      const int nb = std::min(per, batch - row0);
      const bool skinny = !f32_path && !c->gpt2 && c->prefill_skinny && c->d.vocab >= 128 &&     // a few rows: the weight stream of a decode step
                          (nb * seq <= 32 ? c->prefill_skinny_rows >= nb * seq : (nb * seq <= c->prefill_skinny_rows && c->d.hidden <= c->prefill_skinny_hidden_max && (nb * seq <= 64 || (c->skinny_dma && c->d.hidden <= c->prefill_skinny_hidden_max_wide))));
      int rc = skinny ? ensure_skinny_ws(c, nb * seq) : ensure_prefill_ws(c, nb * seq);
      if (rc) return rc;
      if (f32_path && (rc = ensure_f32_part(c, nb * seq))) return rc;
      if (f32_path) launch_prefill_f32(c, row0, nb, seq);
      else if (skinny) launch_prefill_skinny(c, row0, nb, seq); else launch_prefill(c, row0, nb, seq, (int)c->past);   // tgx_forward: every row at the batch's pastLength
      for (int b = row0; b < row0 + nb;) {
        const int rem = row0 + nb - b, R = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
        launch_lm_head(c, b, R);
        b += R;
      }
      for (int b = row0; b < row0 + nb; b++) launch_add_pos(c, c->rows[(size_t)b].pos, seq);
    }
  }
  for (int b = 0; b < batch && !mfma_path; b++) {
    RowState& r = c->rows[(size_t)b];
    // prefill by steps (fp32 storage, GPT-2, prompts shorter than 4 tokens, shapes the GEMM tile does not cover): up to 4 consecutive
    // positions per pass through the decode kernels — the chunk rows share this row's cache (kv_stride 0), each attends the
    // keys up to its own position, so the result equals position-by-position passes at a quarter of the weight traffic
    update_attn_modes(c, seq, 1, /*step=*/false);      // the chunk rows of a pass are positions of ONE sequence
    for (int s0 = 0; s0 < seq;) {
      const int rem = seq - s0, R = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
      launch_embed_chunk(c, r.prompt + s0, R, (int)c->past + s0);
      for (int k = 0; k < R; k++) { c->chunk[k].kcache = r.kcache; c->chunk[k].vcache = r.vcache; c->chunk[k].tbl = r.tbl; }
      launch_layers(c, c->chunk, R, 0);
      s0 += R;
      if (s0 == seq) {                                   // the last position's hidden state feeds lm_head; publish token and length
        (void)hipMemcpyAsync(r.x, c->chunk[R - 1].x, (size_t)c->d.hidden * 4, hipMemcpyDeviceToDevice, c->stream);
        launch_add_pos(c, r.pos, seq);
        launch_lm_head(c, b, 1);
      }
    }
  }
  HIP_OK(c, hipGetLastError());
  if (c->launch_fault) { (void)hipStreamSynchronize(c->stream); c->poisoned = true; }   // nothing of a failed pass stays in flight (the prompt copies read the caller's buffer); the caches hold a partial pass
  LAUNCH_OK(c);
  HIP_OK(c, hipStreamSynchronize(c->stream));   // host `ids` may be pageable and reused by the caller
  c->past += seq;
  for (int b = 0; b < batch; b++) { c->row_past[(size_t)b] = c->past; c->row_tok[(size_t)b] = 0; c->row_idle[(size_t)b] = 0; }
  c->have_logits = true;
  c->have_token = false;
  c->have_probs = false;                          // the logits are new: no sampled step belongs to them yet
  std::fill(c->row_probs_ok.begin(), c->row_probs_ok.end(), 0);
  return TGX_OK;
}

int tgx_read_logits(tgx_ctx* c, float* out, int rounded) {
  if (!c || !out) return TGX_ERR_INVALID;
  if (!c->have_logits) return set_err(c, TGX_ERR_STATE, "no logits: call tgx_forward/tgx_decode first");
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  const size_t V = (size_t)c->d.vocab;
  for (int b = 0; b < c->batch; b++) HIP_OK(c, hipMemcpy(out + b * V, c->rows[(size_t)b].logits, V * 4, hipMemcpyDeviceToHost));
  if (rounded && c->dt != tgx::DT_F32)
    for (size_t i = 0; i < V * (size_t)c->batch; i++)
      out[i] = c->dt == tgx::DT_BF16 ? host_bf16_to_f32(host_f32_to_bf16(out[i])) : host_half_to_f32(host_f32_to_half(out[i]));
  return TGX_OK;
}

int tgx_sample(tgx_ctx* c, const tgx_sampler_cfg* cfg, uint64_t seed, int64_t* out_ids) {
  if (!c || !cfg) return TGX_ERR_INVALID;
  if (!c->have_logits) return set_err(c, TGX_ERR_STATE, "no logits to sample from");
  HIP_OK(c, hipSetDevice(c->device));
  if (!is_greedy(cfg)) {
    if (!c->seed_valid || c->seed_on_dev != (unsigned long long)seed) {
      HIP_OK(c, hipStreamSynchronize(c->stream));
      const unsigned long long s = seed;
      HIP_OK(c, hipMemcpy(c->seed_dev, &s, 8, hipMemcpyHostToDevice));
      c->seed_on_dev = s; c->seed_valid = true;
    }
  }
  note_sampled(c, 0, c->batch, *cfg);
  launch_sample(c, 0, c->batch, *cfg, /*advance_pos=*/false, /*log_step=*/false);
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipStreamSynchronize(c->stream));
  for (int b = 0; b < c->batch; b++) {
    int t = 0;
    HIP_OK(c, hipMemcpy(&t, c->rows[(size_t)b].tok, 4, hipMemcpyDeviceToHost));
    if (out_ids) out_ids[b] = t;
    if (b == 0) c->last_sampled0 = t;
    c->row_tok[(size_t)b] = 1;
  }
  c->have_token = true;
  return TGX_OK;
}

int tgx_decode(tgx_ctx* c, const tgx_sampler_cfg* cfg, uint64_t seed, int n_steps, int64_t* out_ids) {
  if (!c || !cfg || n_steps < 0) return TGX_ERR_INVALID;
  if (c->poisoned) return set_err(c, TGX_ERR_STATE, "an earlier pass failed half-way: call tgx_reset_cache first");
  if (!c->have_token) return set_err(c, TGX_ERR_STATE, "decode needs a current token: call tgx_sample after tgx_forward");
  if (c->past + n_steps > c->d.max_ctx) return set_err(c, TGX_ERR_CONTEXT, "context size exceeded: %lld + %d > %d", (long long)c->past, n_steps, c->d.max_ctx);
  if (n_steps > c->log_cap) return set_err(c, TGX_ERR_INVALID, "n_steps exceeds the token log capacity %d", c->log_cap);
  HIP_OK(c, hipSetDevice(c->device));
  const int64_t start = c->steps_issued;
  int rc = run_decode_steps(c, *cfg, seed, n_steps);
  if (rc) return rc;
  c->have_logits = true;
  if (out_ids && n_steps > 0) {
    const size_t B = (size_t)c->batch;
    std::vector<int> tmp((size_t)n_steps * B);
    const int64_t s0 = start % c->log_cap;
    const int64_t first = (s0 + n_steps <= c->log_cap) ? n_steps : c->log_cap - s0;
    HIP_OK(c, hipMemcpyAsync(tmp.data(), c->tok_log + (size_t)s0 * B, (size_t)first * B * 4, hipMemcpyDeviceToHost, c->stream));
    if (first < n_steps)
      HIP_OK(c, hipMemcpyAsync(tmp.data() + (size_t)first * B, c->tok_log, (size_t)(n_steps - first) * B * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < tmp.size(); i++) out_ids[i] = tmp[i];
    c->last_sampled0 = tmp[(size_t)(n_steps - 1) * B];
  }
  return TGX_OK;
}

int tgx_step_async(tgx_ctx* c, const tgx_sampler_cfg* cfg, uint64_t seed, int64_t* out_ticket) {
  if (!c || !cfg || !out_ticket) return TGX_ERR_INVALID;
  if (c->poisoned) return set_err(c, TGX_ERR_STATE, "an earlier pass failed half-way: call tgx_reset_cache first");
  if (!c->have_token) return set_err(c, TGX_ERR_STATE, "step needs a current token: call tgx_sample after tgx_forward");
  if (c->past + 1 > c->d.max_ctx) return set_err(c, TGX_ERR_CONTEXT, "context size exceeded");
  HIP_OK(c, hipSetDevice(c->device));
  int rc = run_decode_steps(c, *cfg, seed, 1);
  if (rc) return rc;
  const int64_t ticket = c->steps_issued;
  HIP_OK(c, hipEventRecord(c->ticket_ev[ticket % MAX_TICKET_EVENTS], c->stream));
  c->have_logits = true;
  *out_ticket = ticket;
  return TGX_OK;
}

int tgx_fetch_token(tgx_ctx* c, int64_t ticket, int32_t* out_id) {
  if (!c || !out_id) return TGX_ERR_INVALID;
  if (ticket == 0) { *out_id = c->last_sampled0; return TGX_OK; }
  if (ticket < 0 || ticket > c->steps_issued || c->steps_issued - ticket >= MAX_TICKET_EVENTS)
    return set_err(c, TGX_ERR_INVALID, "ticket %lld is not outstanding", (long long)ticket);
  HIP_OK(c, hipEventSynchronize(c->ticket_ev[ticket % MAX_TICKET_EVENTS]));
  *out_id = c->host_ring[((ticket - 1) % HOST_RING) * c->batch];   // row 0 of the step that ticket names
  return TGX_OK;
}

int tgx_reset_cache(tgx_ctx* c) {
  if (!c) return TGX_ERR_INVALID;
  if (!c->finalized) return set_err(c, TGX_ERR_STATE, "reset before finalize");
  HIP_OK(c, hipSetDevice(c->device));
  for (auto& r : c->rows) HIP_OK(c, hipMemsetAsync(r.pos, 0, 4, c->stream));
  for (int b = 0; b < c->d.max_batch; b++) kv_release_row(c, b);
  if (c->slab_acc) HIP_OK(c, hipMemsetAsync(c->slab_acc, 0, (size_t)c->d.max_batch * c->d.hidden * 8, c->stream));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  c->past = 0;
  std::fill(c->row_past.begin(), c->row_past.end(), 0);
  std::fill(c->row_tok.begin(), c->row_tok.end(), 0);
  std::fill(c->row_idle.begin(), c->row_idle.end(), 0);
  c->have_logits = c->have_token = false;
  c->poisoned = false;
  return TGX_OK;
}

int64_t tgx_past_length(const tgx_ctx* c) { return c ? c->past : -1; }

// ---- per-row sequence lifecycle (include/tgx.h, ABI 3).  The step kernels read every row's position from its own device word; the host keeps the
// mirror row_past[] and `past` = the longest row of the batch (capacity checks, attention-form limits: a form chosen for the longest row is valid for
// the shorter ones — the direct form's pass count and the split form's active splits are derived on the device from each row's position).
// Retired rows (row_idle) are left out of both: nothing waits for them and they bound nothing (run_decode_steps keeps them below the longest live row).
static void refresh_longest(tgx_ctx* c) {
  int64_t m = 0;
  int live = 0;
  bool all = true;
  for (int b = 0; b < c->batch; b++) {
    if (c->row_idle[(size_t)b]) continue;
    live++;
    m = std::max(m, c->row_past[(size_t)b]);
    all = all && c->row_tok[(size_t)b];
  }
  c->past = m;
  c->have_token = live > 0 && all;
}

int64_t tgx_past_length_row(const tgx_ctx* c, int row) {
  if (!c || !c->finalized || row < 0 || row >= c->d.max_batch) return -1;
  return c->row_idle[(size_t)row] ? 0 : c->row_past[(size_t)row];        // a retired row holds no sequence, wherever its position word stands
}

int tgx_reset_row(tgx_ctx* c, int row) {
  if (!c) return TGX_ERR_INVALID;
  if (!c->finalized) return set_err(c, TGX_ERR_STATE, "reset before finalize");
  if (row < 0 || row >= c->d.max_batch) return set_err(c, TGX_ERR_INVALID, "row %d out of range [0,%d)", row, c->d.max_batch);
  if (c->poisoned) return set_err(c, TGX_ERR_STATE, "an earlier pass failed half-way: call tgx_reset_cache first");
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipMemsetAsync(c->rows[(size_t)row].pos, 0, 4, c->stream));     // stream-ordered behind the steps already enqueued
  kv_release_row(c, row);                                                     // paged KV: its blocks go back to the pool (a retired row rides along on the scratch block)
  c->row_past[(size_t)row] = 0;
  c->row_tok[(size_t)row] = 0;
  c->row_idle[(size_t)row] = row < c->batch;                                  // a live slot becomes a retired one: the batch keeps stepping without it
  refresh_longest(c);
  return TGX_OK;
}

int tgx_forward_row(tgx_ctx* c, int row, const int64_t* ids, int seq) {
  if (!c || !ids) return c ? set_err(c, TGX_ERR_INVALID, "null argument") : TGX_ERR_INVALID;
  if (!c->finalized) return set_err(c, TGX_ERR_STATE, "forward before finalize");
  if (c->poisoned) return set_err(c, TGX_ERR_STATE, "an earlier pass failed half-way: call tgx_reset_cache first");
  if (row < 0 || row >= c->d.max_batch || row > c->batch) return set_err(c, TGX_ERR_INVALID, "row %d: a live row [0,%d) or the next free one (max_batch %d)", row, c->batch, c->d.max_batch);
  if (seq < 1 || seq > c->d.max_ctx) return set_err(c, seq < 1 ? TGX_ERR_INVALID : TGX_ERR_CONTEXT, "seq %d out of range (context size %d)", seq, c->d.max_ctx);
  if (!c->row_idle[(size_t)row] && c->row_past[(size_t)row] != 0) return set_err(c, TGX_ERR_STATE, "row %d holds %lld positions: tgx_reset_row first", row, (long long)c->row_past[(size_t)row]);
  for (int i = 0; i < seq; i++)
    if (ids[i] < 0 || ids[i] >= c->d.vocab) return set_err(c, TGX_ERR_INVALID, "token id out of range");
  HIP_OK(c, hipSetDevice(c->device));
  // the prompt runs as a one-row pass of the same prefill paths tgx_forward takes (they address rows by index and read the pass's past from c->past)
  const int64_t longest = c->past;
  const int batch_before = c->batch;
  c->past = 0;
  const bool f32_path = c->dt == tgx::DT_F32 && seq >= c->prefill_f32_min_rows && c->prefill_mfma;
  const bool mfma_path = (f32_path || (seq >= c->prefill_min_rows && seq >= 4 && c->prefill_mfma && c->dt != tgx::DT_F32 && prefill_shapes_ok(c->d)));
  if (c->kv_paged) { kv_release_row(c, row); int rc0 = kv_ensure_blocks(c, row, seq); if (rc0) { c->past = longest; return rc0; } }
  RowState& r = c->rows[(size_t)row];
  int rc = TGX_OK;
  hipError_t e = hipMemcpyAsync(r.prompt, ids, (size_t)seq * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && c->row_past[(size_t)row] != 0) {                    // a retired row that rode along since its reset: back to position 0
    e = hipMemsetAsync(r.pos, 0, 4, c->stream);
    c->row_past[(size_t)row] = 0;
  }
  if (e == hipSuccess && mfma_path) {
    const bool skinny = !f32_path && !c->gpt2 && c->prefill_skinny && c->d.vocab >= 128 &&
                        (seq <= 32 ? c->prefill_skinny_rows >= seq : (seq <= c->prefill_skinny_rows && c->d.hidden <= c->prefill_skinny_hidden_max && (seq <= 64 || (c->skinny_dma && c->d.hidden <= c->prefill_skinny_hidden_max_wide))));
    rc = skinny ? ensure_skinny_ws(c, seq) : ensure_prefill_ws(c, seq);
    if (!rc && f32_path) rc = ensure_f32_part(c, seq);
    if (!rc) {
      if (f32_path) launch_prefill_f32(c, row, 1, seq);
      else if (skinny) launch_prefill_skinny(c, row, 1, seq); else launch_prefill(c, row, 1, seq, /*past=*/0);
      launch_lm_head(c, row, 1);
      launch_add_pos(c, r.pos, seq);
    }
  } else if (e == hipSuccess) {
    update_attn_modes(c, seq, 1, /*step=*/false);
    for (int s0 = 0; s0 < seq;) {
      const int rem = seq - s0, R = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
      launch_embed_chunk(c, r.prompt + s0, R, s0);
      for (int k = 0; k < R; k++) { c->chunk[k].kcache = r.kcache; c->chunk[k].vcache = r.vcache; c->chunk[k].tbl = r.tbl; }
      launch_layers(c, c->chunk, R, 0);
      s0 += R;
      if (s0 == seq) {
        (void)hipMemcpyAsync(r.x, c->chunk[R - 1].x, (size_t)c->d.hidden * 4, hipMemcpyDeviceToDevice, c->stream);
        launch_add_pos(c, r.pos, seq);
        launch_lm_head(c, row, 1);
      }
    }
  }
  c->past = longest;
  if (rc) return rc;
  HIP_OK(c, e);
  HIP_OK(c, hipGetLastError());
  if (c->launch_fault) { (void)hipStreamSynchronize(c->stream); c->poisoned = true; }
  LAUNCH_OK(c);
  HIP_OK(c, hipStreamSynchronize(c->stream));   // host `ids` may be pageable and reused by the caller
  c->batch = std::max(batch_before, row + 1);
  c->row_past[(size_t)row] = seq;
  c->row_tok[(size_t)row] = 0;
  c->row_idle[(size_t)row] = 0;
  refresh_longest(c);
  c->have_logits = true;
  if ((size_t)row < c->row_probs_ok.size()) c->row_probs_ok[(size_t)row] = 0;
  return TGX_OK;
}

int tgx_sample_row(tgx_ctx* c, int row, const tgx_sampler_cfg* cfg, uint64_t seed, int64_t* out_id) {
  if (!c || !cfg) return TGX_ERR_INVALID;
  if (!c->have_logits) return set_err(c, TGX_ERR_STATE, "no logits to sample from");
  if (row < 0 || row >= c->batch) return set_err(c, TGX_ERR_INVALID, "row %d out of range [0,%d)", row, c->batch);
  HIP_OK(c, hipSetDevice(c->device));
  if (!is_greedy(cfg)) {
    if (!c->seed_valid || c->seed_on_dev != (unsigned long long)seed) {
      HIP_OK(c, hipStreamSynchronize(c->stream));
      const unsigned long long s = seed;
      HIP_OK(c, hipMemcpy(c->seed_dev, &s, 8, hipMemcpyHostToDevice));
      c->seed_on_dev = s; c->seed_valid = true;
    }
  }
  note_sampled(c, row, 1, *cfg);
  launch_sample(c, row, 1, *cfg, /*advance_pos=*/false, /*log_step=*/false);
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipStreamSynchronize(c->stream));
  int t = 0;
  HIP_OK(c, hipMemcpy(&t, c->rows[(size_t)row].tok, 4, hipMemcpyDeviceToHost));
  if (out_id) *out_id = t;
  if (row == 0) c->last_sampled0 = t;
  c->row_tok[(size_t)row] = 1;
  refresh_longest(c);
  return TGX_OK;
}
int64_t tgx_context_size(const tgx_ctx* c) { return c ? c->d.max_ctx : -1; }
int32_t tgx_num_layers(const tgx_ctx* c) { return c ? c->d.layers : -1; }

int tgx_synchronize(tgx_ctx* c) {
  if (!c) return TGX_ERR_INVALID;
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  return TGX_OK;
}

int tgx_read_kv(tgx_ctx* c, int row, int layer, float* k_out, float* v_out) {
  if (!c || !c->finalized || row < 0 || row >= c->d.max_batch || layer < 0 || layer >= c->d.layers) return TGX_ERR_INVALID;
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  const tgx_model_desc& d = c->d;
  const size_t hd = (size_t)d.head_dim, per_head = (size_t)d.max_ctx * hd, T = (size_t)tgx_past_length_row(c, row);
  std::vector<unsigned char> tmp(per_head * c->esz);
  for (int which = 0; which < 2; which++) {
    float* out = which ? v_out : k_out;
    if (!out) continue;
    const ebyte* base = (which ? c->rows[(size_t)row].vcache : c->rows[(size_t)row].kcache) + (size_t)layer * d.kv_heads * per_head * c->esz;
    if (c->kv_paged) base = (which ? c->slab_v : c->slab_k) + (size_t)layer * c->kv_nblocks * d.kv_heads * tgx::KV_BLOCK * hd * c->esz;
    for (int h = 0; h < d.kv_heads; h++) {
      if (c->kv_paged) {      // the row's tokens block by block through its table
        for (size_t t0 = 0; t0 < T; t0 += tgx::KV_BLOCK) {
          const size_t n = std::min<size_t>(tgx::KV_BLOCK, T - t0);
          const size_t blk = (size_t)c->kv_tbl_host[(size_t)row * c->kv_tbl_stride + t0 / tgx::KV_BLOCK];
          HIP_OK(c, hipMemcpy(tmp.data() + t0 * hd * c->esz, base + ((blk * d.kv_heads + h) * tgx::KV_BLOCK) * hd * c->esz, n * hd * c->esz, hipMemcpyDeviceToHost));
        }
      } else
      HIP_OK(c, hipMemcpy(tmp.data(), base + (size_t)h * per_head * c->esz, T * hd * c->esz, hipMemcpyDeviceToHost));
      for (size_t t = 0; t < T; t++)
        for (size_t k = 0; k < hd; k++) {   // BSHD view
          const size_t i = t * hd + k;
          float v;
          if (c->dt == tgx::DT_F32) memcpy(&v, tmp.data() + 4 * i, 4);
          else { uint16_t u; memcpy(&u, tmp.data() + 2 * i, 2); v = c->dt == tgx::DT_BF16 ? host_bf16_to_f32(u) : host_half_to_f32(u); }
          out[(t * d.kv_heads + h) * hd + k] = v;
        }
    }
  }
  return TGX_OK;
}

int tgx_write_kv(tgx_ctx* c, int row, int layer, const float* k_in, const float* v_in, int64_t n_rows) {
  if (!c || !c->finalized || row < 0 || row >= c->d.max_batch || layer < 0 || layer >= c->d.layers || n_rows < 0 || n_rows > tgx_past_length_row(c, row)) return c ? set_err(c, TGX_ERR_INVALID, "write_kv: row / layer / n_rows out of range") : TGX_ERR_INVALID;
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  const tgx_model_desc& d = c->d;
  const size_t hd = (size_t)d.head_dim, per_head = (size_t)d.max_ctx * hd, T = (size_t)n_rows;
  std::vector<unsigned char> tmp(T * hd * c->esz);
  for (int which = 0; which < 2; which++) {
    const float* in = which ? v_in : k_in;
    if (!in || !T) continue;
    ebyte* base = (which ? c->rows[(size_t)row].vcache : c->rows[(size_t)row].kcache) + (size_t)layer * d.kv_heads * per_head * c->esz;
    if (c->kv_paged) base = (which ? c->slab_v : c->slab_k) + (size_t)layer * c->kv_nblocks * d.kv_heads * tgx::KV_BLOCK * hd * c->esz;
    for (int h = 0; h < d.kv_heads; h++) {
      for (size_t t = 0; t < T; t++)
        for (size_t k = 0; k < hd; k++) {   // BSHD view in, head-major cache out; one round-to-nearest-even into the storage dtype
          const float v = in[(t * d.kv_heads + h) * hd + k];
          const size_t i = t * hd + k;
          if (c->dt == tgx::DT_F32) memcpy(tmp.data() + 4 * i, &v, 4);
          else { const uint16_t u = c->dt == tgx::DT_BF16 ? host_f32_to_bf16(v) : host_f32_to_half(v); memcpy(tmp.data() + 2 * i, &u, 2); }
        }
      if (c->kv_paged) {
        for (size_t t0 = 0; t0 < T; t0 += tgx::KV_BLOCK) {
          const size_t n = std::min<size_t>(tgx::KV_BLOCK, T - t0);
          const size_t blk = (size_t)c->kv_tbl_host[(size_t)row * c->kv_tbl_stride + t0 / tgx::KV_BLOCK];
          HIP_OK(c, hipMemcpy(base + ((blk * d.kv_heads + h) * tgx::KV_BLOCK) * hd * c->esz, tmp.data() + t0 * hd * c->esz, n * hd * c->esz, hipMemcpyHostToDevice));
        }
      } else
      HIP_OK(c, hipMemcpy(base + (size_t)h * per_head * c->esz, tmp.data(), T * hd * c->esz, hipMemcpyHostToDevice));
    }
  }
  return TGX_OK;
}

int tgx_profile_decode(tgx_ctx* c, int n_reps, int64_t* launches, double* total_ms) {
  if (!c || !launches || !total_ms || n_reps < 0) return TGX_ERR_INVALID;
  if (!c->have_token) return set_err(c, TGX_ERR_STATE, "profile needs a current token: call tgx_sample after tgx_forward");
  if (c->past + 1 > c->d.max_ctx) return set_err(c, TGX_ERR_CONTEXT, "context size exceeded");
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < TGX_KERNEL_COUNT; i++) { launches[i] = 0; total_ms[i] = 0.0; }
  update_attn_modes(c, 1);
  // Each class is launched back-to-back over all layers (every launch streams a different layer's weights, so
  // nothing is served from the Infinity Cache) between two events on the launch stream.  The residual
  // epilogues write to a scratch vector: the model state (x, KV cache up to pastLength, token) is untouched.
  for (int rep = 0; rep < n_reps; rep++) {
    for (int cls = 0; cls < TGX_KERNEL_COUNT; cls++) {
      HIP_OK(c, hipEventRecord(c->prof.ev[0], c->stream));
      int n = 0;
      if (cls == TGX_KERNEL_LMHEAD) { launch_lm_head(c, 0, 1); n = 1; }
      else for (int l = 0; l < c->d.layers; l++) n += launch_layer_kernel(c, &c->rows[0], 1, c->prof_same_layer ? 0 : l, cls, c->scratch_x, (long long)c->kv_row_elems);
      HIP_OK(c, hipEventRecord(c->prof.ev[1], c->stream));
      HIP_OK(c, hipEventSynchronize(c->prof.ev[1]));
      float ms = 0.f;
      HIP_OK(c, hipEventElapsedTime(&ms, c->prof.ev[0], c->prof.ev[1]));
      launches[cls] += n;
      total_ms[cls] += ms;
    }
  }
  HIP_OK(c, hipGetLastError());
  c->have_logits = false;   // the lm_head replay overwrote the logits buffer
  return TGX_OK;
}

int tgx_read_probs(tgx_ctx* c, float* out) {
  if (!c || !out) return TGX_ERR_INVALID;
  if (!c->have_probs || !c->have_logits) return set_err(c, TGX_ERR_STATE, "no probabilities: the last sample was greedy or none was taken");
  HIP_OK(c, hipSetDevice(c->device));
  // a sampled step leaves its logits, thresholds and normalisers on the device, not the vector: evaluate it now (rows whose last step was greedy read as zeros)
  const size_t V = (size_t)c->d.vocab;
  for (int b = 0; b < c->batch; b++) {
    if (c->row_probs_ok[(size_t)b]) launch_probs(c, b, c->row_probs_cfg[(size_t)b]);
    else HIP_OK(c, hipMemsetAsync(c->rows[(size_t)b].probs, 0, V * 4, c->stream));
  }
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipStreamSynchronize(c->stream));
  for (int b = 0; b < c->batch; b++) HIP_OK(c, hipMemcpy(out + b * V, c->rows[(size_t)b].probs, V * 4, hipMemcpyDeviceToHost));
  return TGX_OK;
}

int tgx_set_logits(tgx_ctx* c, const float* logits, int batch) {
  if (!c || !logits) return TGX_ERR_INVALID;
  if (!c->finalized) return set_err(c, TGX_ERR_STATE, "set_logits before finalize");
  if (batch < 1 || batch > c->d.max_batch) return set_err(c, TGX_ERR_INVALID, "batch out of range");
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  const size_t V = (size_t)c->d.vocab;
  for (int b = 0; b < batch; b++) HIP_OK(c, hipMemcpy(c->rows[(size_t)b].logits, logits + b * V, V * 4, hipMemcpyHostToDevice));
  // the greedy path reads per-workgroup argmax partials: rebuild them from the injected logits
  for (int b = 0; b < batch; b++) {
    RowState& r = c->rows[(size_t)b];
    launch_argmax_partials(c, r.logits, (int)V, r.part_val, r.part_idx);
  }
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipStreamSynchronize(c->stream));
  c->batch = batch;
  c->have_probs = false;
  std::fill(c->row_probs_ok.begin(), c->row_probs_ok.end(), 0);
  c->have_logits = true;
  return TGX_OK;
}

int tgx_get_option(const tgx_ctx* c, const char* key, int* out_value) {
  if (!c || !key || !out_value) return TGX_ERR_INVALID;
  if (!strcmp(key, "attn.direct_limit")) { *out_value = (int)std::min<long long>(direct_limit(c, c->batch, true), INT_MAX); return TGX_OK; }
  if (!strcmp(key, "attn.nw4_limit")) { *out_value = (int)nw4_limit(c, c->batch, true); return TGX_OK; }
  if (!strcmp(key, "graph.steps")) { *out_value = c->graph_steps; return TGX_OK; }
  if (!strcmp(key, "act.round16")) { *out_value = c->act16; return TGX_OK; }
  if (!strcmp(key, "kv.budget_tokens")) { *out_value = c->kv_budget_tokens; return TGX_OK; }
  if (!strcmp(key, "kv.free_tokens")) { *out_value = c->kv_paged ? (int)c->kv_free.size() * tgx::KV_BLOCK : -1; return TGX_OK; }      // paged KV: tokens' worth of unassigned blocks
  return TGX_ERR_INVALID;
}

int tgx_set_option(tgx_ctx* c, const char* key, int value) {
  if (!c || !key) return TGX_ERR_INVALID;
  static const char* cls_names[TGX_KERNEL_COUNT] = {"qkv", "attn", "oproj", "gateup", "down", "lmhead"};
  drop_step_graphs(c);
  if (!strcmp(key, "graph")) { c->use_graph = value != 0; return TGX_OK; }
  if (!strcmp(key, "graph.steps")) { if (value < 1 || value > 64) return set_err(c, TGX_ERR_INVALID, "graph.steps out of range"); c->graph_steps = value; return TGX_OK; }
  if (!strcmp(key, "debug.skip")) { c->debug_skip = value; return TGX_OK; }
  if (!strcmp(key, "debug.attn")) { c->debug_attn = value; return TGX_OK; }
  if (!strcmp(key, "attn.gmax")) {   // query heads per attention workgroup: the kernel is instantiated for 1..4 (0 = default)
    if (value < 0 || value > 4) return set_err(c, TGX_ERR_INVALID, "attn.gmax must be 0 (default) or 1..4");
    c->attn_gmax = value; return TGX_OK;
  }
  if (!strcmp(key, "attn.direct_max")) { c->attn_direct_max = c->attn_fused_max = value; return TGX_OK; }      // (both forms of the limit: an explicit value forces the form)
  if (!strcmp(key, "attn.direct_nw4")) { drop_step_graphs(c); c->attn_direct_nw4 = c->attn_fused_nw4 = value; return TGX_OK; }
  if (!strcmp(key, "attn.fused_max")) { c->attn_fused_max = value; return TGX_OK; }
  if (!strcmp(key, "attn.fused_nw4")) { drop_step_graphs(c); c->attn_fused_nw4 = value; return TGX_OK; }
  if (!strcmp(key, "skinny.terms_above")) { drop_step_graphs(c); c->skinny_terms_above = value; return TGX_OK; }
  if (!strcmp(key, "prefill.qkv_nosplit")) { c->qkv_nosplit = value; return TGX_OK; }
  if (!strcmp(key, "prefill.terms_rows")) { c->prefill_terms_rows = value; return TGX_OK; }
  if (!strcmp(key, "prefill.splitk_8k")) { c->splitk_8k = value; return TGX_OK; }
  if (!strcmp(key, "act.round16")) { drop_step_graphs(c); c->act16 = value != 0; return TGX_OK; }
  if (!strcmp(key, "act.one_term_kernels")) { c->act16_kernels = value != 0; return TGX_OK; }      // 0: the two-term kernels on the all-zero second term (bit-identical; tests)      // (the all-zero term buffer is allocated with the next workspace check)
  if (!strcmp(key, "oproj.fused")) { drop_step_graphs(c); c->oproj_fused = value != 0; return TGX_OK; }
  if (!strcmp(key, "attn.raw_fuse")) { drop_step_graphs(c); c->attn_raw_fuse = value; return TGX_OK; }
  if (!strcmp(key, "attn.batch_nw8")) { drop_step_graphs(c); c->attn_batch_nw8 = value; return TGX_OK; }
  if (!strcmp(key, "attn.batch_la")) { if (value < -1 || value > 1) return set_err(c, TGX_ERR_INVALID, "attn.batch_la is -1, 0 or 1"); drop_step_graphs(c); c->attn_batch_la = value; return TGX_OK; }
  if (!strcmp(key, "attn.batch_mfma")) { if (value < 0) return set_err(c, TGX_ERR_INVALID, "attn.batch_mfma is a row count (0 = off)"); drop_step_graphs(c); c->attn_batch_mfma = value; return TGX_OK; }
  if (!strcmp(key, "attn.direct_g")) { if (value != 0 && value != 1 && value != -1 && value != -2 && value != -4) return set_err(c, TGX_ERR_INVALID, "attn.direct_g is 0, 1 or -1 / -2 / -4"); drop_step_graphs(c); c->attn_direct_g = value; return TGX_OK; }
  if (!strcmp(key, "attn.mfma_min")) { c->attn_mfma_min = value; return TGX_OK; }
  if (!strcmp(key, "prefill.defer_min_rows")) { c->defer_min_rows = value; return TGX_OK; }
  if (!strcmp(key, "prefill.defer_reduce")) { c->defer_reduce = value != 0; return TGX_OK; }
  if (!strcmp(key, "skinny.dma")) { drop_step_graphs(c); c->skinny_dma = value != 0; return TGX_OK; }
  if (!strcmp(key, "skinny.dma_oproj")) { drop_step_graphs(c); c->skinny_dma_oproj = value; return TGX_OK; }
  if (!strcmp(key, "skinny.dma_qkv")) { drop_step_graphs(c); c->skinny_dma_qkv = value; return TGX_OK; }
  if (!strcmp(key, "skinny.dma_nbw")) { if (value < 0 || value > 2) return set_err(c, TGX_ERR_INVALID, "skinny.dma_nbw is 0, 1 or 2"); drop_step_graphs(c); c->skinny_dma_nbw = value; return TGX_OK; }
  if (!strcmp(key, "skinny.dma_rows")) { if (value < 1) return set_err(c, TGX_ERR_INVALID, "skinny.dma_rows is a row count"); drop_step_graphs(c); c->skinny_dma_rows = value; return TGX_OK; }
  if (!strcmp(key, "skinny.terms")) { drop_step_graphs(c); c->skinny_terms = value; return TGX_OK; }     // 2: the QKV and lm_head products of 17-32-row batches as well (experiment)
  if (!strcmp(key, "skinny.ksplit")) { if (value < 0 || value > 2) return set_err(c, TGX_ERR_INVALID, "skinny.ksplit must be 0, 1 (<= 16 rows) or 2 (<= 32 rows)"); c->skinny_ksplit = value; return TGX_OK; }
  if (!strcmp(key, "prefill.gemm_tm")) { c->gemm_tm = value; return TGX_OK; }
  if (!strcmp(key, "prefill.gemm_dma")) { c->gemm_dma = value; return TGX_OK; }
  if (!strcmp(key, "debug.gemv")) { c->debug_gemv = value; return TGX_OK; }
  if (!strcmp(key, "prefill.mfma")) { c->prefill_mfma = value != 0; return TGX_OK; }
  if (!strcmp(key, "prefill.min_rows")) { c->prefill_min_rows = value; return TGX_OK; }
  if (!strcmp(key, "prefill.f32_min_rows")) { c->prefill_f32_min_rows = value; return TGX_OK; }
  if (!strcmp(key, "prefill.splitk")) { c->gemm_splitk = value; return TGX_OK; }
  if (!strcmp(key, "decode.step_rows")) { if (value != 32 && value != 64 && value != 128) return set_err(c, TGX_ERR_INVALID, "decode.step_rows is 32, 64 or 128"); drop_step_graphs(c); c->decode_step_rows = value; return TGX_OK; }
  if (!strcmp(key, "prefill.skinny_hidden_max_wide")) { c->prefill_skinny_hidden_max_wide = value; return TGX_OK; }
  if (!strcmp(key, "prefill.skinny_hidden_max")) { c->prefill_skinny_hidden_max = value; return TGX_OK; }
  if (!strcmp(key, "prefill.skinny_rows")) { if (value < 0 || value > 128) return set_err(c, TGX_ERR_INVALID, "prefill.skinny_rows is 0..128"); c->prefill_skinny_rows = value; return TGX_OK; }
  if (!strcmp(key, "prefill.wide_8k_eff")) { c->wide_8k_eff = value; return TGX_OK; }
  if (!strcmp(key, "prefill.wide_8k_max")) { c->wide_8k_max = value; return TGX_OK; }
  if (!strcmp(key, "prefill.wide_n_min")) { c->wide_n_min = value; return TGX_OK; }
  if (!strcmp(key, "prefill.attn_dma")) { c->attn_dma = value; return TGX_OK; }
  if (!strcmp(key, "prefill.attn_ksplit")) { c->attn_ksplit = value; return TGX_OK; }
  if (!strcmp(key, "attn.qk_fuse")) { c->qk_fuse = value; return TGX_OK; }
  if (!strcmp(key, "skinny.wgs")) { if (value < 1) return set_err(c, TGX_ERR_INVALID, "skinny.wgs must be >= 1"); c->skinny_wgs = value; return TGX_OK; }
  if (!strcmp(key, "prefill.skinny")) { c->prefill_skinny = value; return TGX_OK; }
  if (!strcmp(key, "skinny.gu_split")) { c->skinny_gu_split = value; return TGX_OK; }
  if (!strcmp(key, "skinny.cfg_mid")) { if (value < 0 || value > 2) return set_err(c, TGX_ERR_INVALID, "skinny.cfg_mid is 0..2"); c->skinny_cfg_mid = value; return TGX_OK; }
  if (!strcmp(key, "skinny.cfg")) { if (value < -1 || value > 2) return set_err(c, TGX_ERR_INVALID, "skinny.cfg is -1..2"); c->skinny_cfg_force = value; return TGX_OK; }
  if (!strcmp(key, "decode.mfma_min_batch")) { if (value < 1) return set_err(c, TGX_ERR_INVALID, "decode.mfma_min_batch must be >= 1"); c->decode_mfma_min = value; return TGX_OK; }
  if (!strcmp(key, "debug.profile_same_layer")) { c->prof_same_layer = value; return TGX_OK; }
  if (!strcmp(key, "oproj.sliced")) { drop_step_graphs(c); c->oproj_sliced = value != 0; return TGX_OK; }
  if (!strcmp(key, "kv.budget_tokens")) {
    if (c->finalized) return set_err(c, TGX_ERR_STATE, "kv.budget_tokens is set before tgx_finalize (it sizes the caches)");
    if (value < 0) return set_err(c, TGX_ERR_INVALID, "kv.budget_tokens >= 0 (0 = one max_ctx slab per row)");
    c->kv_budget_tokens = value; return TGX_OK;
  }
  if (!strcmp(key, "attn.nsplit")) {
    if (c->finalized) return set_err(c, TGX_ERR_STATE, "attn.nsplit must be set before tgx_finalize");
    if (value < 1 || value > 32) return set_err(c, TGX_ERR_INVALID, "attn.nsplit out of range");
    c->attn_nsplit_opt = value;
    return TGX_OK;
  }
  for (int i = 0; i < TGX_KERNEL_COUNT; i++) {
    const size_t n = strlen(cls_names[i]);
    if (strncmp(key, cls_names[i], n) || key[n] != '.') continue;
    if (!strcmp(key + n + 1, "ks")) {
      if (value != 1 && value != 2 && value != 4) return set_err(c, TGX_ERR_INVALID, "ks must be 1, 2 or 4");
      c->tune[i].ks = value;
      return TGX_OK;
    }
    if (!strcmp(key + n + 1, "bpc")) {
      if (value < 1 || value > 16) return set_err(c, TGX_ERR_INVALID, "bpc out of range");
      if (i == TGX_KERNEL_LMHEAD && c->finalized) return set_err(c, TGX_ERR_STATE, "lmhead.bpc must be set before tgx_finalize");
      c->tune[i].bpc = value;
      return TGX_OK;
    }
  }
  return set_err(c, TGX_ERR_INVALID, "unknown option %s", key);
}

int64_t tgx_bytes_per_token(const tgx_ctx* c, int64_t T) {
  if (!c) return -1;
  const tgx_model_desc& d = c->d;
  const int64_t H = d.hidden, I = d.inter, V = d.vocab, L = d.layers, q = (int64_t)d.heads * d.head_dim, kv = (int64_t)d.kv_heads * d.head_dim;
  const int64_t b = (int64_t)c->esz;   // bytes per stored parameter / cache element
  if (c->gpt2) {   // c_attn, c_proj, c_fc, mlp.c_proj with their biases, two LayerNorms (weight + bias); ln_f, one wpe row, the wte head
    const int64_t per_layer = 3 * H * H + 3 * H + H * H + H + I * H + I + H * I + H + 4 * H;
    return b * (L * per_layer + 2 * H + H + V * H) + b * 2 * L * kv * T;
  }
  const int64_t per_layer = (q + 2 * kv) * H + (d.qkv_bias ? (q + 2 * kv) : 0) + H * q + 2 * I * H + H * I + 2 * H;
  return b * (L * per_layer + H + V * H) + b * 2 * L * kv * T;
}

}  // extern "C"
