// decode.hip — the batch-1..4 decode step of the MI355X shim: one launch per kernel class and layer (kernels/gemv.h), the K-sliced o_proj behind
// split-form attention (kernels/oproj_sliced.h), lm_head + per-workgroup argmax, the greedy finalize.
// == for (auto& layer : layers_) x = layer->forward(x); norm; lm_head  (GPTModel.h:51-58, DecoderLayer.h:38-43, Attention.h:71-91, GatedMLP.h:37-41)
#include "ctx.h"
#include "kernels/attn_decode.h"
#include "kernels/gemv.h"
#include "kernels/oproj_sliced.h"

int gemv_grid(const tgx_ctx* c, int units, int ks, int bpc) {
  const int upb = 4 / ks;
  const int want = (units + upb - 1) / upb;
  const int cap = c->num_cus * bpc;
  return want < cap ? want : cap;
}

// 16-byte slices per lane per row for a K range split over ks waves (the kernel's NX template parameter)
static int gemv_nx(int K, int ks) { return ((K / 8) + ks * 64 - 1) / (ks * 64); }

// smallest K split that keeps a wave's slice within 8 x 512 elements; norm-fused launches must use 1
static int gemv_auto_ks(int K, int want) {
  int ks = want;
  while (ks < 4 && gemv_nx(K, ks) > 8) ks *= 2;
  return ks;
}

// the argument block of batch row `r` alone (every slab pointer advanced by r row strides)
static tgx::GemvArgs gemv_row(const tgx_ctx* c, tgx::GemvArgs a, int r) {
  a.x += (size_t)r * a.x_stride;
  if (a.out) a.out += (size_t)r * a.out_stride;
  if (a.q_out) a.q_out += (size_t)r * a.q_stride;
  if (a.k_raw) a.k_raw += (size_t)r * a.kraw_stride;
  if (a.k_cache) a.k_cache = (ebyte*)a.k_cache + (size_t)r * a.kv_stride * c->esz;
  if (a.v_cache) a.v_cache = (ebyte*)a.v_cache + (size_t)r * a.kv_stride * c->esz;
  if (a.pos) a.pos += r;
  if (a.blk_tbl) a.blk_tbl += (size_t)r * a.tbl_stride;
  if (a.logits) a.logits += (size_t)r * a.logits_stride;
  if (a.part_val) a.part_val += (size_t)r * a.part_stride;
  if (a.part_idx) a.part_idx += (size_t)r * a.part_stride;
  return a;
}

template <int DT, int PRO, int EPI, int NX>
static void launch_gemv_nx(tgx_ctx* c, const tgx::GemvArgs& a, int grid, int R) {
  const dim3 g(grid), b(256);
  // rows share the weight pass; R x NX activation slices of 8 floats stay in registers (4 x 8 x 8 = 256 of the 512 a wave
  // of a 256-thread workgroup may use)
  if (R == 4) { hipLaunchKernelGGL((tgx::gemv_kernel<DT, PRO, EPI, NX, 4>), g, b, 0, c->stream, a); return; }
  if (R == 2) { hipLaunchKernelGGL((tgx::gemv_kernel<DT, PRO, EPI, NX, 2>), g, b, 0, c->stream, a); return; }
  // the fixed-point residual stream behind the K-sliced o_proj (batch 1, 16-bit storage, RMSNorm families): gate_up reads it, down adds to it
  if constexpr (DT != tgx::DT_F32 && ((PRO == tgx::PRO_RMSNORM && EPI == tgx::EPI_SILU_MUL) || (PRO == tgx::PRO_PLAIN && EPI == tgx::EPI_RESIDUAL))) {
    if (R == 1 && (a.x_acc || a.res_acc)) { hipLaunchKernelGGL((tgx::gemv_kernel<DT, PRO, EPI, NX, 1, true>), g, b, a.x_acc ? (size_t)a.K * 4 : 0, c->stream, a); return; }
  }
  for (int r = 0; r < R; r++) hipLaunchKernelGGL((tgx::gemv_kernel<DT, PRO, EPI, NX, 1>), g, b, 0, c->stream, gemv_row(c, a, r));
}

template <int PRO, int EPI>
static void launch_gemv(tgx_ctx* c, tgx::GemvArgs a, int cls, int R) {
  const Tune& tn = c->tune[cls];
  a.dbg = ((c->debug_gemv >> (8 + cls)) & 1) ? (c->debug_gemv & 15) : 0;
  if (!a.ldw) a.ldw = a.K;
  a.ks = gemv_auto_ks(a.K, tn.ks);   // norm-fused launches K-split too (their waves exchange the sums of squares through LDS)
  while (R > 1 && a.ks < 4 && gemv_nx(a.K, a.ks) > 4) a.ks *= 2;   // batch rows: at most 4 slices per row and lane (measured: B = 2 and 4 on the 1B / 3B / 7B shapes)
  // two rows on the gate_up launch: 2 slices per row and lane leave room for the double-buffered weight registers (R x NX <= 4):
  // Llama-3.2-1B B = 2: 2394 -> 2506 tok/s; the same split loses 2-6 % at B = 4 and on the other launches (tools/batch_bench.py --opts)
  if (R == 2 && cls == TGX_KERNEL_GATEUP && a.ks == 1 && gemv_nx(a.K, 1) == 4) a.ks = 2;
  const int grid = (EPI == tgx::EPI_LOGITS) ? c->lm_grid : gemv_grid(c, a.units, a.ks, tn.bpc);
  const int nx = gemv_nx(a.K, a.ks);
  if constexpr (PRO == tgx::PRO_LAYERNORM || EPI == tgx::EPI_GELU) {   // GPT-2 (hidden <= 2048, checked in tgx_create): at most 4 slices per lane
    TGX_DT_SWITCH(c->dt, switch (nx) {
      case 1: launch_gemv_nx<DT, PRO, EPI, 1>(c, a, grid, R); break;
      case 2: launch_gemv_nx<DT, PRO, EPI, 2>(c, a, grid, R); break;
      case 3: launch_gemv_nx<DT, PRO, EPI, 3>(c, a, grid, R); break;
      default: launch_gemv_nx<DT, PRO, EPI, 4>(c, a, grid, R); break;
    })
  } else {
    TGX_DT_SWITCH(c->dt, switch (nx) {
      case 1: launch_gemv_nx<DT, PRO, EPI, 1>(c, a, grid, R); break;
      case 2: launch_gemv_nx<DT, PRO, EPI, 2>(c, a, grid, R); break;
      case 3: launch_gemv_nx<DT, PRO, EPI, 3>(c, a, grid, R); break;
      case 4: launch_gemv_nx<DT, PRO, EPI, 4>(c, a, grid, R); break;
      case 5: launch_gemv_nx<DT, PRO, EPI, 5>(c, a, grid, R); break;
      case 6: launch_gemv_nx<DT, PRO, EPI, 6>(c, a, grid, R); break;
      case 7: launch_gemv_nx<DT, PRO, EPI, 7>(c, a, grid, R); break;
      default: launch_gemv_nx<DT, PRO, EPI, 8>(c, a, grid, R); break;
    })
  }
}

static void fill_strides(const tgx_ctx* c, tgx::GemvArgs& a) {
  const tgx_model_desc& d = c->d;
  a.x_stride = 0; a.out_stride = 0;       // set per call site (x and out come from different slabs)
  a.q_stride = (long long)d.heads * d.head_dim; a.kraw_stride = (long long)d.kv_heads * d.head_dim;
  a.kv_stride = (long long)c->kv_row_elems; a.logits_stride = d.vocab; a.part_stride = c->lm_grid;
  a.act16 = c->act16;
}

// Qwen3: independent batch rows (each its own cache) take the q/k norm inside the attention launch; the positions of one sequence that a
// prefill-by-steps pass handles together (kv_stride 0) need each other's finished keys, so they keep the separate norm launch
static bool qk_fused(const tgx_ctx* c, long long kv_stride) { return c->d.qk_norm && c->d.head_dim == 128 && kv_stride != 0 && c->qk_fuse; }

// The K-sliced o_proj with the attention merge in its prologue (kernels/oproj_sliced.h; no attn_combine launch, fixed-point residual stream between o_proj
// and down) serves batch-1 steps on the VALU split attention form: head_dim 64 (a slice of 2-4 heads reads 9-35 KB of records; at head_dim 128 the records
// outweigh the launch it removes: Llama-3.2-3B -0.2 us per layer, Mistral-7B +3.4, tools/probes/layer_lab.hip), 16-bit storage, RMSNorm families
// without an o_proj bias, one down launch (intermediate <= 16384), row blocks that tile hidden.  Measured on Llama-3.2-1B at context 2064:
// {attention, combine, o_proj} 12.9 -> {attention, sliced o_proj} 10.2 us per layer; Qwen2.5-0.5B 10.1 -> 8.4 (profiles/r04_oproj_sliced.txt).
bool oproj_sliced_ok(const tgx_ctx* c, int R, long long kv_stride) {
  const tgx_model_desc& d = c->d;
  if (!c->oproj_sliced || R != 1 || kv_stride == 0 || c->attn_direct || c->attn_mfma || c->gpt2 || c->dt == tgx::DT_F32 || !c->slab_acc) return false;
  if (d.head_dim != 64 || d.qk_norm || d.inter > 16384 || c->attn_nsplit > 32) return false;
  if (d.hidden > tgx::XACC_HIDDEN_MAX) return false;     // the fixed-point gate_up hands the converted residual to its waves through hidden * 4 bytes of LDS
  const int qd = d.heads * d.head_dim;
  if (qd % 256 == 0) return d.hidden % tgx::oproj_sliced_rows<32>() == 0;
  return qd % 128 == 0 && d.hidden % tgx::oproj_sliced_rows<16>() == 0;
}

// The direct attention form with the o_proj product in its epilogue (attn_decode_kernel template OPJ, kernels/attn_decode.h): batch-1 steps at short
// contexts run {qkv, attention + o_proj, gate_up, down}.  Each query head's workgroups (oj_rsplit of them, each repeating the head's attention over the few
// hundred keys) multiply the normalised head output by their rows of W_o[:, head columns] and add into the fixed-point residual accumulators that gate_up
// and down already read (see above).  Measured per layer (tools/probes/layer_lab.hip, profiles/r04_attn_oproj_fused.txt): Qwen2.5-0.5B at context 30 / 270 /
// 600 20.2 / 21.0 / 21.8 -> 18.8 / 19.9 / 21.4 us, Llama-3.2-1B 33.3 / 34.2 / 35.5 -> 30.9 / 32.9 / 34.3; the split form is ahead from ~700 / ~900 keys.
bool oproj_fused_capable(const tgx_ctx* c) {
  const tgx_model_desc& d = c->d;
  if (!c->oproj_fused || c->gpt2 || c->dt == tgx::DT_F32 || !c->slab_acc) return false;
  return d.head_dim == 64 && !d.qk_norm && d.inter <= 16384 && d.hidden % 8 == 0 && d.hidden <= tgx::XACC_HIDDEN_MAX;
}
bool oproj_fused_ok(const tgx_ctx* c, int R, long long kv_stride) {
  return R == 1 && c->batch == 1 && kv_stride != 0 && c->attn_direct && oproj_fused_capable(c);
}
// the residual stream between o_proj and down lives in the fixed-point accumulators (either o_proj form above)
static bool resid_fixed(const tgx_ctx* c, int R, long long kv_stride) { return oproj_sliced_ok(c, R, kv_stride) || oproj_fused_ok(c, R, kv_stride); }

// One kernel class of one decoder layer for R rows (batch rows of the slabs, or the chunk rows of a prefill-by-steps pass).  `resid` is the residual stream of row0 that the
// o_proj/down epilogues update (slab_x in the real pass; a scratch vector when tgx_profile_decode replays a class).
int launch_layer_kernel(tgx_ctx* c, RowState* rv, int R, int l, int cls, float* resid, long long kv_stride) {
  const tgx_model_desc& d = c->d;
  RowState& r = rv[0];   // R consecutive row views with the slabs' row strides; kv_stride = 0 when the rows are positions of ONE sequence
  const int H = d.hidden, I = d.inter, hd = d.head_dim, qd = d.heads * hd, kvd = d.kv_heads * hd;
  // bytes between a layer's caches (ebyte pointers): a row's slab [kv_heads][max_ctx][hd], or — paged KV — the layer's pool of KV_BLOCK-token blocks
  const size_t kv_layer = (c->kv_paged ? (size_t)c->kv_nblocks * d.kv_heads * tgx::KV_BLOCK : (size_t)d.kv_heads * d.max_ctx) * hd * c->esz;
  // paged: the rows share the pools (no row stride) and differ by their block tables; rows that are positions of ONE sequence (kv_stride 0) share one table
  const long long kvs = c->kv_paged ? 0 : kv_stride, tbs = (c->kv_paged && kv_stride != 0) ? c->kv_tbl_stride : 0;
  const LayerW& w = c->L[(size_t)l];
  switch (cls) {
    case TGX_KERNEL_QKV: {   // input_layernorm -> qkv_proj -> RoPE -> cache append   (DecoderLayer.h:40, Attention.h:94-106)
      tgx::GemvArgs a{};
      fill_strides(c, a);
      a.W = w.wqkv; a.bias = w.bqkv; a.x = r.x; a.x_stride = H; a.norm_w = w.in_norm; a.eps = d.norm_eps;
      a.N = qd + 2 * kvd; a.K = H; a.units = a.N / 2;
      a.q_out = r.q; a.k_cache = r.kcache + (size_t)l * kv_layer; a.v_cache = r.vcache + (size_t)l * kv_layer; a.kv_stride = kvs;
      a.blk_tbl = r.tbl; a.tbl_stride = tbs;
      a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.pos = r.pos;
      a.heads = d.heads; a.kv_heads = d.kv_heads; a.hd = hd; a.max_ctx = d.max_ctx;
      a.raw_qk = d.qk_norm ? 1 : 0; a.k_raw = r.k_raw;
      if (c->gpt2) {   // ln_1 -> c_attn (+bias) -> split into heads -> cache append; no rotation: the tables hold cos = 1, sin = 0 (ModelGPT2.h:60-75)
        a.norm_b = w.in_norm_b;
        launch_gemv<tgx::PRO_LAYERNORM, tgx::EPI_QKV_ROPE>(c, a, TGX_KERNEL_QKV, R);
        break;
      }
      launch_gemv<tgx::PRO_RMSNORM, tgx::EPI_QKV_ROPE>(c, a, TGX_KERNEL_QKV, R);
      if (d.qk_norm && !qk_fused(c, kv_stride)) {   // q_norm / k_norm -> RoPE -> cache append (Attention.h:156-163); batch rows on blockIdx.y
        tgx::QkNormArgs n{};
        n.q = r.q; n.k_raw = r.k_raw; n.k_cache = r.kcache + (size_t)l * kv_layer; n.q_norm_w = w.q_norm; n.k_norm_w = w.k_norm;
        n.rope_cos = c->rope_cos; n.rope_sin = c->rope_sin; n.pos = r.pos;
        n.heads = d.heads; n.kv_heads = d.kv_heads; n.hd = hd; n.max_ctx = d.max_ctx; n.eps = d.norm_eps;
        n.q_stride = qd; n.kraw_stride = kvd; n.kv_stride = kvs; n.blk_tbl = r.tbl; n.tbl_stride = tbs;
        TGX_DT_SWITCH(c->dt, hipLaunchKernelGGL(tgx::qk_norm_rope_kernel<DT>, dim3(d.heads + d.kv_heads, R), dim3(64), 0, c->stream, n))
      }
      break;
    }
    case TGX_KERNEL_ATTN: {  // flashAttention(q, Kall, Vall) over keys [0, pos[row]]; blockIdx.y = batch row (own cache, own length)
      tgx::AttnArgs a{};
      a.q = r.q; a.k_cache = r.kcache + (size_t)l * kv_layer; a.v_cache = r.vcache + (size_t)l * kv_layer;
      a.pos = r.pos; a.part = r.attn_part; a.out = r.attn;
      a.heads = d.heads; a.kv_heads = d.kv_heads; a.max_ctx = d.max_ctx; a.nsplit = c->attn_nsplit;
      a.scale = 1.0f / sqrtf((float)hd);
      a.q_stride = qd; a.kv_stride = kvs; a.part_stride = (long long)c->attn_part_row; a.dbg = c->debug_attn;
      a.blk_tbl = r.tbl; a.tbl_stride = tbs;
      a.act16 = c->act16 ? (c->dt == tgx::DT_BF16 ? 1 : (c->dt == tgx::DT_F16 ? 2 : 0)) : 0;
      if (qk_fused(c, kv_stride)) {
        a.k_raw = r.k_raw; a.kraw_stride = kvd; a.q_norm_w = w.q_norm; a.k_norm_w = w.k_norm;
        a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.eps = d.norm_eps;
      }
      if (oproj_fused_ok(c, R, kv_stride)) {      // + o_proj and the residual add in the same launch (Attention.h:111-112, DecoderLayer.h:40)
        a.oj_w = w.wo; a.oj_x = r.x; a.oj_acc = c->slab_acc + (size_t)(&r - c->rows.data()) * H; a.oj_H = H; a.oj_ldw = qd;
      }
      launch_attn(c, a, R, /*combine=*/!oproj_sliced_ok(c, R, kv_stride));
      break;
    }
    case TGX_KERNEL_OPROJ: { // o_proj + residual                                     (Attention.h:90, DecoderLayer.h:40)
      if (oproj_fused_ok(c, R, kv_stride)) return 0;       // done by the attention launch
      tgx::GemvArgs a{};
      fill_strides(c, a);
      a.W = w.wo; a.bias = w.bo; a.x = r.attn; a.x_stride = qd; a.N = H; a.K = qd; a.units = (H + 1) / 2; a.out = resid; a.out_stride = H; a.hd = 2;
      if (oproj_sliced_ok(c, R, kv_stride)) {   // split-form attention, batch 1: K-sliced product, the split records merged in its prologue, partial sums into
        // the row's fixed-point accumulators (kernels/oproj_sliced.h); the residual x is added by K slice 0; gate_up reads the accumulators, down empties them
        tgx::OprojSlicedArgs o{};
        o.W = w.wo; o.ldw = qd; o.part = r.attn_part; o.nsplit = c->attn_nsplit; o.x = r.x; o.acc = c->slab_acc + (size_t)(&r - c->rows.data()) * H; o.H = H; o.act16 = c->act16;
        const dim3 blk(256);
        TGX_DT16_SWITCH(c->dt,
          if (qd % 256 == 0) hipLaunchKernelGGL((tgx::oproj_sliced_kernel<DT, 64, 32>), dim3(H / tgx::oproj_sliced_rows<32>(), qd / 256), blk, 0, c->stream, o);
          else hipLaunchKernelGGL((tgx::oproj_sliced_kernel<DT, 64, 16>), dim3(H / tgx::oproj_sliced_rows<16>(), qd / 128), blk, 0, c->stream, o);)
        break;
      }
      launch_gemv<tgx::PRO_PLAIN, tgx::EPI_RESIDUAL>(c, a, TGX_KERNEL_OPROJ, R);
      break;
    }
    case TGX_KERNEL_GATEUP: { // post_attention_layernorm -> gate_up_proj -> siluMul  (DecoderLayer.h:41, GatedMLP.h:37-39)
      tgx::GemvArgs a{};
      fill_strides(c, a);
      a.W = w.wgu; a.x = r.x; a.x_stride = H; a.norm_w = w.post_norm; a.eps = d.norm_eps;
      if (c->gpt2) {   // ln_2 -> c_fc (+bias) -> gelu_new   (ModelGPT2.h:96-107,131-134)
        a.norm_b = w.post_norm_b; a.bias = w.bfc;
        a.N = I; a.K = H; a.units = (I + 1) / 2; a.out = r.h; a.out_stride = I; a.hd = 2;
        launch_gemv<tgx::PRO_LAYERNORM, tgx::EPI_GELU>(c, a, TGX_KERNEL_GATEUP, R);
        break;
      }
      a.N = 2 * I; a.K = H; a.units = I; a.out = r.h; a.out_stride = I; a.hd = 2;
      if (resid_fixed(c, R, kv_stride)) a.x_acc = c->slab_acc + (size_t)(&r - c->rows.data()) * H;     // x' = fp32(acc)
      launch_gemv<tgx::PRO_RMSNORM, tgx::EPI_SILU_MUL>(c, a, TGX_KERNEL_GATEUP, R);
      break;
    }
    case TGX_KERNEL_DOWN: {  // down_proj + residual                                  (GatedMLP.h:40, DecoderLayer.h:41)
      tgx::GemvArgs a{};
      fill_strides(c, a);
      a.W = w.wdown; a.bias = w.bdown; a.x = r.h; a.x_stride = I; a.N = H; a.K = I; a.units = (H + 1) / 2; a.out = resid; a.out_stride = H; a.hd = 2;
      if (I > 16384) {   // 32B/70B-class intermediate sizes: one launch keeps at most 16384 elements of x in registers — the K range is
        // covered by 2-4 launches that accumulate into the residual stream in order (x += W[:, k0:k1] . h[k0:k1])
        const int parts = (I + 16383) / 16384, per = ((I / 8 + parts - 1) / parts) * 8;
        for (int k0 = 0; k0 < I; k0 += per) {
          tgx::GemvArgs p = a;
          p.W = w.wdown + (size_t)k0 * c->esz; p.x = r.h + k0; p.K = std::min(per, I - k0); p.ldw = I;
          if (k0) p.bias = nullptr;
          launch_gemv<tgx::PRO_PLAIN, tgx::EPI_RESIDUAL>(c, p, TGX_KERNEL_DOWN, R);
        }
        break;
      }
      if (resid_fixed(c, R, kv_stride)) a.res_acc = c->slab_acc + (size_t)(&r - c->rows.data()) * H;   // x = fp32(acc) + down(h); acc <- 0
      launch_gemv<tgx::PRO_PLAIN, tgx::EPI_RESIDUAL>(c, a, TGX_KERNEL_DOWN, R);
      break;
    }
    default: return 0;
  }
  return 1;
}

// All decoder layers for R rows: their current tokens' embeddings sit in the x slab, positions in the pos slab.
// == for (auto& layer : layers_) x = layer->forward(x)  (GPTModel.h:53-55)
void launch_layers(tgx_ctx* c, RowState* rv, int R, long long kv_stride) {
  for (int l = 0; l < c->d.layers; l++)
    for (int cls = TGX_KERNEL_QKV; cls <= TGX_KERNEL_DOWN; cls++) launch_layer_kernel(c, rv, R, l, cls, rv[0].x, kv_stride);
}
void launch_layers(tgx_ctx* c, int row0, int R) { launch_layers(c, &c->rows[(size_t)row0], R, (long long)c->kv_row_elems); }

// model.norm -> lm_head on the current position + per-workgroup argmax partials   (GPTModel.h:56-57)
void launch_lm_head(tgx_ctx* c, int row0, int R) {
  const tgx_model_desc& d = c->d;
  RowState& r = c->rows[(size_t)row0];
  tgx::GemvArgs a{};
  fill_strides(c, a);
  a.W = d.tied ? c->embed : c->lm_head; a.x = r.x; a.x_stride = d.hidden; a.norm_w = c->final_norm; a.eps = d.norm_eps;
  a.N = d.vocab; a.K = d.hidden; a.units = (d.vocab + 1) / 2; a.hd = 2;
  a.logits = r.logits; a.part_val = r.part_val; a.part_idx = r.part_idx;
  if (c->gpt2) {   // ln_f -> wte^T (tied head, ModelGPT2.h:170-176)
    a.norm_b = c->final_norm_b;
    launch_gemv<tgx::PRO_LAYERNORM, tgx::EPI_LOGITS>(c, a, TGX_KERNEL_LMHEAD, R);
    return;
  }
  launch_gemv<tgx::PRO_RMSNORM, tgx::EPI_LOGITS>(c, a, TGX_KERNEL_LMHEAD, R);
}

tgx::FinalizeArgs make_finalize_args(tgx_ctx* c, int row, bool advance_pos, bool log_step) {
  RowState& r = c->rows[(size_t)row];
  tgx::FinalizeArgs a{};
  a.part_val = r.part_val; a.part_idx = r.part_idx; a.n_part = c->lm_grid;
  a.tok = r.tok; a.pos = r.pos; a.step = c->step; a.tok_log = c->tok_log; a.host_ring = c->mirror_to_host ? c->host_ring_dev : nullptr;
  a.log_cap = c->log_cap; a.ring_cap = HOST_RING;
  a.row = row; a.rows = c->batch;
  a.log = log_step ? 1 : 0; a.bump_step = (row == c->batch - 1) ? 1 : 0;
  if (log_step && c->batch > 1) { a.done = c->step_done; a.done_total = c->batch; a.bump_step = 0; }      // rows of a batch may be finalized concurrently: the last to count itself moves the step
  a.embed = c->embed; a.x = r.x; a.H = c->d.hidden; a.V = c->d.vocab; a.advance_pos = advance_pos ? 1 : 0;
  a.wpe = c->gpt2 ? c->wpe : nullptr; a.n_pos = c->d.n_positions > 0 ? c->d.n_positions : 1;
  return a;
}

// == argmax (Sampler.cpp:26-29) + token publish / pastLength / next embedding: one finalize launch per row
void launch_finalize_greedy(tgx_ctx* c, int row0, int R, bool advance_pos, bool log_step) {
  for (int b = row0; b < row0 + R; b++) {
    const tgx::FinalizeArgs a = make_finalize_args(c, b, advance_pos, log_step);
    TGX_DT_SWITCH(c->dt, hipLaunchKernelGGL(tgx::finalize_greedy_kernel<DT>, dim3(1), dim3(256), 0, c->stream, a))
  }
}

// the batched step: rows are finalized concurrently; the row that completes the batch's count moves the step counter (FinalizeArgs.done)
void launch_finalize_rows(tgx_ctx* c, int row0, int M) {
  tgx::FinalizeRowsArgs fa{};
  fa.f = make_finalize_args(c, row0, /*advance_pos=*/true, /*log_step=*/true);
  fa.part_stride = c->lm_grid; fa.x_stride = c->d.hidden;
  TGX_DT_SWITCH(c->dt, hipLaunchKernelGGL(tgx::finalize_rows_kernel<DT>, dim3(M), dim3(256), 0, c->stream, fa))
}

// prefill by steps: chunk row r <- embedding of prompt token r at position pos0 + r
void launch_embed_chunk(tgx_ctx* c, const long long* ids, int R, int pos0) {
  tgx::EmbedChunkArgs e{};
  e.ids = ids; e.embed = c->embed; e.x = c->ch_x; e.H = c->d.hidden; e.pos = c->ch_pos; e.pos0 = pos0; e.wpe = c->gpt2 ? c->wpe : nullptr;
  TGX_DT_SWITCH(c->dt, hipLaunchKernelGGL(tgx::embed_chunk_kernel<DT>, dim3(R), dim3(256), 0, c->stream, e))
}
void launch_add_pos(tgx_ctx* c, int* pos, int n) { hipLaunchKernelGGL(tgx::add_pos_kernel, dim3(1), dim3(64), 0, c->stream, pos, n); }
void launch_argmax_partials(tgx_ctx* c, const float* logits, int V, float* part_val, int* part_idx) {
  hipLaunchKernelGGL(tgx::argmax_partials_kernel, dim3(c->lm_grid), dim3(256), 0, c->stream, logits, V, part_val, part_idx);
}
