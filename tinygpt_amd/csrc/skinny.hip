// skinny.hip — the batched decode step and short prompts of the MI355X shim on the skinny MFMA GEMMs (kernels/skinny.h, skinny_ksplit.h, skinny_dma.h):
// every nn::Linear of a [B,1] step (or of a prompt of <= 128 rows) is ONE pass over its weights (GPTEngine.cpp:154-168: the reference runs the whole
// batch through each Linear).
#include "ctx.h"
#include "kernels/attn_decode_mfma.h"
#include "kernels/prefill.h"
#include "kernels/skinny.h"
#include "kernels/skinny_ksplit.h"
#include "kernels/skinny_dma.h"

// ---- batched decode on the matrix cores (kernels/skinny.h) ---------------------------------------------------------------------------
// the (epilogue, terms, activation source) combinations the batched step uses; every one exists for 2 dtypes x MB 1,2 x NBW 1,2
#define TGX_SKINNY_COMBOS(X)                                                                                                   \
  X(tgx::GEMM_PARTIAL, 3, 2) X(tgx::GEMM_STORE, 3, 2) X(tgx::GEMM_PARTIAL, 2, 2) X(tgx::GEMM_STORE, 2, 2) X(tgx::GEMM_SILU, 2, 2) \
  X(tgx::GEMM_PARTIAL, 2, 1) X(tgx::GEMM_RESIDUAL, 2, 1) X(tgx::GEMM_PARTIAL, 2, 0) X(tgx::GEMM_RESIDUAL, 2, 0) X(tgx::GEMM_SILU, 2, 0) X(tgx::GEMM_STORE, 2, 0) \
  X(tgx::GEMM_PARTIAL, 3, 0) X(tgx::GEMM_STORE, 3, 0)

template <int DT, int EPI, int NT, int ASRC>
static int skinny_set_attr_dt(tgx_ctx* c) {
#define TGX_SK_A(MB_, CFG_) HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::skinny_gemm_kernel<DT, EPI, MB_, NT, CFG_, ASRC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::skinny_lds_bytes(MB_, NT, CFG_)));
  TGX_SK_A(1, 0) TGX_SK_A(1, 1) TGX_SK_A(1, 2) TGX_SK_A(2, 0) TGX_SK_A(2, 1) TGX_SK_A(2, 2)
  if constexpr (ASRC == 0) { TGX_SK_A(4, 0) TGX_SK_A(4, 1) TGX_SK_A(4, 2) }
  if constexpr (ASRC == 1) { TGX_SK_A(4, 1) }
#undef TGX_SK_A
  return TGX_OK;
}
// the skinny GEMM's LDS image (weight tiles + activation panels) exceeds the 64 KB default for 32 rows
static int skinny_panel_set_attrs(tgx_ctx* c) {
  int rc;
#define X(E, N, A) if ((rc = skinny_set_attr_dt<tgx::DT_BF16, E, N, A>(c)) || (rc = skinny_set_attr_dt<tgx::DT_F16, E, N, A>(c))) return rc;
  TGX_SKINNY_COMBOS(X)
#undef X
  return TGX_OK;
}

template <int EPI, int NT, int ASRC>
static void skinny_dispatch(tgx_ctx* c, dim3 grid, int mb, int cfg, const tgx::GemmArgs& g) {
  const dim3 blk(256);
  const size_t lds = tgx::skinny_lds_bytes(mb, NT, cfg);
#define TGX_SK_L(MB_, CFG_) hipLaunchKernelGGL((tgx::skinny_gemm_kernel<DT, EPI, MB_, NT, CFG_, ASRC>), grid, blk, lds, c->stream, g)
  TGX_DT16_SWITCH(c->dt,
    if (mb == 4) {     // 33-64 rows (round 3): four activation blocks, geometries 0 and 2, stored terms only (staging with RMSNorm spills: 262 us for gate_up)
      if constexpr (ASRC == 0) { if (cfg == 2) TGX_SK_L(4, 2); else if (cfg == 1) TGX_SK_L(4, 1); else TGX_SK_L(4, 0); }
      else if constexpr (ASRC == 1) TGX_SK_L(4, 1);             // fp32 rows split on the way (the o_proj product of a decode step): 128-k panels only
      else c->launch_fault = "internal: 33-64-row skinny GEMM takes stored 16-bit terms or plain fp32 rows";
    }
    else if (mb == 2) { if (cfg == 2) TGX_SK_L(2, 2); else if (cfg == 1) TGX_SK_L(2, 1); else TGX_SK_L(2, 0); }
    else { if (cfg == 2) TGX_SK_L(1, 2); else if (cfg == 1) TGX_SK_L(1, 1); else TGX_SK_L(1, 0); })
#undef TGX_SK_L
}

// the LDS-DMA ring form of the products on stored terms (kernels/skinny_dma.h): same grid, same results
template <int EPI, int NT = 2>
static void skinny_dma_dispatch(tgx_ctx* c, dim3 grid, int mb, int nbw, const tgx::GemmArgs& g) {
  const dim3 blk(256);
  const size_t lds = tgx::skd_lds_bytes(mb, nbw, NT);
#define TGX_SKD_L(MB_, NBW_) hipLaunchKernelGGL((tgx::skinny_dma_kernel<DT, EPI, MB_, NBW_, NT>), grid, blk, lds, c->stream, g)
  TGX_DT16_SWITCH(c->dt,
    if (mb == 8) TGX_SKD_L(8, 1);
    else if (nbw == 2) { if (mb == 4) TGX_SKD_L(4, 2); else if (mb == 2) TGX_SKD_L(2, 2); else TGX_SKD_L(1, 2); }
    else { if (mb == 4) TGX_SKD_L(4, 1); else if (mb == 2) TGX_SKD_L(2, 1); else TGX_SKD_L(1, 1); })
#undef TGX_SKD_L
}
template <int DT, int EPI, int NT = 2>
static int skinny_dma_set_attr_dt(tgx_ctx* c) {
#define TGX_SKD_A(MB_, NBW_) HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::skinny_dma_kernel<DT, EPI, MB_, NBW_, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::skd_lds_bytes(MB_, NBW_, NT)));
  TGX_SKD_A(1, 1) TGX_SKD_A(2, 1) TGX_SKD_A(4, 1) TGX_SKD_A(1, 2) TGX_SKD_A(2, 2) TGX_SKD_A(4, 2) TGX_SKD_A(8, 1)
#undef TGX_SKD_A
  return TGX_OK;
}
static int skinny_dma_set_attrs(tgx_ctx* c) {
  int rc;
#define X(E) if ((rc = skinny_dma_set_attr_dt<tgx::DT_BF16, E>(c)) || (rc = skinny_dma_set_attr_dt<tgx::DT_F16, E>(c))) return rc;
  X(tgx::GEMM_PARTIAL) X(tgx::GEMM_RESIDUAL) X(tgx::GEMM_SILU) X(tgx::GEMM_STORE)
#undef X
  if ((rc = skinny_dma_set_attr_dt<tgx::DT_BF16, tgx::GEMM_PARTIAL, 3>(c)) || (rc = skinny_dma_set_attr_dt<tgx::DT_BF16, tgx::GEMM_STORE, 3>(c))) return rc;   // three terms: the bf16 QKV product
  return TGX_OK;
}

// One nn::Linear of a batched step: Y[M][N] = X[M][K] . W^T for M <= 32 activation rows, X given as 16-bit terms (asrc 0), fp32 rows
// (1) or fp32 rows to be RMS-normalised on the way (2).  Wide products (>= ~one 64-row group per CU) run unsplit with their epilogue;
// narrow ones (N = hidden, the QKV rows) split K over blockIdx.y into fp32 slabs — the return value is the number of slabs the caller's
// finishing kernel has to sum (1 = the epilogue already ran).
namespace {
struct SkinnyCall {
  int epi = tgx::GEMM_STORE;
  const ebyte* W = nullptr; const ebyte* bias = nullptr;
  float* C = nullptr; int ldc = 0;
  int M = 0, N = 0, K = 0;
  int nt = 2, asrc = 0;
  const bf16_t *a_hi = nullptr, *a_lo = nullptr, *a_lo2 = nullptr;
  const float* a_f32 = nullptr; int lda = 0;
  const ebyte* norm_w = nullptr; const float* ssq_in = nullptr;
  bool allow_split = true;
};
}  // namespace
static int launch_skinny(tgx_ctx* c, const SkinnyCall& k) {
  tgx::GemmArgs g{};
  g.A_hi = k.a_hi; g.A_lo = k.a_lo; g.A_lo2 = k.a_lo2; g.A_f32 = k.a_f32; g.lda = k.lda;
  if (c->act16 && c->ws_zero) {      // option act.round16: the stored lo terms read zeros.  (fp32-row sources: the callers below take the stored-term route for every
    // norm-fused product in this mode; the attention rows of asrc 1 arrive rounded, so their second term IS zero)
    if (g.A_lo) g.A_lo = c->ws_zero;
    if (g.A_lo2) g.A_lo2 = c->ws_zero;
    if (k.asrc == 2) { c->launch_fault = "internal: act.round16 needs stored terms for norm-fused skinny products"; return 1; }
  }
  g.norm_w = reinterpret_cast<const bf16_t*>(k.norm_w); g.ssq_part = k.ssq_in; g.ssq_ncb = tgx::SK_NCB; g.eps = c->d.norm_eps;
  g.inter = k.N / 2; g.out_hi = c->ws_hh; g.out_lo = c->ws_hl;
  g.B = reinterpret_cast<const bf16_t*>(k.W); g.bias = reinterpret_cast<const bf16_t*>(k.bias); g.C = k.C; g.M = k.M; g.N = k.N; g.K = k.K; g.ldc = k.ldc;
  const int mb = k.M > 64 ? 8 : (k.M > 32 ? 4 : (k.M > 16 ? 2 : 1));       // eight blocks (65-128 rows): the LDS-DMA ring kernel only
  // 128-row groups when they alone oversubscribe the chip (the lm_head), else 64-row groups: twice the workgroups for the same bytes
  int cfg = (k.N + 127) / 128 >= 2 * c->num_cus ? 2 : c->skinny_cfg_mid;
  if (c->skinny_cfg_force >= 0) cfg = c->skinny_cfg_force;
  if (mb == 4 && k.asrc == 1) cfg = 1;
  if (mb == 4 && k.nt == 3 && cfg == 0) cfg = 1;     // three terms x four blocks: the 256-k panel's register image spills (80 us for the QKV product); 128-k panels
  const int kp = tgx::skinny_kp(cfg);
  const int panels = (k.K + kp - 1) / kp;
  const int gx = (k.N + tgx::skinny_rows(cfg) - 1) / tgx::skinny_rows(cfg);
  // split K until ~skinny_wgs workgroups exist (default two per CU: the bytes in flight per CU are what the stream rate follows)
  int nsplit = 1;
  if (k.allow_split && c->gemm_splitk && gx < c->skinny_wgs) nsplit = std::max(1, std::min(std::min(16, panels), (c->skinny_wgs + gx / 2) / gx));
  if (nsplit > 1 && (size_t)nsplit * k.M * k.N * 4 > c->ws_part_bytes) nsplit = 1;      // the slab buffer is sized before capture (ensure_skinny_ws)
  int epi = k.epi;
  if (nsplit > 1) {
    g.part = c->ws_part; g.nsplit = nsplit; g.interleave = k.epi == tgx::GEMM_SILU ? 1 : 0;
    g.k_per = ((panels + nsplit - 1) / nsplit) * kp;
    nsplit = (k.K + g.k_per - 1) / g.k_per;      // splits that actually hold a K range
    g.nsplit = nsplit;
    epi = tgx::GEMM_PARTIAL;
  }
  const dim3 grid(gx, nsplit);
  if (c->skinny_dma && k.asrc == 0 && k.nt == 3 && k.a_lo2 && c->dt == tgx::DT_BF16 && k.M >= c->skinny_dma_rows && k.K % 64 == 0 && (nsplit == 1 || g.k_per % 64 == 0) &&
      (epi == tgx::GEMM_PARTIAL || epi == tgx::GEMM_STORE)) {
    const int nbw = mb == 8 ? 1 : (c->skinny_dma_nbw ? c->skinny_dma_nbw : tgx::skinny_nbw(cfg));
    const dim3 grid((k.N + 64 * nbw - 1) / (64 * nbw), nsplit);
    const dim3 blk(256);
    const size_t lds = tgx::skd_lds_bytes(mb, nbw, 3);
    if (mb == 8) {
      if (epi == tgx::GEMM_PARTIAL) hipLaunchKernelGGL((tgx::skinny_dma_kernel<tgx::DT_BF16, tgx::GEMM_PARTIAL, 8, 1, 3>), grid, blk, lds, c->stream, g);
      else hipLaunchKernelGGL((tgx::skinny_dma_kernel<tgx::DT_BF16, tgx::GEMM_STORE, 8, 1, 3>), grid, blk, lds, c->stream, g);
      return nsplit;
    }
#define TGX_SKD3(E_) do { if (nbw == 2) { if (mb == 4) hipLaunchKernelGGL((tgx::skinny_dma_kernel<tgx::DT_BF16, E_, 4, 2, 3>), grid, blk, lds, c->stream, g); \
                                          else if (mb == 2) hipLaunchKernelGGL((tgx::skinny_dma_kernel<tgx::DT_BF16, E_, 2, 2, 3>), grid, blk, lds, c->stream, g); \
                                          else hipLaunchKernelGGL((tgx::skinny_dma_kernel<tgx::DT_BF16, E_, 1, 2, 3>), grid, blk, lds, c->stream, g); } \
                          else { if (mb == 4) hipLaunchKernelGGL((tgx::skinny_dma_kernel<tgx::DT_BF16, E_, 4, 1, 3>), grid, blk, lds, c->stream, g); \
                                 else if (mb == 2) hipLaunchKernelGGL((tgx::skinny_dma_kernel<tgx::DT_BF16, E_, 2, 1, 3>), grid, blk, lds, c->stream, g); \
                                 else hipLaunchKernelGGL((tgx::skinny_dma_kernel<tgx::DT_BF16, E_, 1, 1, 3>), grid, blk, lds, c->stream, g); } } while (0)
    if (epi == tgx::GEMM_PARTIAL) TGX_SKD3(tgx::GEMM_PARTIAL); else TGX_SKD3(tgx::GEMM_STORE);
#undef TGX_SKD3
    return nsplit;
  }
  if (c->skinny_dma && k.asrc == 0 && k.nt == 2 && k.M >= c->skinny_dma_rows && k.K % 64 == 0 && (nsplit == 1 || g.k_per % 64 == 0)) {
    const int nbw = mb == 8 ? 1 : (c->skinny_dma_nbw ? c->skinny_dma_nbw : tgx::skinny_nbw(cfg));
    const dim3 grid((k.N + 64 * nbw - 1) / (64 * nbw), nsplit);
    switch (epi) {
      case tgx::GEMM_PARTIAL: skinny_dma_dispatch<tgx::GEMM_PARTIAL>(c, grid, mb, nbw, g); return nsplit;
      case tgx::GEMM_RESIDUAL: skinny_dma_dispatch<tgx::GEMM_RESIDUAL>(c, grid, mb, nbw, g); return nsplit;
      case tgx::GEMM_SILU: skinny_dma_dispatch<tgx::GEMM_SILU>(c, grid, mb, nbw, g); return nsplit;
      case tgx::GEMM_STORE: skinny_dma_dispatch<tgx::GEMM_STORE>(c, grid, mb, nbw, g); return nsplit;
      default: break;
    }
  }
  if (mb == 8) { c->launch_fault = "internal: 65-128 activation rows need the LDS-DMA ring kernel (stored terms, K a multiple of 64)"; return 1; }
  bool launched = false;
#define X(E, N, A) if (!launched && epi == E && k.nt == N && k.asrc == A) { skinny_dispatch<E, N, A>(c, grid, mb, cfg, g); launched = true; }
  TGX_SKINNY_COMBOS(X)
#undef X
  if (!launched) { c->launch_fault = "internal: skinny GEMM combination not instantiated"; return 1; }
  return nsplit;
}

// finishes a split product into C (store / residual add) and leaves the rows' partial sums of squares for the next RMSNorm-fused product
static void launch_reduce_rows(tgx_ctx* c, int epi, int nsplit, const ebyte* bias, float* C, int ldc, int M, int N, float* ssq_out) {
  tgx::GemmArgs g{};
  g.part = c->ws_part; g.nsplit = nsplit; g.bias = reinterpret_cast<const bf16_t*>(bias); g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.ssq_out = ssq_out;
  const dim3 grid(M, tgx::SK_NCB), blk(256);
  TGX_DT16_SWITCH(c->dt,
    if (epi == tgx::GEMM_RESIDUAL) hipLaunchKernelGGL((tgx::reduce_rows_kernel<DT, tgx::GEMM_RESIDUAL>), grid, blk, 0, c->stream, g);
    else hipLaunchKernelGGL((tgx::reduce_rows_kernel<DT, tgx::GEMM_STORE>), grid, blk, 0, c->stream, g);)
}

// rows beyond 4 of a decode batch take the matrix-core path when the model has 16-bit storage and tile-friendly shapes
bool decode_mfma_ok(const tgx_ctx* c) {
  return c->batch >= c->decode_mfma_min && c->dt != tgx::DT_F32 && !c->gpt2 && prefill_shapes_ok(c->d) && c->d.vocab >= 128;
}

// workspace of the batched step, sized before the step is captured: qkv rows, siluMul terms, split-K slabs, sums of squares
int ensure_skinny_ws(tgx_ctx* c, int rows) {
  int rc = ensure_prefill_ws(c, rows);
  if (rc) return rc;
  const tgx_model_desc& d = c->d;
  const size_t widest = std::max<size_t>((size_t)d.heads * d.head_dim + 2 * (size_t)d.kv_heads * d.head_dim, (size_t)2 * d.inter);
  const size_t need = (size_t)16 * rows * std::max<size_t>(widest, (size_t)d.hidden) * 4;       // up to 16 K splits
  if (need > c->ws_part_bytes) {
    drop_step_graphs(c);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    if (c->ws_part) (void)hipFree(c->ws_part);
    c->ws_part = nullptr; c->ws_part_bytes = 0;
    HIP_OK(c, hipMalloc((void**)&c->ws_part, need));
    c->ws_part_bytes = need;
  }
  if (!c->ws_ssq) HIP_OK(c, hipMalloc((void**)&c->ws_ssq, (size_t)128 * tgx::SK_NCB * 4));
  return TGX_OK;
}

// The wide products of a batch of <= 32 rows on the barrier-free K-split kernel (kernels/skinny_ksplit.h); activations = the 16-bit terms
// rmsnorm_split_kernel left in ws_ah / ws_al.  false: shape not covered (the caller takes the panel kernel).
static bool ksplit_ok(const tgx_ctx* c, int M, int N, int K) {
  // 17-32 rows (two activation blocks, two weight slots), ms/step panel / K-split kernel (round 3 closing build): Llama-3.2-1B (K = 2048) B = 17 1.314 / 1.253,
  // 24 1.275 / 1.237, 32 1.316 / 1.312; Llama-3.2-3B (K = 3072) 2.834 / 2.853, 2.891 / 2.984, 3.015 / 3.199; Mistral-7B (K = 4096) B = 32 5.11 / 5.38: at K = 2048
  // only (option value 2: always)
  const int max_rows = c->skinny_ksplit >= 2 || K == 2048 ? 32 : 16;
  return c->skinny_ksplit && M <= max_rows && K % 256 == 0 && K >= 768 && N >= 64 * c->num_cus;
}
static void launch_ksplit(tgx_ctx* c, int epi, const ebyte* W, float* C, int ldc, int M, int N, int K) {
  tgx::GemmArgs g{};
  g.A_hi = c->ws_ah; g.A_lo = (c->act16 && c->ws_zero) ? c->ws_zero : c->ws_al; g.B = reinterpret_cast<const bf16_t*>(W); g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.inter = N / 2; g.out_hi = c->ws_hh; g.out_lo = c->ws_hl;
  const dim3 grid((N + 63) / 64), blk(256);
#define TGX_KS(E_, K_) do { if (M > 16) hipLaunchKernelGGL((tgx::skinny_ksplit_kernel<DT, E_, K_, 2>), grid, blk, 0, c->stream, g); \
                            else hipLaunchKernelGGL((tgx::skinny_ksplit_kernel<DT, E_, K_, 1>), grid, blk, 0, c->stream, g); } while (0)
#define TGX_KS_K(E_) do { if (K == 2048) TGX_KS(E_, 2048); else if (K == 3072) TGX_KS(E_, 3072); else if (K == 4096) TGX_KS(E_, 4096); else TGX_KS(E_, 0); } while (0)
  TGX_DT16_SWITCH(c->dt, if (epi == tgx::GEMM_SILU) TGX_KS_K(tgx::GEMM_SILU); else TGX_KS_K(tgx::GEMM_STORE);)
#undef TGX_KS_K
#undef TGX_KS
}
// One decode step for rows [row0, row0 + M), M <= 32, with every nn::Linear as ONE pass over its weights (GPTEngine.cpp:154-168: the
// reference runs the whole [B,1] batch through each Linear).  Same per-row math as the GEMV path in the prefill's arithmetic: fp32
// activations enter the matrix cores as exact sums of 16-bit terms (three for the QKV product, whose K/V results are rounded into the cache).
// Per layer: qkv product (RMSNorm applied while staging) -> {sum slabs, bias, RoPE, cache append} -> attention -> o_proj product on the
// fp32 attention output -> {sum slabs, residual, sums of squares} -> gate_up product (RMSNorm while staging, siluMul epilogue) ->
// down product -> {sum slabs, residual, sums of squares}: 7 launches (8 with the split-form attention's combine).
void launch_decode_step_mfma(tgx_ctx* c, int row0, int M, const tgx_sampler_cfg& cfg) {
  const tgx_model_desc& d = c->d;
  const int H = d.hidden, I = d.inter, hd = d.head_dim, qd = d.heads * hd, kvd = d.kv_heads * hd, V = d.vocab;
  const size_t kv_layer = (c->kv_paged ? (size_t)c->kv_nblocks * d.kv_heads * tgx::KV_BLOCK : (size_t)d.kv_heads * d.max_ctx) * hd * c->esz;      // paged KV: a layer's pool
  const long long kvs = c->kv_paged ? 0 : (long long)c->kv_row_elems, tbs = c->kv_paged ? c->kv_tbl_stride : 0;
  RowState& r = c->rows[(size_t)row0];
  const int nt_qkv = c->dt == tgx::DT_BF16 ? 3 : 2;
  float* ssq = c->ws_ssq;
  const bool lm_ks = ksplit_ok(c, M, V, H);
  // 33-64 rows (round 3): four activation blocks; every RMSNorm-fused product takes its activations as 16-bit terms prepared once per product by the
  // row-wise launch that also adds the pending split-K residual (the RMSNorm-on-the-way staging runs out of registers at four blocks)
  const bool terms = M > 32 || c->act16 || ((c->skinny_terms >= 2 || (c->skinny_dma && c->skinny_dma_qkv && M >= c->skinny_dma_rows && H % 64 == 0)) && M > (c->skinny_dma_qkv >= 2 ? c->skinny_terms_above : 16));      // (act.round16: the row-wise launch's first term IS the rounded input)
  int pend = 0;            // terms form: slabs of the previous layer's down product not yet added to the rows
  // the rows start as embedding rows (the finalize of the previous step gathered them): their sums of squares for the first RMSNorm
  if (!terms) hipLaunchKernelGGL(tgx::row_ssq_kernel, dim3(M, tgx::SK_NCB), dim3(256), 0, c->stream, (const float*)r.x, (long long)H, H, ssq);
  for (int l = 0; l < d.layers; l++) {
    const LayerW& w = c->L[(size_t)l];
    SkinnyCall q;
    q.epi = tgx::GEMM_STORE; q.W = w.wqkv; q.bias = w.bqkv; q.C = c->ws_out; q.ldc = qd + 2 * kvd; q.M = M; q.N = qd + 2 * kvd; q.K = H;
    q.nt = nt_qkv; q.asrc = 2; q.a_f32 = r.x; q.lda = H; q.norm_w = w.in_norm; q.ssq_in = ssq;
    if (terms) {
      launch_norm_terms(c, r.x, w.in_norm, M, H, pend, nt_qkv == 3); pend = 0;
      q.asrc = 0; q.a_hi = c->ws_ah; q.a_lo = c->ws_al; q.a_lo2 = c->ws_al2; q.a_f32 = nullptr; q.norm_w = nullptr; q.ssq_in = nullptr;
    }
    const int qs = launch_skinny(c, q);
    // the QKV product's finish (slab sums + bias, q / k norm, RoPE, cache append) inside the attention launch when that is the batched matrix-core form
    // (option attn.raw_fuse): one launch per layer less
    const bool raw_fuse = c->attn_raw_fuse && c->attn_direct && !(c->debug_skip & 1) && !(d.qk_norm && hd != 128) &&
                          (attn_batch_on_mfma(c, M) ? d.heads / d.kv_heads <= tgx::ATTN_RAW_GMAX : c->attn_raw_fuse >= 2);      // 2: the VALU direct forms as well
    // the direct-form attention of the step leaves its rows as 16-bit terms for the o_proj product (option skinny.dma_oproj: 1 = the matrix-core form only, 2 = every direct form)
    const bool attn_terms = c->skinny_dma && c->skinny_dma_oproj && c->attn_direct && (c->skinny_dma_oproj >= 2 || attn_batch_on_mfma(c, M)) && !(c->debug_skip & 1) &&
                            M >= c->skinny_dma_rows && qd % 64 == 0;
    if (!raw_fuse) {
      tgx::RopeRowsArgs a{};
      if (qs > 1) { a.part = c->ws_part; a.nsplit = qs; a.bias = w.bqkv; } else a.QKV = c->ws_out;
      a.rows = M; a.q_out = r.q; a.q_stride = qd; a.k_cache = r.kcache + (size_t)l * kv_layer; a.v_cache = r.vcache + (size_t)l * kv_layer;
      a.kv_stride = kvs; a.blk_tbl = r.tbl; a.tbl_stride = tbs; a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.pos = r.pos;
      a.heads = d.heads; a.kv_heads = d.kv_heads; a.hd = hd; a.max_ctx = d.max_ctx;
      a.q_norm_w = d.qk_norm ? w.q_norm : nullptr; a.k_norm_w = d.qk_norm ? w.k_norm : nullptr; a.eps = d.norm_eps;
      TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::rope_kv_rows_kernel<DT>, dim3(M, d.heads + 2 * d.kv_heads), dim3(64), 0, c->stream, a))
    }
    {
      tgx::AttnArgs a{};
      a.q = r.q; a.k_cache = r.kcache + (size_t)l * kv_layer; a.v_cache = r.vcache + (size_t)l * kv_layer;
      a.pos = r.pos; a.part = r.attn_part; a.out = r.attn;
      a.heads = d.heads; a.kv_heads = d.kv_heads; a.max_ctx = d.max_ctx; a.nsplit = c->attn_nsplit;
      a.scale = 1.0f / sqrtf((float)hd);
      a.q_stride = qd; a.kv_stride = kvs; a.blk_tbl = r.tbl; a.tbl_stride = tbs; a.part_stride = (long long)c->attn_part_row; a.dbg = c->debug_attn;
      a.act16 = c->act16 ? (c->dt == tgx::DT_BF16 ? 1 : (c->dt == tgx::DT_F16 ? 2 : 0)) : 0;
      if (raw_fuse) {
        if (qs > 1) { a.raw_part = c->ws_part; a.raw_nsplit = qs; a.raw_bias = w.bqkv; } else a.raw_qkv = c->ws_out;
        a.raw_rows = M; a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.eps = d.norm_eps;
        a.q_norm_w = d.qk_norm ? w.q_norm : nullptr; a.k_norm_w = d.qk_norm ? w.k_norm : nullptr;
      }
      if (attn_terms) { a.out_hi = c->ws_ah; a.out_lo = c->ws_al; }
      launch_attn(c, a, M);
    }
    SkinnyCall o;
    o.epi = tgx::GEMM_RESIDUAL; o.W = w.wo; o.C = r.x; o.ldc = H; o.M = M; o.N = H; o.K = qd; o.nt = 2; o.asrc = 1; o.a_f32 = r.attn; o.lda = qd;
    if (attn_terms) { o.asrc = 0; o.a_hi = c->ws_ah; o.a_lo = c->ws_al; o.a_f32 = nullptr; }
    const int os = launch_skinny(c, o);
    const bool gu_dma = c->skinny_dma && M >= c->skinny_dma_rows && M > 16 && H % 64 == 0;     // 17+ rows: the LDS-DMA ring kernel on stored terms (16.8 vs 19.5 us at 32 rows)
    const bool gu_ks = !gu_dma && ksplit_ok(c, M, 2 * I, H);
    int gs = 1;
    if (gu_ks) {     // {sum slabs, residual, RMSNorm, 16-bit terms} in one row-wise launch, then the barrier-free wide product
      launch_norm_terms(c, r.x, w.post_norm, M, H, os);
      launch_ksplit(c, tgx::GEMM_SILU, w.wgu, nullptr, 2 * I, M, 2 * I, H);
    } else if (c->skinny_terms || terms || gu_dma) {
      // 17-32 rows (round 3): {sum slabs, residual, RMSNorm, 16-bit terms} ONCE per layer in the row-wise launch that replaces reduce_rows; the panel
      // kernel then stages stored terms instead of normalising and splitting every 256-k panel in each of its 256 workgroups
      launch_norm_terms(c, r.x, w.post_norm, M, H, os);
      SkinnyCall gu;
      gu.epi = tgx::GEMM_SILU; gu.W = w.wgu; gu.M = M; gu.N = 2 * I; gu.K = H; gu.ldc = 2 * I; gu.nt = 2; gu.asrc = 0; gu.a_hi = c->ws_ah; gu.a_lo = c->ws_al;
      gu.allow_split = c->skinny_gu_split != 0;
      gs = launch_skinny(c, gu);
    } else {
    if (os > 1) launch_reduce_rows(c, tgx::GEMM_RESIDUAL, os, nullptr, r.x, H, M, H, ssq);
    else hipLaunchKernelGGL(tgx::row_ssq_kernel, dim3(M, tgx::SK_NCB), dim3(256), 0, c->stream, (const float*)r.x, (long long)H, H, ssq);
    SkinnyCall gu;
    gu.epi = tgx::GEMM_SILU; gu.W = w.wgu; gu.M = M; gu.N = 2 * I; gu.K = H; gu.ldc = 2 * I; gu.nt = 2; gu.asrc = 2; gu.a_f32 = r.x; gu.lda = H;
    gu.norm_w = w.post_norm; gu.ssq_in = ssq; gu.allow_split = c->skinny_gu_split != 0;
    gs = launch_skinny(c, gu);                                             // -> ws_hh / ws_hl: the down product's activation terms
    }
    if (gs > 1) launch_silu_slab_reduce(c, M, I, gs);                       // slabs -> siluMul -> terms (z-ordered sums)
    SkinnyCall dn;
    dn.epi = tgx::GEMM_RESIDUAL; dn.W = w.wdown; dn.C = r.x; dn.ldc = H; dn.M = M; dn.N = H; dn.K = I; dn.nt = 2; dn.asrc = 0; dn.a_hi = c->ws_hh; dn.a_lo = c->ws_hl;
    const int ds = launch_skinny(c, dn);
    if (l + 1 == d.layers && (lm_ks || terms)) launch_norm_terms(c, r.x, c->final_norm, M, H, ds);      // the last residual goes straight into model.norm's terms
    else if (terms) pend = ds;                                                                         // the next layer's norm launch adds the slabs
    else if (ds > 1) launch_reduce_rows(c, tgx::GEMM_RESIDUAL, ds, nullptr, r.x, H, M, H, ssq);
    else hipLaunchKernelGGL(tgx::row_ssq_kernel, dim3(M, tgx::SK_NCB), dim3(256), 0, c->stream, (const float*)r.x, (long long)H, H, ssq);
  }
  if (lm_ks) {
    launch_ksplit(c, tgx::GEMM_STORE, d.tied ? c->embed : c->lm_head, r.logits, V, M, V, H);
  } else {
  SkinnyCall lm;
  lm.epi = tgx::GEMM_STORE; lm.W = d.tied ? c->embed : c->lm_head; lm.C = r.logits; lm.ldc = V; lm.M = M; lm.N = V; lm.K = H; lm.nt = 2; lm.asrc = 2;
  lm.a_f32 = r.x; lm.lda = H; lm.norm_w = c->final_norm; lm.ssq_in = ssq; lm.allow_split = false;
  if (terms) { lm.asrc = 0; lm.a_hi = c->ws_ah; lm.a_lo = c->ws_al; lm.a_f32 = nullptr; lm.norm_w = nullptr; lm.ssq_in = nullptr; }
  launch_skinny(c, lm);
  }
  hipLaunchKernelGGL(tgx::argmax_partials_rows_kernel, dim3(std::min(c->lm_grid, std::max(1, (V / 4 + 255) / 256)), M), dim3(256), 0, c->stream, (const float*)r.logits, (long long)V, V, r.part_val, r.part_idx,
                     (long long)c->lm_grid, c->lm_grid);
  if (is_greedy(&cfg)) {
    launch_finalize_rows(c, row0, M);
  } else {
    launch_sample(c, row0, M, cfg, /*advance_pos=*/true, /*log_step=*/true);
  }
}

// Prompts of a few tokens (NB * S <= 32 workspace rows): the batched prefill with every product as a skinny MFMA GEMM (kernels/skinny.h) —
// the 128-row tiles of gemm_x2_kernel would stream the weights for 4-25 % useful rows through a two-barrier K loop; here the weight stream
// is the decode step's, RMSNorm rides in the activation staging and narrow products finish through the row-wise slab reducers.
void launch_prefill_skinny(tgx_ctx* c, int row0, int NB, int S) {
  const tgx_model_desc& d = c->d;
  const int H = d.hidden, I = d.inter, hd = d.head_dim, qd = d.heads * hd, kvd = d.kv_heads * hd;
  const size_t kv_layer = c->kv_paged ? (size_t)c->kv_nblocks * d.kv_heads * tgx::KV_BLOCK * hd : (size_t)d.kv_heads * d.max_ctx * hd;      // elements (paged KV: a layer's pool)
  const int M = NB * S, nq = qd + 2 * kvd;
  const int nt_qkv = c->dt == tgx::DT_BF16 ? 3 : 2;
  float* ssq = c->ws_ssq;
  launch_embed_rows(c, (const long long*)c->rows[(size_t)row0].prompt, c->ws_x, M, S);
  // 33-64 rows (four activation blocks): RMSNorm + the 16-bit terms once per product in a row-wise launch (which also takes the pending split-K
  // residual), the panel kernel stages stored terms — its RMSNorm-on-the-way form runs out of registers at four blocks
  const bool terms = M > c->prefill_terms_rows || c->act16;      // (option prefill.terms_rows)
  int pend = 0;             // slabs of the previous layer's down product not yet added to ws_x (terms form)
  if (!terms) hipLaunchKernelGGL(tgx::row_ssq_kernel, dim3(M, tgx::SK_NCB), dim3(256), 0, c->stream, (const float*)c->ws_x, (long long)H, H, ssq);
  for (int l = 0; l < d.layers; l++) {
    const LayerW& w = c->L[(size_t)l];
    SkinnyCall q;
    q.epi = tgx::GEMM_STORE; q.W = w.wqkv; q.bias = w.bqkv; q.C = c->ws_out; q.ldc = nq; q.M = M; q.N = nq; q.K = H;
    q.nt = nt_qkv; q.asrc = 2; q.a_f32 = c->ws_x; q.lda = H; q.norm_w = w.in_norm; q.ssq_in = ssq;
    if (terms) {
      launch_norm_terms(c, c->ws_x, w.in_norm, M, H, pend, nt_qkv == 3); pend = 0;
      q.asrc = 0; q.a_hi = c->ws_ah; q.a_lo = c->ws_al; q.a_lo2 = c->ws_al2; q.a_f32 = nullptr; q.norm_w = nullptr; q.ssq_in = nullptr;
    }
    const int qs = launch_skinny(c, q);
    for (int b = 0; b < NB; b++) {
      RowState& r = c->rows[(size_t)(row0 + b)];
      const size_t ro = (size_t)b * S;
      tgx::RopeKvArgs a{};
      a.QKV = c->ws_out + ro * nq; a.q_hi = c->ws_qh + ro * qd; a.q_lo = c->ws_ql + ro * qd;
      // a K-split product: the RoPE launch sums the slabs itself (z order + bias: what reduce_rows_kernel<GEMM_STORE> did in its own launch until round 6)
      if (qs > 1) { a.QKV = nullptr; a.part = c->ws_part + ro * nq; a.nsplit = qs; a.slab = (long long)M * nq; a.bias = reinterpret_cast<const bf16_t*>(w.bqkv); }
      a.k_cache = reinterpret_cast<bf16_t*>(r.kcache) + (size_t)l * kv_layer; a.v_cache = reinterpret_cast<bf16_t*>(r.vcache) + (size_t)l * kv_layer;
      a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin;
      a.heads = d.heads; a.kv_heads = d.kv_heads; a.hd = hd; a.max_ctx = d.max_ctx; a.past = (int)c->past; a.blk_tbl = r.tbl;
      a.q_norm_w = d.qk_norm ? (const bf16_t*)w.q_norm : nullptr; a.k_norm_w = d.qk_norm ? (const bf16_t*)w.k_norm : nullptr; a.eps = d.norm_eps;
      launch_rope_kv_split(c, a, S);
    }
    for (int b = 0; b < NB; b++) {
      RowState& r = c->rows[(size_t)(row0 + b)];
      const size_t ro = (size_t)b * S;
      tgx::AttnPrefillArgs a{};
      a.q_hi = c->ws_qh + ro * qd; a.q_lo = c->ws_ql + ro * qd;
      a.k_cache = reinterpret_cast<bf16_t*>(r.kcache) + (size_t)l * kv_layer; a.v_cache = reinterpret_cast<bf16_t*>(r.vcache) + (size_t)l * kv_layer;
      a.o_hi = c->ws_ah + ro * qd; a.o_lo = c->ws_al + ro * qd; a.S = S; a.heads = d.heads; a.kv_heads = d.kv_heads; a.max_ctx = d.max_ctx; a.past = (int)c->past;
      a.scale = 1.0f / sqrtf((float)hd); a.qblk_mirror = 1; a.blk_tbl = r.tbl;
      launch_attn_prefill(c, a, /*allow_lean=*/false);
    }
    SkinnyCall o;
    o.epi = tgx::GEMM_RESIDUAL; o.W = w.wo; o.C = c->ws_x; o.ldc = H; o.M = M; o.N = H; o.K = qd; o.nt = 2; o.asrc = 0; o.a_hi = c->ws_ah; o.a_lo = c->ws_al;
    const int os = launch_skinny(c, o);
    int gs = 1;
    if (ksplit_ok(c, M, 2 * I, H)) {       // prompts of <= 16 rows: as the batched decode step (the o_proj product has consumed ws_ah / ws_al by now)
      launch_norm_terms(c, c->ws_x, w.post_norm, M, H, os);
      launch_ksplit(c, tgx::GEMM_SILU, w.wgu, nullptr, 2 * I, M, 2 * I, H);
    } else if (terms) {
      launch_norm_terms(c, c->ws_x, w.post_norm, M, H, os);
      SkinnyCall gu;
      gu.epi = tgx::GEMM_SILU; gu.W = w.wgu; gu.M = M; gu.N = 2 * I; gu.K = H; gu.ldc = 2 * I; gu.nt = 2; gu.asrc = 0; gu.a_hi = c->ws_ah; gu.a_lo = c->ws_al;
      gu.allow_split = c->skinny_gu_split != 0;
      gs = launch_skinny(c, gu);
    } else {
    if (os > 1) launch_reduce_rows(c, tgx::GEMM_RESIDUAL, os, nullptr, c->ws_x, H, M, H, ssq);
    else hipLaunchKernelGGL(tgx::row_ssq_kernel, dim3(M, tgx::SK_NCB), dim3(256), 0, c->stream, (const float*)c->ws_x, (long long)H, H, ssq);
    SkinnyCall gu;
    gu.epi = tgx::GEMM_SILU; gu.W = w.wgu; gu.M = M; gu.N = 2 * I; gu.K = H; gu.ldc = 2 * I; gu.nt = 2; gu.asrc = 2; gu.a_f32 = c->ws_x; gu.lda = H;
    gu.norm_w = w.post_norm; gu.ssq_in = ssq; gu.allow_split = c->skinny_gu_split != 0;
    gs = launch_skinny(c, gu);
    }
    if (gs > 1) launch_silu_slab_reduce(c, M, I, gs);
    SkinnyCall dn;
    dn.epi = tgx::GEMM_RESIDUAL; dn.W = w.wdown; dn.C = c->ws_x; dn.ldc = H; dn.M = M; dn.N = H; dn.K = I; dn.nt = 2; dn.asrc = 0; dn.a_hi = c->ws_hh; dn.a_lo = c->ws_hl;
    const int ds = launch_skinny(c, dn);
    if (terms && l + 1 < d.layers) pend = ds;        // the next layer's norm launch adds the slabs
    else if (ds > 1) launch_reduce_rows(c, tgx::GEMM_RESIDUAL, ds, nullptr, c->ws_x, H, M, H, ssq);
    else if (!terms) hipLaunchKernelGGL(tgx::row_ssq_kernel, dim3(M, tgx::SK_NCB), dim3(256), 0, c->stream, (const float*)c->ws_x, (long long)H, H, ssq);
  }
  for (int b = 0; b < NB; b++)     // the last position of every batch row feeds lm_head
    (void)hipMemcpyAsync(c->rows[(size_t)(row0 + b)].x, c->ws_x + ((size_t)(b + 1) * S - 1) * H, (size_t)H * 4, hipMemcpyDeviceToDevice, c->stream);
}

// the skinny kernels' LDS images (weight tiles + activation panels / ring stages) exceed the 64 KB default
int skinny_set_attrs(tgx_ctx* c) {
  int rc;
  if ((rc = skinny_panel_set_attrs(c))) return rc;
  return skinny_dma_set_attrs(c);
}
