// ctx.h — the context of the MI355X shim (include/tgx.h) and the host-side launch interface shared by its translation units:
//   abi.hip          C ABI entry points, weight upload, step graphs, decode loop
//   decode.hip       batch-1..4 decode step: GEMV launches (kernels/gemv.h, oproj_sliced.h), lm_head, greedy finalize
//   attn.hip         decode attention launches (kernels/attn_decode.h, attn_decode_mfma.h)
//   sampler.hip      Sampler::sample (kernels/sampler.h)
//   prefill.hip      batched MFMA prefill, 16-bit storage (kernels/prefill.h, gemm_dma.h)
//   prefill_f32.hip  batched prefill, fp32 storage (kernels/gemm_f32.h)
//   skinny.hip       batched decode step / short prompts on the skinny MFMA GEMMs (kernels/skinny*.h)
// One context = one GPU = one HIP stream; every call comes from one host thread.  There is NO CPU path in this library.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tgx.h"
#include "kernels/common.h"

using tgx::bf16_t;
typedef unsigned char ebyte;   // parameter / KV-cache storage in the compute dtype: offsets are elements * ctx.esz

namespace tgx { struct SampScratch; }

constexpr int MAX_TICKET_EVENTS = 64;
constexpr int HOST_RING = 256;

struct LayerW {
  ebyte *in_norm = nullptr, *post_norm = nullptr;
  ebyte *wqkv = nullptr, *bqkv = nullptr, *wo = nullptr, *wgu = nullptr, *wdown = nullptr;
  ebyte *q_norm = nullptr, *k_norm = nullptr;   // Qwen3 [head_dim]
  bool q_norm_ok = false, k_norm_ok = false;
  // one bit per checkpoint tensor that lands in a merged weight: q, k, v, gate, up (bits 0-4) and the q/k/v biases (bits 5-7) — a tensor
  // uploaded twice must not stand in for a missing one of the same size
  int merged_filled = 0;
  bool in_norm_ok = false, post_norm_ok = false, wo_ok = false, wdown_ok = false;
  // GPT-2 (ModelGPT2.h:23-135): LayerNorm biases and a bias on every Conv1D; wgu holds c_fc [inter][hidden]
  ebyte *in_norm_b = nullptr, *post_norm_b = nullptr, *bo = nullptr, *bfc = nullptr, *bdown = nullptr;
  int gpt2_filled = 0;      // bit per GPT-2 tensor of the layer (12 of them)
};

struct RowState {       // independent KV/sequence state of one batch row
  float *x = nullptr, *q = nullptr, *attn = nullptr, *h = nullptr;   // fp32 activations
  float* k_raw = nullptr;   // Qwen3: un-normalised k of the current position
  float* logits = nullptr;
  float* probs = nullptr;   // final probabilities of the last sampled step [V]
  float* part_val = nullptr;
  int* part_idx = nullptr;
  float* attn_part = nullptr;
  int *tok = nullptr, *pos = nullptr;
  long long* prompt = nullptr;
  ebyte *kcache = nullptr, *vcache = nullptr;   // [layers][kv_heads][max_ctx][hd] in the compute dtype; paged KV: the pools [layers][blocks][kv_heads][KV_BLOCK][hd], shared by the rows
  const int* tbl = nullptr;                     // paged KV: this row's block table on the device
};

struct Tune {
  int ks = 1;    // waves sharing one unit's K range (1, 2, 4)
  int bpc = 4;   // grid cap in workgroups per CU
};

struct Profiler {
  bool on = false;
  hipEvent_t ev[2 * 8] = {};
  int64_t launches[TGX_KERNEL_COUNT] = {};
  double ms[TGX_KERNEL_COUNT] = {};
};

// where the QKV product of a one-sequence prefill may finish its rows (prefill.hip launch_prefill -> launch_gemm)
struct QkvEpi { bf16_t *q_hi = nullptr, *q_lo = nullptr, *k = nullptr, *v = nullptr; int past = 0; const int* tbl = nullptr; };     // past: the position of the pass's first row (RoPE, cache append)

struct tgx_ctx {
  tgx_model_desc d{};
  int device = 0;
  int num_cus = 256;
  hipStream_t stream = nullptr;
  std::string err;
  bool finalized = false;

  int dt = tgx::DT_BF16;   // storage dtype of parameters and KV cache (kernel template argument)
  size_t esz = 2;          // bytes per stored element
  ebyte *embed = nullptr, *lm_head = nullptr, *final_norm = nullptr;
  ebyte *wpe = nullptr, *final_norm_b = nullptr;      // GPT-2: learned positions [n_positions][H], ln_f.bias
  bool embed_ok = false, lm_head_ok = false, final_norm_ok = false, wpe_ok = false, final_norm_b_ok = false;
  bool gpt2 = false;
  std::vector<LayerW> L;
  float *rope_cos = nullptr, *rope_sin = nullptr;
  std::vector<RowState> rows;   // views into the per-row slabs below (constant row stride: batched GEMV walks them)
  float *slab_x = nullptr, *slab_q = nullptr, *slab_kraw = nullptr, *slab_attn = nullptr, *slab_h = nullptr, *slab_logits = nullptr;
  float *slab_probs = nullptr, *slab_part_val = nullptr, *slab_attn_part = nullptr;
  int *slab_part_idx = nullptr, *slab_tok = nullptr, *slab_pos = nullptr;
  long long* slab_prompt = nullptr;
  ebyte *slab_k = nullptr, *slab_v = nullptr;
  size_t kv_row_elems = 0, attn_part_row = 0;
  // prefill-by-steps processes up to 4 consecutive POSITIONS of one sequence per pass (rows of the batched kernels that share one
  // KV cache: kv_stride 0, pos[r] = past + r): fp32 storage, prompts shorter than 4 tokens, shapes the GEMM tile does not cover
  RowState chunk[4];
  float *ch_x = nullptr, *ch_q = nullptr, *ch_kraw = nullptr, *ch_attn = nullptr, *ch_h = nullptr, *ch_part = nullptr;
  int* ch_pos = nullptr;

  int64_t past = 0;       // host mirror of the device-resident pos: the LONGEST row of the batch (all rows, unless the per-row calls made them differ)
  std::vector<int64_t> row_past;   // host mirror of each row's own pos (tgx_reset_row / tgx_forward_row, include/tgx.h)
  std::vector<char> row_tok;       // the row has a current token (sampled after its last forward)
  std::vector<char> row_idle;      // the row was retired (tgx_reset_row) and not refilled: it rides in the steps, nothing waits for it, its output means nothing
  int batch = 0;          // rows used by the last forward
  bool have_logits = false, have_token = false;

  int* step = nullptr;    // device: number of decode steps finalized (monotonic)
  int* step_done = nullptr;   // device: rows of the current step that have read `step` (batches; 0 between steps)
  int* tok_log = nullptr; // device ring [log_cap][rows]
  int log_cap = 0;
  int* host_ring = nullptr;  // pinned host ring [HOST_RING][rows]
  int* host_ring_dev = nullptr;
  int64_t steps_issued = 0;
  hipEvent_t ticket_ev[MAX_TICKET_EVENTS] = {};
  int32_t last_sampled0 = -1;

  hipGraphExec_t step_graph = nullptr;      // the current entry of the cache below
  hipGraphExec_t multi_graph = nullptr;   // graph_steps consecutive decode steps (tgx_decode with many steps)
  // captured step graphs by (batch, sampler config, attention form): a generation that crosses an attention-form limit, or an engine that alternates between
  // sampler configurations / batch sizes, re-uses what it captured before instead of re-capturing (round 3; round 2 dropped the graphs at every change)
  struct GraphSet { hipGraphExec_t step = nullptr, multi = nullptr; int batch = 0; tgx_sampler_cfg cfg{}; bool direct = false, mfma = false, nw4 = false; unsigned long long used = 0; };
  GraphSet graph_cache[6];
  unsigned long long graph_clock = 0;
  int graph_cur = -1;
  bool mirror_to_host = true;             // finalize / pick kernels also store the token into the pinned host ring (tgx_fetch_token)
  int graph_steps = 16;                   // measured: 1 -> 1389 tok/s, 8 -> 1396, 16 -> 1399 (the gap between two graph launches is ~4 us)
  unsigned long long* seed_dev = nullptr;
  unsigned long long seed_on_dev = 0;         // value last copied to seed_dev: an unchanged seed costs no copy and no stream sync
  bool seed_valid = false;
  tgx::SampScratch* samp_scratch = nullptr;   // [max_batch] histograms / thresholds / partial sums of the staged sampler
  unsigned long long* samp_list_comp = nullptr;   // [max_batch][vocab] compacted threshold-bin entries of a filter (kernels/sampler.h): composite keys ...
  float* samp_list_v = nullptr;                   // ... and logit / T
  bool have_probs = false;
  std::vector<tgx_sampler_cfg> row_probs_cfg;   // per row: the sampler configuration of its last sampled step (tgx_read_probs evaluates the vector on demand) ...
  std::vector<char> row_probs_ok;               // ... and whether that step was a non-greedy one
  bool use_graph = true;

  Tune tune[TGX_KERNEL_COUNT];   // per kernel class: K-split and workgroups per CU
  int lm_grid = 0, attn_nsplit = 1, attn_nsplit_opt = 0;
  // batched-prefill workspace (grown on demand to the longest prompt seen)
  int ws_rows = 0;
  float *ws_x = nullptr, *ws_out = nullptr;           // [S][H] residual stream, [S][max(q+2kv, 2I)] GEMM output
  bf16_t *ws_ah = nullptr, *ws_al = nullptr;          // [S][max(H, qd, I)] GEMM A operand (hi, lo)
  bf16_t* ws_al2 = nullptr;                           // [S][H] third term for the QKV projection
  bf16_t *ws_qh = nullptr, *ws_ql = nullptr;          // [S][qd] rotated queries (hi, lo)
  bf16_t *ws_hh = nullptr, *ws_hl = nullptr;          // [S][I] siluMul output (hi, lo): the down product's A operand
  float* ws_part = nullptr; size_t ws_part_bytes = 0;   // split-K slabs of the short-prompt GEMMs
  bool poisoned = false;                                // a pass failed after some of its kernels were issued (device-side position / cache / counters may have moved): every entry point refuses until tgx_reset_cache
  const char* launch_fault = nullptr;                   // a launcher could not issue a kernel (a combination that is not instantiated): the issuing entry point fails with it
  int* ws_pos = nullptr;                                // fp32 prefill: [rows] positions of the prompt rows
  float* ws_ssq = nullptr;                              // [32][SK_NCB] partial sums of squares of the batched step's rows
  int gemm_splitk = 1;       // experiment: 0 disables split-K
  int wide_n_min = 2;        // ... from this many chips' worth of its workgroups (option prefill.wide_n_min)
  QkvEpi qkv_epi;
  int attn_dma = 1;          // option prefill.attn_dma: head_dim 64 prompts of three or more workgroups per CU take kernels/attn_prefill_dma.h (0 never, 2 always)
  int attn_ksplit = 1;       // option prefill.attn_ksplit: key split inside the prefill attention workgroup (0 never, 1 auto: head_dim 128, and head_dim 64 below three workgroups per CU, 2 always)
  int qk_fuse = 1;           // experiment: 0 keeps Qwen3's separate q/k norm launch
  // option prefill.skinny_rows: prompts of up to this many workspace rows take the skinny GEMMs (32: round 2; 33-64: four activation blocks, round 3).
  // Measured ms per prompt, four-block skinny / tiled split-K: Llama-3.2-1B S = 33 1.48 / 1.51, 48 1.51 / 1.58, 64 1.57 / 1.70; Qwen2.5-0.5B S = 48 1.46 / 1.81;
  // Llama-3.2-3B S = 48 3.54 / 3.33, Mistral-7B 6.76 / 5.52 — four blocks put 8 MFMAs + 9 LDS fragment reads behind every 32 k of a weight row: at
  // hidden > 2048 the tiled path's weight stream is faster (option prefill.skinny_hidden_max)
  // 65-128 rows (eight blocks, LDS-DMA ring kernel only), skinny / tiled: Llama-3.2-1B S = 65 1.65 / 1.89, 96 1.71 / 1.96, 128 1.82 / 2.05; Mistral-7B S = 96 8.01 / 7.56 -> hidden <= 2048
  // (option prefill.skinny_hidden_max_wide)
  int prefill_skinny_hidden_max_wide = 2048;
  int prefill_skinny_rows = 128;
  int prefill_skinny_hidden_max = 8192;   // (the 2048 limit of the panel-kernel form is gone with the LDS-DMA ring kernel: Llama-3.2-3B S = 48 3.12 -> 2.98 ms, Mistral-7B 5.36 / 5.38)
  int skinny_dma = 1;          // option skinny.dma: products on stored 16-bit terms (two terms) run on the LDS-DMA ring kernel (kernels/skinny_dma.h) from skinny.dma_rows rows
  int skinny_dma_rows = 1;
  int skinny_dma_oproj = 2;    // option skinny.dma_oproj: the matrix-core attention of a batched step writes 16-bit terms, the o_proj product takes that kernel
  // option skinny.dma_qkv: batches of up to 32 rows prepare the QKV / lm_head activations as stored terms as well (the 33-64-row form), so that the QKV product takes
  // that kernel: 1 = from 17 rows, 2 = from 5.  Llama-3.2-1B ms/step 1 / 2: B = 5 0.898 / 0.867, 8 0.905 / 0.880, 12 0.962 / 0.939, 16 1.022 / 1.001; context 2k B = 8
  // 1.071 / 1.050; Mistral-7B B = 8 3.587 / 3.556, B = 16 3.876 / 3.909
  int skinny_dma_qkv = 2;
  int skinny_dma_nbw = 0;      // option skinny.dma_nbw: weight blocks per wave of that kernel (0: as the panel kernel's geometry, 1 = 64-row, 2 = 128-row workgroups)
  int decode_step_rows = 128;   // option decode.step_rows: rows of a batch that share one pass over the weights in the matrix-core step (32: round 2; 128: eight blocks on the LDS-DMA ring kernel — Llama-3.2-1B B = 128 2.99 -> 2.32 ms/step, Mistral-7B 13.25 -> 10.47)
  int prefill_skinny = 1;    // option prefill.skinny: 0 sends prompts of <= 32 rows through the tiled GEMMs as well
  int skinny_wgs = 256;      // option skinny.wgs: workgroups a skinny product aims for by splitting K
  int skinny_gu_split = 0;   // option skinny.gu_split: 0 keeps the gate_up product unsplit (siluMul in its epilogue, one launch less)
  int skinny_cfg_mid = 0;    // option skinny.cfg_mid: tile geometry (kernels/skinny.h SkinnyCfg) of the products that do not oversubscribe the chip
  int skinny_cfg_force = -1; // option skinny.cfg: force one geometry for every product (experiments)
  // decode batches of at least this many rows run their Linears as skinny MFMA GEMMs (option decode.mfma_min_batch).  Measured ms/step,
  // GEMV row groups vs matrix cores: Llama-3.2-1B B = 2 0.794 / 0.991, B = 3 ~1.45 / 1.003, B = 4 1.042 / 1.007; Mistral-7B B = 4 5.04 / 4.01
  int decode_mfma_min = 3;
  bool prefill_mfma = true;
  int prefill_min_rows = 4;  // prompts shorter than this go through the decode kernels, 4 positions per pass (set in tgx_create)
  int prefill_f32_min_rows = 16;   // fp32 storage: prompts from this length on take the f32-input MFMA GEMMs (64-row tiles; option prefill.f32_min_rows)
  int gemm_tm = 0;           // experiment: force the GEMM row tile (64 / 128); 0 = by the number of tiles
  // option prefill.gemm_dma: bit 0 / 1 = unsplit prefill GEMMs take their tiles by LDS-DMA (kernels/gemm_dma.h), bit 2 = the wide product (gate_up / c_fc)
  // on the 8-wave 256 x 256 three-stage kernel; bits 4-7 / 8-11 = ring geometry of the 128-row / 64-row tiles (k per stage, stages).  0 = the register-staged
  // gemm_x2_kernel everywhere (round 1).  Default 7 | k32x2 << 4 | k64x2 << 8: Llama-3.2-1B 2048 tokens 11.4-11.7 -> 10.0-10.3 ms (tools/dma_sweep.py)
  // bit 3 = the N = hidden products (o_proj, down) on the 8-wave 128 x 128 kernel with the K step split between wave pairs when their tiles number ~one per CU
  int gemm_dma = 15 | (1 << 4) | (2 << 8);
  int wide_8k_max = 8;       // option prefill.wide_8k_max: ... while its tiles number at most this many half-chips (8 = 4 tiles per CU: everything below the 256 x 256 kernel's range;
                             // 3 / 8: Llama-3.2-1B S = 512 3.62 / 3.44 ms, 768 5.06 / 4.94; Mistral-7B S = 256 11.16 / 10.81, 512 22.5 / 21.4)
  int debug_attn = 0;        // experiment: AttnArgs.dbg
  int attn_gmax = 0;         // experiment: query heads per attention workgroup (default 2)
  int attn_direct_nw4 = 0;   // option attn.direct_nw4: contexts up to this many keys run the direct attention form with four waves per head (set in tgx_create)
  bool attn_nw4 = false;     // mode of the launches being issued / captured
  int attn_raw_fuse = 2;     // option attn.raw_fuse: that form also finishes the QKV product (slab sums, bias, q / k norm, RoPE, cache append) in its prologue
  int attn_batch_nw8 = 1;    // option attn.batch_nw8: eight waves per workgroup of that form while its workgroups number at most one per CU (Llama-3.2-1B B = 17 1.052 -> 1.030 ms/step, 32 1.189 -> 1.171; context 2k B = 17 1.239 -> 1.183; 2 = always: B = 64 1.518 -> 1.582)
  int attn_batch_la = 0;     // option attn.batch_la: K / V look-ahead registers of that form at head_dim 64 (-1: only while its workgroups number at most one per CU)
  int attn_batch_mfma = 17;  // option attn.batch_mfma: batches of this many rows and more run their direct-form attention on the matrix cores (0 = never)
  int attn_direct_g = 1;     // option attn.direct_g: 1 = heads per workgroup of the direct attention form by batch rows (2 from 12 rows, 4 from 24 at head_dim 64), 0 = always one, -g = force g
  int attn_direct_max = 384; // contexts up to this many keys take the one-workgroup-per-head attention (no split, no combine launch); set in tgx_create
  bool attn_direct = false;  // mode of the launches being issued / captured
  // contexts from attn_mfma_min keys on take the MFMA decode attention (kernels/attn_decode_mfma.h); like the direct form it is a mode of the
  // captured step: the graphs are re-captured when a decode call crosses the limit.  Not for Qwen3's fused q/k norm, not for fp32 storage.
  int skinny_terms = 1;            // option skinny.terms: batches of 17-32 rows take gate_up's activations as terms prepared once per layer (round 3)
  int skinny_ksplit = 1;           // wide products (gate_up, lm_head) of the batched step on the barrier-free K-split kernel (option skinny.ksplit)
  int defer_min_rows = 129;        // option prefill.defer_min_rows (192 until the row-wise norm launch loaded its slabs eight at a time: Llama-3.2-1B S = 160 2.51 -> 2.41 ms, 191 2.54 -> 2.46; Mistral-7B S = 160 9.90 -> 9.65)
  int defer_reduce = 1;            // split-K slabs of the prefill's N = hidden / QKV products are summed by the next row-wise kernel (option prefill.defer_reduce)
  int attn_mfma_min = -1;          // -1: the measured crossover of the geometry (attn_mfma_threshold); option attn.mfma_min overrides
  bool attn_mfma = false;
  int debug_gemv = 0;        // experiment: GemvArgs.dbg = value & 15 for the kernel classes selected by bits 8.. (1 << (8 + class))
  int debug_skip = 0;        // experiment: bit0 skip attn decode kernel, bit1 skip combine (results invalid)
  int prof_same_layer = 0;   // experiment: tgx_profile_decode replays ONE layer's weights (Infinity-Cache resident)
  // option oproj.sliced (default 1): batch-1 decode steps on the split attention form run o_proj K-sliced with the merge of the attention splits in its
  // prologue (kernels/oproj_sliced.h): no attn_combine launch; the residual stream between o_proj and down lives in fixed-point accumulators
  int oproj_sliced = 1;
  // option act.round16 (default 0): the input of every Linear is rounded to the storage dtype — the part of the reference's 16-bit-module contract
  // (ModelLlama.h:62) that has a price: the matrix-core products then take ONE 16-bit term per activation.  ws_zero: the "lo term"
  // every stored-term product reads in this mode
  int act16 = 0, act16_kernels = 1;
  int prefill_terms_rows = 16;   // option prefill.terms_rows: skinny prompts with more rows than this take their norm-fused products on stored 16-bit terms (round 4: from 17 rows instead of 33 — Mistral-7B S = 20 / 32 4.47 / 4.67 -> 3.94 / 4.30 ms, Llama-3.2-1B within noise)
  int wide_8k_eff = 76;      // (Llama-3.2-1B S = 1152 7.28 -> 6.78 ms, 1280 7.52 -> 6.95, 1400 8.45 -> 8.14; at 87 % fill the 256 x 256 tiles win again) option prefill.wide_8k_eff (per cent): gate_up on 128 x 128 tiles when the 256 x 256 tiling's rounds are filled less than this
  int skinny_terms_above = 2;  // (round 4: 2 instead of 4 — Llama-3.2-1B B = 3 / 4 0.822 / 0.823 -> 0.794 / 0.798 ms per step) option skinny.terms_above: batched steps with more rows than this prepare stored terms for the norm-fused products (with skinny.dma_qkv 2)
  int qkv_nosplit = 1;       // option prefill.qkv_nosplit: the balanced QKV launch instead of K slabs when it alone covers 3/4 of the chip
  int splitk_8k = 1;         // option prefill.splitk_8k: N = hidden products of 129-1500-row prompts as 2-4 K slabs on the eight-wave LDS-DMA kernel
  bf16_t* ws_zero = nullptr; size_t ws_zero_elems = 0;
  // batch-1 steps on the direct attention form (short contexts, head_dim 64): the o_proj product runs in the attention launch's epilogue (attn_decode_kernel
  // template OPJ) — 4 launches per layer; the direct form then serves contexts up to attn_fused_max keys, with four wave-loads per softmax block up to attn_fused_nw4 keys and eight beyond (four waves per head either way); was: four / eight waves up to attn_fused_nw4
  int oproj_fused = 1, attn_fused_max = 640, attn_fused_nw4 = 384;
  // ---- paged KV (round 6; option kv.budget_tokens before tgx_finalize; include/tgx.h).  The caches become pools of KV_BLOCK-token blocks shared by the rows; a row's
  // blocks are assigned on the host as its sequence grows (before the launch that writes them) and returned by tgx_reset_row / tgx_reset_cache.  Block 0 is scratch.
  int kv_budget_tokens = 0;        // > 0: paged
  bool kv_paged = false;
  int kv_nblocks = 0;              // physical blocks incl. the scratch block
  int kv_tbl_stride = 0;           // table entries per row = ceil(max_ctx / KV_BLOCK)
  int* kv_tbl = nullptr;           // device [max_batch][kv_tbl_stride]
  std::vector<int> kv_tbl_host;    // its host mirror
  std::vector<int> kv_free;        // free physical blocks
  std::vector<int> kv_row_nblk;    // blocks assigned to each row
  long long* slab_acc = nullptr;   // [max_batch][hidden], resting at zero between layers
  float* scratch_x = nullptr;   // [hidden] residual sink for tgx_profile_decode
  Profiler prof;
};

// ---- errors: every launcher returns void / a count; a kernel it could not issue is recorded in launch_fault and turned into a status by the entry point
int set_err(tgx_ctx* c, int code, const char* fmt, ...);

#define HIP_OK(c, call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return set_err((c), TGX_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define LAUNCH_OK(c)                                                                             \
  do {                                                                                           \
    if ((c)->launch_fault) { const char* f_ = (c)->launch_fault; (c)->launch_fault = nullptr; return set_err((c), TGX_ERR_UNSUPPORTED, "%s", f_); } \
  } while (0)

// Runs `body` with DT bound to the context's storage dtype as a compile-time constant (kernel template argument).
#define TGX_DT_SWITCH(dt_, ...)                                                          \
  switch (dt_) {                                                                         \
    case tgx::DT_BF16: { constexpr int DT = tgx::DT_BF16; __VA_ARGS__; } break;          \
    case tgx::DT_F16: { constexpr int DT = tgx::DT_F16; __VA_ARGS__; } break;            \
    default: { constexpr int DT = tgx::DT_F32; __VA_ARGS__; } break;                     \
  }

// The MFMA prefill kernels exist for the two 16-bit storage dtypes.
#define TGX_DT16_SWITCH(dt_, ...)                                                        \
  if ((dt_) == tgx::DT_F16) { constexpr int DT = tgx::DT_F16; __VA_ARGS__; }             \
  else { constexpr int DT = tgx::DT_BF16; __VA_ARGS__; }

template <typename T>
int dev_alloc(tgx_ctx* c, T** p, size_t n) {
  HIP_OK(c, hipMalloc((void**)p, n * sizeof(T)));
  return TGX_OK;
}

namespace tgx { struct AttnArgs; struct FinalizeArgs; struct AttnPrefillArgs; struct RopeKvArgs; }

// ---- abi.hip
void drop_step_graphs(tgx_ctx* c);
int kv_ensure_blocks(tgx_ctx* c, int row, long long tokens);       // paged KV: row `row` may hold `tokens` tokens after this (assigns blocks, updates the device table, stream-ordered)
bool is_greedy(const tgx_sampler_cfg* s);   // Sampler.cpp:15-21
// ---- decode.hip (kernels/gemv.h, kernels/oproj_sliced.h)
int gemv_grid(const tgx_ctx* c, int units, int ks, int bpc);
bool oproj_sliced_ok(const tgx_ctx* c, int R, long long kv_stride);
bool oproj_fused_capable(const tgx_ctx* c);                               // static conditions (geometry, dtype, option)
bool oproj_fused_ok(const tgx_ctx* c, int R, long long kv_stride);       // ... and this launch is a batch-1 step on the direct form
int launch_layer_kernel(tgx_ctx* c, RowState* rv, int R, int l, int cls, float* resid, long long kv_stride);   // returns the number of launches issued
void launch_layers(tgx_ctx* c, RowState* rv, int R, long long kv_stride);
void launch_layers(tgx_ctx* c, int row0, int R);
void launch_lm_head(tgx_ctx* c, int row0, int R);
tgx::FinalizeArgs make_finalize_args(tgx_ctx* c, int row, bool advance_pos, bool log_step);
void launch_finalize_greedy(tgx_ctx* c, int row0, int R, bool advance_pos, bool log_step);   // one finalize_greedy launch per row
void launch_finalize_rows(tgx_ctx* c, int row0, int M);                                     // the batched step's greedy finalize (+ step counter)
void launch_embed_chunk(tgx_ctx* c, const long long* ids, int R, int pos0);
void launch_add_pos(tgx_ctx* c, int* pos, int n);
void launch_argmax_partials(tgx_ctx* c, const float* logits, int V, float* part_val, int* part_idx);
// ---- attn.hip (kernels/attn_decode.h, attn_decode_mfma.h)
bool attn_batch_on_mfma(const tgx_ctx* c, int R);
void launch_attn(tgx_ctx* c, const tgx::AttnArgs& a, int R, bool combine = true);   // combine = false: the caller's o_proj merges the split records
int attn_set_attrs(tgx_ctx* c);
// ---- sampler.hip (kernels/sampler.h)
void launch_sample(tgx_ctx* c, int row0, int R, const tgx_sampler_cfg& cfg, bool advance_pos, bool log_step);
void launch_probs(tgx_ctx* c, int row, const tgx_sampler_cfg& cfg);
int sampler_alloc(tgx_ctx* c);
// ---- prefill.hip (kernels/prefill.h, gemm_dma.h)
bool prefill_shapes_ok(const tgx_model_desc& d);
int ensure_prefill_ws(tgx_ctx* c, int S);
void launch_prefill(tgx_ctx* c, int row0, int NB, int S, int past);      // past: the rows' common pastLength (their positions are past .. past + S - 1) — explicit, not read from the context
int prefill_set_attrs(tgx_ctx* c);
void launch_attn_prefill(tgx_ctx* c, const tgx::AttnPrefillArgs& a, bool allow_lean);
void launch_rope_kv_split(tgx_ctx* c, const tgx::RopeKvArgs& a, int S);
void launch_norm_terms(tgx_ctx* c, float* x, const ebyte* norm_w, int M, int H, int nsplit, bool third = false);
void launch_silu_slab_reduce(tgx_ctx* c, int M, int I, int nsplit);
void launch_embed_rows(tgx_ctx* c, const long long* ids, float* X, int M, int S);
// ---- prefill_f32.hip (kernels/gemm_f32.h)
int ensure_f32_part(tgx_ctx* c, int rows);
void launch_prefill_f32(tgx_ctx* c, int row0, int NB, int S);
// ---- skinny.hip (kernels/skinny.h, skinny_ksplit.h, skinny_dma.h)
bool decode_mfma_ok(const tgx_ctx* c);
int ensure_skinny_ws(tgx_ctx* c, int rows);
void launch_decode_step_mfma(tgx_ctx* c, int row0, int M, const tgx_sampler_cfg& cfg);
void launch_prefill_skinny(tgx_ctx* c, int row0, int NB, int S);
int skinny_set_attrs(tgx_ctx* c);
