// tgx_mi355x.hip — the extern "C" shim of include/tgx.h for MI355X (gfx950): context, weight upload by HF
// name, KV cache, decode-step hipGraph, and the launch sequence of the hand-written kernels in kernels/*.h.
//
// One context = one GPU = one HIP stream; every call comes from one host thread (the reference's engine is
// entered by one thread only: examples/inference/main.cpp, server/HttpServer.cpp:118-163).
// There is NO CPU path in this library: every entry point either runs on the GPU or returns an error.
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
#include "kernels/attn_decode.h"
#include "kernels/attn_decode_mfma.h"
#include "kernels/common.h"
#include "kernels/gemv.h"
#include "kernels/prefill.h"
#include "kernels/sampler.h"
#include "kernels/skinny.h"
#include "kernels/skinny_ksplit.h"
#include "kernels/gemm_f32.h"
#include "kernels/gemm_dma.h"
#include "kernels/skinny_dma.h"
#include "kernels/engine.h"

using tgx::bf16_t;
typedef unsigned char ebyte;   // parameter / KV-cache storage in the compute dtype: offsets are elements * ctx.esz

namespace {

constexpr int MAX_TICKET_EVENTS = 64;
constexpr int F32_ATTN_ROWS = 64;      // prompt rows per attention launch of the fp32 prefill (bounds the split-partials workspace)
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
  ebyte *kcache = nullptr, *vcache = nullptr;   // [layers][kv_heads][max_ctx][hd] in the compute dtype
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

}  // namespace

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

  int64_t past = 0;       // host mirror of every row's device-resident pos
  int batch = 0;          // rows used by the last forward
  bool have_logits = false, have_token = false;

  int* step = nullptr;    // device: number of decode steps finalized (monotonic)
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
  int graph_steps = 8;                    // measured: 1 -> 1389 tok/s, 8 -> 1396, 16 -> 1399 (the gap between two graph launches is ~4 us)
  unsigned long long* seed_dev = nullptr;
  unsigned long long seed_on_dev = 0;         // value last copied to seed_dev: an unchanged seed costs no copy and no stream sync
  bool seed_valid = false;
  tgx::SampScratch* samp_scratch = nullptr;   // [max_batch] histograms / thresholds / partial sums of the staged sampler
  bool have_probs = false;
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
  const char* launch_fault = nullptr;                   // a launcher could not issue a kernel (a combination that is not instantiated): the issuing entry point fails with it
  int* ws_pos = nullptr;                                // fp32 prefill: [rows] positions of the prompt rows
  float* ws_attn_part = nullptr;                        // fp32 prefill: split-attention partials of one block of rows
  float* ws_ssq = nullptr;                              // [32][SK_NCB] partial sums of squares of the batched step's rows
  int gemm_splitk = 1;       // experiment: 0 disables split-K
  int splitk_dma = 1;        // option prefill.splitk_dma: the split-K slabs of a short prompt through the LDS-DMA GEMM (round 3)
  int qkv_balanced = 1;      // option prefill.qkv_balanced: the bf16 QKV product as one launch of equal-work tiles (round 3)
  int attn_mirror = 1;       // experiment: prefill attention block order
  int qk_fuse = 1;           // experiment: 0 keeps Qwen3's separate q/k norm launch
  // Two launch folds that are built, parity-tested and OFF by default because they measured slower on MI355X (profiles/r02_launch_folds.txt):
  // attn.fold_combine = 1 merges the attention splits in the o_proj launch's prologue (98 -> 82 launches per token on Llama-3.2-1B): every one of
  // the 256 o_proj workgroups then re-reads the same ~150 KB of partial records through L2, which costs 3.3 us against the 2.9 us launch it removes
  // (context 2.4k: 0.7109 -> 0.7173 ms/token; +1.7 % only on Qwen2.5-0.5B's 14 heads; -20 % at head_dim 128 / context 6k);
  // lmhead.fuse_finalize = 1 lets the lm_head launch's last-arriving workgroup do the greedy finalize (arrival ticket): the drain + atomic on
  // all ~1000 workgroups costs what the 1-workgroup finalize launch cost (0.7109 -> 0.7123 ms/token).
  int attn_fold = 0;
  int lm_fuse = 0;
  // attn.fold_ticket = 1 (round 3): the last-arriving split workgroup of each (row, kv head, head group) merges the group's records inside the
  // attention launch (per-group arrival ticket) — no attn_combine launch.  Off by default: measured in profiles/r03_attn_fold.txt
  int attn_ticket = 0;
  unsigned* attn_tickets = nullptr;   // [max_batch][kv_heads][8] counters resting at 0
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
  unsigned int* lm_ticket = nullptr;   // arrival counter of the lm_head launch (rests at 0)
  tgx::FinalizeArgs* fin_dev = nullptr;   // [4] finalize arguments of the fused lm_head launch (option lmhead.fuse_finalize)
  bool prefill_mfma = true;
  int prefill_min_rows = 4;  // prompts shorter than this go through the decode kernels, 4 positions per pass (set in tgx_create)
  int f32_flash = 1;               // option prefill.f32_flash: 0 = attention of the fp32 prefill through the decode attention kernel
  int prefill_f32_min_rows = 16;   // fp32 storage: prompts from this length on take the f32-input MFMA GEMMs (64-row tiles; option prefill.f32_min_rows)
  int gemm_tm = 0;           // experiment: force the GEMM row tile (64 / 128); 0 = by the number of tiles
  // option prefill.gemm_dma: bit 0 / 1 = unsplit prefill GEMMs take their tiles by LDS-DMA (kernels/gemm_dma.h), bit 2 = the wide product (gate_up / c_fc)
  // on the 8-wave 256 x 256 three-stage kernel; bits 4-7 / 8-11 = ring geometry of the 128-row / 64-row tiles (k per stage, stages).  0 = the register-staged
  // gemm_x2_kernel everywhere (round 1).  Default 7 | k32x2 << 4 | k64x2 << 8: Llama-3.2-1B 2048 tokens 11.4-11.7 -> 10.0-10.3 ms (tools/dma_sweep.py)
  // bit 3 = the N = hidden products (o_proj, down) on the 8-wave 128 x 128 kernel with the K step split between wave pairs when their tiles number ~one per CU
  int gemm_dma = 15 | (1 << 4) | (2 << 8);
  int wide_8k_max = 8;       // option prefill.wide_8k_max: ... while its tiles number at most this many half-chips (8 = 4 tiles per CU: everything below the 256 x 256 kernel's range;
                             // 3 / 8: Llama-3.2-1B S = 512 3.62 / 3.44 ms, 768 5.06 / 4.94; Mistral-7B S = 256 11.16 / 10.81, 512 22.5 / 21.4)
  int wide_8k = 1;           // option prefill.wide_8k: gate_up of 129-384-row prompts on the eight-wave 128 x 128 kernel
  int hidden_256 = 1;        // option prefill.hidden_256: o_proj / down on the 256 x 256 eight-wave kernel when their tiles fill the chip
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
  int debug_nops = 0;     // extra no-op launches per layer (launch-overhead experiments only)
  // L2 prefetch chaining (kernels/l2_prefetch.h): batch-1 decode launches carry prefetch workgroups that pull the next launches' weights into
  // the L2 of the XCD that will read them.  Option pf.mode: 0 off, 1 on.  Budgets are KB per XCD (L2 = 4 MB per XCD).
  int pf_mode = 0;
  int pf_wgs = 64;            // prefetch workgroups appended to a launch (256 threads each)
  int pf_stride = 128;        // bytes between touches
  int pf_oproj_kb = 4096;     // qkv(l)     -> o_proj(l)      (whole matrix: 1 MB per XCD on Llama-3.2-1B)
  int pf_gu_kb = 2048;        // attn(l)    -> gate_up(l) head
  int pf_dn_kb = 0;           // gate_up(l) -> down(l) head
  int pf_qkv_kb = 4096;       // down(l)    -> qkv(l + 1)     (whole matrix: 1.6 MB per XCD)
  int pf_comb_kb = 0;         // combine(l) -> gate_up(l) head (the combine launch carries the prefetch workgroups)
  int pf_oproj_gu_kb = 0;     // o_proj(l)  -> gate_up(l), behind what attention / combine covered
  // The persistent weight-streaming engine (kernels/engine.h), batch-1 decode steps of the RMSNorm families in 16-bit storage.  Option
  // engine.mode: 0 = off (default: the GEMV launches measured faster on MI355X, profiles/r03_engine.txt), 1 = gate_up + down in one launch,
  // 2 = o_proj + gate_up + down + the next layer's qkv in one launch (3 launches per layer instead of 6).
  int engine_mode = 0;
  int engine_ns = 0;         // ring slots (0: as many as LDS holds)
  int engine_depth = 3;      // fills in flight per CU
  int engine_thin = 0;
  int engine_stats = 0;      // 1: the STATS instantiation records its timeline per layer (tgx_engine_read_stats)
  tgx::u64 *eng_gx1 = nullptr, *eng_gh = nullptr, *eng_gx2 = nullptr;   // granule buffers of the three in-launch edges
  unsigned *eng_epoch = nullptr, *eng_err = nullptr;
  unsigned long long* eng_stats = nullptr;   // [layers][num_cus][ENG_NSTAT]
  int* nop_word = nullptr;
  float* scratch_x = nullptr;   // [hidden] residual sink for tgx_profile_decode
  Profiler prof;
};

namespace {

std::string g_create_err;

int set_err(tgx_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_err = buf;
  return code;
}

#define HIP_OK(c, call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return set_err((c), TGX_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// launchers return void / slab counts: a kernel they could not issue is recorded in launch_fault and turned into a status by the entry point
#define LAUNCH_OK(c)                                                                             \
  do {                                                                                           \
    if ((c)->launch_fault) { const char* f_ = (c)->launch_fault; (c)->launch_fault = nullptr; return set_err((c), TGX_ERR_UNSUPPORTED, "%s", f_); } \
  } while (0)

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

// ------------------------------------------------------------------------------------------------
// kernel launch helpers
// ------------------------------------------------------------------------------------------------
int gemv_grid(const tgx_ctx* c, int units, int ks, int bpc) {
  const int upb = 4 / ks;
  const int want = (units + upb - 1) / upb;
  const int cap = c->num_cus * bpc;
  return want < cap ? want : cap;
}

// 16-byte slices per lane per row for a K range split over ks waves (the kernel's NX template parameter)
int gemv_nx(int K, int ks) { return ((K / 8) + ks * 64 - 1) / (ks * 64); }

// smallest K split that keeps a wave's slice within 8 x 512 elements; norm-fused launches must use 1
int gemv_auto_ks(int K, int want) {
  int ks = want;
  while (ks < 4 && gemv_nx(K, ks) > 8) ks *= 2;
  return ks;
}

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

// the argument block of batch row `r` alone (every slab pointer advanced by r row strides)
tgx::GemvArgs gemv_row(const tgx_ctx* c, tgx::GemvArgs a, int r) {
  a.x += (size_t)r * a.x_stride;
  if (a.out) a.out += (size_t)r * a.out_stride;
  if (a.q_out) a.q_out += (size_t)r * a.q_stride;
  if (a.k_raw) a.k_raw += (size_t)r * a.kraw_stride;
  if (a.k_cache) a.k_cache = (ebyte*)a.k_cache + (size_t)r * a.kv_stride * c->esz;
  if (a.v_cache) a.v_cache = (ebyte*)a.v_cache + (size_t)r * a.kv_stride * c->esz;
  if (a.pos) a.pos += r;
  if (a.logits) a.logits += (size_t)r * a.logits_stride;
  if (a.part_val) a.part_val += (size_t)r * a.part_stride;
  if (a.part_idx) a.part_idx += (size_t)r * a.part_stride;
  if (a.attn_part) a.attn_part += (size_t)r * a.part_in_stride;
  if (a.ticket && r) a.fin += r;
  return a;
}

template <int DT, int PRO, int EPI, int NX>
void launch_gemv_nx(tgx_ctx* c, const tgx::GemvArgs& a, int grid, int R) {
  const dim3 g(grid), b(256);
  // PRO_ATTNCOMB stages the merged attention output of its rows in LDS: [rows][K] fp32 (attn_fold_ok() keeps it within 64 KB)
  const size_t smem_row = PRO == tgx::PRO_ATTNCOMB ? (size_t)a.K * 4 : 0;
  // rows share the weight pass; R x NX activation slices of 8 floats stay in registers (4 x 8 x 8 = 256 of the 512 a wave
  // of a 256-thread workgroup may use)
  if (R == 4) { hipLaunchKernelGGL((tgx::gemv_kernel<DT, PRO, EPI, NX, 4>), g, b, 4 * smem_row, c->stream, a); return; }
  if (R == 2) { hipLaunchKernelGGL((tgx::gemv_kernel<DT, PRO, EPI, NX, 2>), g, b, 2 * smem_row, c->stream, a); return; }
  if constexpr (PRO != tgx::PRO_ATTNCOMB && PRO != tgx::PRO_LAYERNORM && EPI != tgx::EPI_LOGITS && EPI != tgx::EPI_GELU && DT != tgx::DT_F32) {
    if (R == 1 && a.pf.n_compute) { hipLaunchKernelGGL((tgx::gemv_kernel<DT, PRO, EPI, NX, 1, true>), g, b, smem_row, c->stream, a); return; }   // with prefetch workgroups
  }
  for (int r = 0; r < R; r++) hipLaunchKernelGGL((tgx::gemv_kernel<DT, PRO, EPI, NX, 1>), g, b, smem_row, c->stream, gemv_row(c, a, r));
}

template <int PRO, int EPI>
void launch_gemv(tgx_ctx* c, tgx::GemvArgs a, int cls, int R) {
  const Tune& tn = c->tune[cls];
  a.dbg = ((c->debug_gemv >> (8 + cls)) & 1) ? (c->debug_gemv & 15) : 0;
  if (!a.ldw) a.ldw = a.K;
  a.ks = gemv_auto_ks(a.K, tn.ks);   // norm-fused launches K-split too (their waves exchange the sums of squares through LDS)
  while (R > 1 && a.ks < 4 && gemv_nx(a.K, a.ks) > 4) a.ks *= 2;   // batch rows: at most 4 slices per row and lane (measured: B = 2 and 4 on the 1B / 3B / 7B shapes)
  // two rows on the gate_up launch: 2 slices per row and lane leave room for the double-buffered weight registers (R x NX <= 4):
  // Llama-3.2-1B B = 2: 2394 -> 2506 tok/s; the same split loses 2-6 % at B = 4 and on the other launches (tools/batch_bench.py --opts)
  if (R == 2 && cls == TGX_KERNEL_GATEUP && a.ks == 1 && gemv_nx(a.K, 1) == 4) a.ks = 2;
  int grid = (EPI == tgx::EPI_LOGITS) ? c->lm_grid : gemv_grid(c, a.units, a.ks, tn.bpc);
  a.pf.n_compute = 0;
  if (R == 1 && (a.pf.t[0].W || a.pf.t[1].W) && PRO != tgx::PRO_ATTNCOMB && PRO != tgx::PRO_LAYERNORM && EPI != tgx::EPI_LOGITS && EPI != tgx::EPI_GELU && c->dt != tgx::DT_F32) {
    a.pf.n_compute = grid; a.pf.stride = c->pf_stride; a.pf.sink = reinterpret_cast<unsigned*>(c->nop_word); grid += c->pf_wgs;
  }
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

// Split-form attention leaves per-split partial records; the o_proj GEMV merges them in its prologue (one launch per layer less) when the
// merged rows fit its LDS stage: R rows x heads*head_dim fp32 <= 64 KB (every BASELINE geometry at R <= 4).  Option attn.fold_combine = 0
// keeps the separate attn_combine_kernel launch.
bool attn_fold_ok(const tgx_ctx* c, int R) {
  return c->attn_fold && c->engine_mode < 2 && !c->attn_direct && (size_t)R * c->d.heads * c->d.head_dim * 4 <= 65536;
}

// the direct-form attention of a batched step runs on the matrix cores from attn.batch_mfma rows (17) when a kv head serves 3+ query heads (the VALU form's
// cost grows with the heads per workgroup, the MFMA form's does not: Qwen3-1.7B, 2 heads per kv head, B = 32 2.29 (VALU) vs 2.39 ms/step)
bool attn_batch_on_mfma(const tgx_ctx* c, int R) {
  return c->attn_batch_mfma > 0 && R >= c->attn_batch_mfma && (c->d.heads / c->d.kv_heads >= 3 || c->attn_batch_mfma == 1);
}

template <int DT, int HD, bool QKN = false>
void launch_attn_g(tgx_ctx* c, tgx::AttnArgs a, int R) {
  // the query heads of a kv group go to workgroups two at a time (blockIdx.z): the per-head state (8 output registers, the
  // merges) is what a workgroup's time grows with, while the K/V tile the groups re-read is small and mostly L2-resident.
  // Measured (option attn.gmax; tok/s at 4 / 2 / 1 heads per workgroup): Llama-3.2-1B ctx 2.3k 1364 / 1391 / 1388, ctx 8k
  // 1282 / 1312 / 1295; Qwen2.5-0.5B (7 heads per kv head) 1512 / 1610 / 1621; Mistral-7B 337 / 340 / 339
  const int gmax = c->attn_gmax > 0 ? c->attn_gmax : 2;
  const int gfull = a.heads / a.kv_heads, ngroups = gfull > gmax ? (gfull + gmax - 1) / gmax : 1, G = (gfull + ngroups - 1) / ngroups;
  a.gfull = gfull;
  a.direct = c->attn_direct ? 1 : 0;
  if (a.direct || (c->attn_mfma && !QKN && DT != tgx::DT_F32)) a.pf.t[0].W = nullptr;   // prefetch workgroups ride in the VALU split form only
  a.pf.n_compute = 0;
  auto launch_combine = [&]() {      // the merge of the split records; with option pf.comb_kb it carries prefetch workgroups (its own traffic is a few KB)
    if ((c->debug_skip & 2) || attn_fold_ok(c, R)) return;
    if (a.pf_comb.t[0].W && R == 1) {
      tgx::AttnArgs b = a;
      b.pf = a.pf_comb; b.pf.n_compute = a.heads; b.pf.stride = c->pf_stride; b.pf.sink = reinterpret_cast<unsigned*>(c->nop_word);
      const int gx = (a.heads + c->pf_wgs + 7) / 8 * 8;
      hipLaunchKernelGGL((tgx::attn_combine_kernel<HD, true>), dim3(gx, R), dim3(256), 0, c->stream, b);
      return;
    }
    hipLaunchKernelGGL((tgx::attn_combine_kernel<HD>), dim3(a.heads, R), dim3(256), 0, c->stream, a);
  };
  if (a.direct) {   // short context: one 16-wave workgroup per query head, no combine launch.  Measured (tok/s, direct vs split at context
    // ~120 / ~300 / ~430): see DESIGN.md §5; 1 head per workgroup beats 2 and 4 here (the K/V block is L2-resident, the softmax chain is not)
    // Batches (round 3): with R rows the K/V working set (R x kv_heads x T rows) no longer fits the L2s, and one workgroup per QUERY head reads each kv
    // head gfull times — Llama-3.2-1B B = 32 at context ~600: 24 us per layer, a quarter of the step.  From `attn.direct_rows` rows on, a workgroup takes
    // two query heads of a kv head (option attn.direct_g: 1, 2 or 4 heads).
    // measured ms/step by heads per workgroup (1 / 2 / 4): Llama-3.2-1B context 600 B = 16 1.105 / 1.058 / 1.132, B = 32 1.551 / 1.421 / 1.382; context 2k
    // B = 16 1.354 / 1.208 / 1.384, B = 32 2.458 / 1.906 / 1.680; Mistral-7B context 600 B = 16 4.11 / 3.94 / 4.35, B = 32 6.08 / 5.47 / 5.53
    // Batches on the matrix cores (round 3, option attn.batch_mfma = rows from which): the VALU form's arithmetic grows with the heads per workgroup
    // (softmax chain + P.V update per key and head: B = 32 at context ~600 is VALU-bound at 16 us per layer for 39 MB of K / V), the MFMA form's does
    // not — one workgroup per (row, kv head), all the group's query heads as the narrow operand, no split, no combine launch.  Measured ms/step (VALU /
    // MFMA, 4 waves; 8 waves the same within 0.5 %): Llama-3.2-1B context 600 B = 16 1.047 / 1.068, B = 24 1.308 / 1.259, B = 32 1.356 / 1.307; context 2k
    // B = 16 1.210 / 1.224, B = 24 1.608 / 1.453, B = 32 1.708 / 1.562; Mistral-7B B = 16 3.92 / 4.10, B = 32 5.38 / 5.12: from 24 rows (below, rows x kv heads
    // workgroups leave CUs empty).  Closing build (no look-ahead set, QKV finish in the prologue), VALU / MFMA: B = 12 1.005 / 1.007, B = 16 1.052 / 1.072,
    // B = 17 1.247 / 1.159, B = 20 1.293 / 1.200, context 2k B = 17 1.553 / 1.340; Mistral-7B B = 16 3.92 / 4.07, B = 17 4.89 / 4.78: from 17 rows.
    // With AttnArgs.raw_* set the launch also finishes the QKV product (attn.raw_fuse: B = 32 1.335 -> 1.326, B = 8 1.000 -> 0.978)
    if constexpr (!QKN && DT != tgx::DT_F32) {
      if (attn_batch_on_mfma(c, R)) {
        const dim3 gm(a.kv_heads, R);
        if (!(c->debug_skip & 1)) {
          constexpr size_t lds4 = tgx::attn_mfma_lds_bytes<HD, 4>(), ldsr = tgx::attn_mfma_raw_lds_bytes<HD, 4>();
          // head_dim 64: the form without the second K / V register set — 208 instead of 309 registers, two workgroups per CU.  Measured ms/step with /
          // without (Llama-3.2-1B, context 600): B = 32 1.317 / 1.314, B = 48 1.814 / 1.713, B = 64 1.902 / 1.796 — never behind: the default
          // (option attn.batch_la: 1 = look-ahead, -1 = only while the workgroups number at most one per CU)
          const bool la = HD != 64 || (c->attn_batch_la >= 0 ? c->attn_batch_la != 0 : (int)(gm.x * gm.y) <= c->num_cus);
          // eight waves per workgroup (head_dim 64, option attn.batch_nw8: 1 = while the workgroups number at most one per CU, 2 = always): the blocks of 64 keys and the
          // QKV finish's slab sums spread over twice the waves
          if constexpr (HD == 64) {
            if ((a.raw_part || a.raw_qkv) && (c->attn_batch_nw8 >= 2 || (c->attn_batch_nw8 == 1 && (int)(gm.x * gm.y) <= c->num_cus))) {
              constexpr size_t lds8 = tgx::attn_mfma_raw_lds_bytes<HD, 8>();
              hipLaunchKernelGGL((tgx::attn_decode_mfma_kernel<DT, HD, 8, true, false>), gm, dim3(512), lds8, c->stream, a);
              return;
            }
          }
          if (a.raw_part || a.raw_qkv) {     // + the QKV product's finish
            if (la) hipLaunchKernelGGL((tgx::attn_decode_mfma_kernel<DT, HD, 4, true>), gm, dim3(256), ldsr, c->stream, a);
            else hipLaunchKernelGGL((tgx::attn_decode_mfma_kernel<DT, HD, 4, true, false>), gm, dim3(256), ldsr, c->stream, a);
          } else {
            if (la) hipLaunchKernelGGL((tgx::attn_decode_mfma_kernel<DT, HD, 4>), gm, dim3(256), lds4, c->stream, a);
            else hipLaunchKernelGGL((tgx::attn_decode_mfma_kernel<DT, HD, 4, false, false>), gm, dim3(256), lds4, c->stream, a);
          }
        }
        return;
      }
    }
    int dg = 1;
    // (two heads per workgroup as soon as one workgroup per query head would exceed one round of CUs: Llama-3.2-1B B = 9 0.937 -> 0.905 ms/step, B = 10 at context 2k
    //  1.194 -> 1.100; at 8 rows and fewer one head per workgroup stays ahead: B = 8 0.870 vs 0.895)
    if (!QKN && c->attn_direct_g > 0) dg = R >= 24 ? (HD == 64 ? 4 : 2) : ((R >= 12 || R * a.heads > c->num_cus) ? 2 : 1);
    if (!QKN && c->attn_direct_g < 0) dg = -c->attn_direct_g;          // experiments: force
    dg = std::min(dg, gfull);
    const bool raw = a.raw_part || a.raw_qkv;       // + the QKV product's finish in the prologue (batched step)
    if (dg >= 2 && gfull % dg == 0) {
      const dim3 gridg(a.kv_heads, R, gfull / dg), blkg(1024);
      if constexpr (!QKN && DT != tgx::DT_F32) {
        if (raw && !(c->debug_skip & 1)) {
          if (dg == 2) hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 2, 16, false, false, false, true>), gridg, blkg, 0, c->stream, a);
          else hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 4, 16, false, false, false, true>), gridg, blkg, 0, c->stream, a);
          return;
        }
      }
      if constexpr (!QKN) {
        if (!(c->debug_skip & 1)) {
          if (dg == 2) hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 2, 16, false>), gridg, blkg, 0, c->stream, a);
          else hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 4, 16, false>), gridg, blkg, 0, c->stream, a);
        }
      }
      return;
    }
    if constexpr (!QKN && DT != tgx::DT_F32) {
      if (raw && !(c->debug_skip & 1)) {          // (every remaining direct form of a batched step is one head per workgroup: also a group size dg does not divide)
        hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 1, 16, false, false, false, true>), dim3(a.kv_heads, R, gfull), dim3(1024), 0, c->stream, a);
        return;
      }
    }
    // very short contexts (option attn.direct_nw4: keys up to which the direct form runs FOUR waves per head instead of sixteen): one pass of a 4-wave
    // workgroup covers 128 keys at head_dim 64 (64 at 128), and four records merge faster than sixteen
    if (c->attn_nw4 && dg == 1) {
      const dim3 grid4(a.kv_heads, R, gfull), blk4(256);
      if (!(c->debug_skip & 1)) hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 1, 4, QKN>), grid4, blk4, 0, c->stream, a);
      return;
    }
    const dim3 grid(a.kv_heads, R, gfull), blk(1024);
    if (!(c->debug_skip & 1)) hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 1, 16, QKN>), grid, blk, 0, c->stream, a);
    return;
  }
  if (c->attn_mfma && !QKN && DT != tgx::DT_F32) {   // long context: QK^T and PV on the matrix cores, the kv group's query heads as the narrow operand
    if constexpr (!QKN && DT != tgx::DT_F32) {
      const dim3 gm(a.kv_heads * a.nsplit, R), bm(256);
      hipLaunchKernelGGL((tgx::attn_decode_mfma_kernel<DT, HD>), gm, bm, tgx::attn_mfma_lds_bytes<HD>(), c->stream, a);
    }
    launch_combine();
    return;
  }
  int gx = a.kv_heads * a.nsplit;
  if (a.pf.t[0].W && R == 1 && !QKN && DT != tgx::DT_F32) {   // prefetch workgroups behind the split workgroups; the x extent a multiple of 8 (block x -> XCD x % 8 in every z slab)
    a.pf.n_compute = gx; a.pf.stride = c->pf_stride; a.pf.sink = reinterpret_cast<unsigned*>(c->nop_word);
    gx += std::max(8, c->pf_wgs / ngroups); gx = (gx + 7) / 8 * 8;
    const dim3 gridp(gx, R, ngroups), blkp(256);
    if constexpr (!QKN && DT != tgx::DT_F32) {
      if (!(c->debug_skip & 1)) switch (G) {
        case 1: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 1, 4, false, true>), gridp, blkp, 0, c->stream, a); break;
        case 2: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 2, 4, false, true>), gridp, blkp, 0, c->stream, a); break;
        case 3: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 3, 4, false, true>), gridp, blkp, 0, c->stream, a); break;
        default: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 4, 4, false, true>), gridp, blkp, 0, c->stream, a); break;
      }
    }
    launch_combine();
    return;
  }
  a.pf.n_compute = 0;
  if (c->attn_ticket && c->attn_tickets && !QKN && DT != tgx::DT_F32 && ngroups <= 8 && !attn_fold_ok(c, R)) {   // in-kernel combine: no second launch
    if constexpr (!QKN && DT != tgx::DT_F32) {
      a.fold_ticket = c->attn_tickets;
      const dim3 gridf(gx, R, ngroups), blkf(256);
      switch (G) {
        case 1: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 1, 4, false, false, true>), gridf, blkf, 0, c->stream, a); break;
        case 2: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 2, 4, false, false, true>), gridf, blkf, 0, c->stream, a); break;
        case 3: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 3, 4, false, false, true>), gridf, blkf, 0, c->stream, a); break;
        default: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 4, 4, false, false, true>), gridf, blkf, 0, c->stream, a); break;
      }
    }
    return;
  }
  const dim3 grid(gx, R, ngroups), blk(256);
  if (!(c->debug_skip & 1)) switch (G) {
    case 1: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 1, 4, QKN>), grid, blk, 0, c->stream, a); break;
    case 2: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 2, 4, QKN>), grid, blk, 0, c->stream, a); break;
    case 3: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 3, 4, QKN>), grid, blk, 0, c->stream, a); break;
    default: hipLaunchKernelGGL((tgx::attn_decode_kernel<DT, HD, 4, 4, QKN>), grid, blk, 0, c->stream, a); break;
  }
  // the split partials are merged by the o_proj launch's prologue (PRO_ATTNCOMB) unless that fold is off or its LDS stage would not fit
  launch_combine();
}

void launch_attn(tgx_ctx* c, const tgx::AttnArgs& a, int R) {
  // k_raw set: Qwen3's q/k norm + RoPE + cache append happen inside the attention launch (head_dim 128: every released Qwen3 size)
  if (a.k_raw && c->d.head_dim == 128) { TGX_DT_SWITCH(c->dt, (launch_attn_g<DT, 128, true>(c, a, R))) return; }
  TGX_DT_SWITCH(c->dt, if (c->d.head_dim == 64) launch_attn_g<DT, 64>(c, a, R); else launch_attn_g<DT, 128>(c, a, R))
}

void fill_strides(const tgx_ctx* c, tgx::GemvArgs& a) {
  const tgx_model_desc& d = c->d;
  a.x_stride = 0; a.out_stride = 0;       // set per call site (x and out come from different slabs)
  a.q_stride = (long long)d.heads * d.head_dim; a.kraw_stride = (long long)d.kv_heads * d.head_dim;
  a.kv_stride = (long long)c->kv_row_elems; a.logits_stride = d.vocab; a.part_stride = c->lm_grid;
}

// Qwen3: independent batch rows (each its own cache) take the q/k norm inside the attention launch; the positions of one sequence that a
// prefill-by-steps pass handles together (kv_stride 0) need each other's finished keys, so they keep the separate norm launch
bool qk_fused(const tgx_ctx* c, long long kv_stride) { return c->d.qk_norm && c->d.head_dim == 128 && kv_stride != 0 && c->qk_fuse; }

// ---- L2 prefetch chaining (kernels/l2_prefetch.h) ---------------------------------------------------------------------------------
// The consumer launch of class `cls` at layer `l` (batch 1) as a prefetch target: its weight matrix, unit -> rows map and workgroup -> units
// map exactly as launch_gemv will set them, cut at `kb` KB per XCD.
tgx::PfTarget pf_target(const tgx_ctx* c, int l, int cls, int kb, int skip_kb = 0) {
  tgx::PfTarget t{};
  const tgx_model_desc& d = c->d;
  if (!c->pf_mode || kb <= 0 || l < 0 || l >= d.layers) return t;
  const int H = d.hidden, I = d.inter, hd = d.head_dim, qd = d.heads * hd, kvd = d.kv_heads * hd;
  const LayerW& w = c->L[(size_t)l];
  int N = 0, K = 0, units = 0;
  switch (cls) {
    case TGX_KERNEL_QKV: t.W = w.wqkv; N = qd + 2 * kvd; K = H; units = N / 2; t.rows_map = tgx::PF_ROWS_ROPE; break;
    case TGX_KERNEL_OPROJ: t.W = w.wo; N = H; K = qd; units = (H + 1) / 2; t.rows_map = tgx::PF_ROWS_PAIR; break;
    case TGX_KERNEL_GATEUP: t.W = w.wgu; N = 2 * I; K = H; units = I; t.rows_map = tgx::PF_ROWS_SILU; break;
    case TGX_KERNEL_DOWN: if (I > 16384) return t; t.W = w.wdown; N = H; K = I; units = (H + 1) / 2; t.rows_map = tgx::PF_ROWS_PAIR; break;
    default: return t;
  }
  const int ks = gemv_auto_ks(K, c->tune[cls].ks);
  t.grid = gemv_grid(c, units, ks, c->tune[cls].bpc); t.upb = 4 / ks; t.units = units; t.N = N; t.hd = hd;
  t.row_bytes = (int)(K * c->esz);
  const long long wp_bytes = 2LL * t.upb * t.row_bytes;                                     // one consumer workgroup, one pass
  const long long all = ((long long)units + 8LL * t.upb - 1) / (8LL * t.upb);              // passes x workgroups of one XCD
  t.wp_start = (int)std::min(all, (long long)skip_kb * 1024 / wp_bytes);
  t.budget_wp = (int)std::max<long long>(1, std::min(all, t.wp_start + (long long)kb * 1024 / wp_bytes));
  return t;
}
bool pf_active(const tgx_ctx* c, int R, const float* resid, const RowState* rv) { return c->pf_mode && R == 1 && resid == rv[0].x && !c->gpt2 && c->dt != tgx::DT_F32; }

// One kernel class of one decoder layer for R rows (batch rows of the slabs, or the chunk rows of a prefill-by-steps pass).  `resid` is the residual stream of row0 that the
// o_proj/down epilogues update (slab_x in the real pass; a scratch vector when tgx_profile_decode replays a class).
void launch_layer_kernel(tgx_ctx* c, RowState* rv, int R, int l, int cls, float* resid, long long kv_stride) {
  const tgx_model_desc& d = c->d;
  RowState& r = rv[0];   // R consecutive row views with the slabs' row strides; kv_stride = 0 when the rows are positions of ONE sequence
  const int H = d.hidden, I = d.inter, hd = d.head_dim, qd = d.heads * hd, kvd = d.kv_heads * hd;
  const size_t kv_layer = (size_t)d.kv_heads * d.max_ctx * hd * c->esz;   // bytes (ebyte pointers)
  const LayerW& w = c->L[(size_t)l];
  switch (cls) {
    case TGX_KERNEL_QKV: {   // input_layernorm -> qkv_proj -> RoPE -> cache append   (DecoderLayer.h:40, Attention.h:94-106)
      tgx::GemvArgs a{};
      fill_strides(c, a);
      a.W = w.wqkv; a.bias = w.bqkv; a.x = r.x; a.x_stride = H; a.norm_w = w.in_norm; a.eps = d.norm_eps;
      a.N = qd + 2 * kvd; a.K = H; a.units = a.N / 2;
      a.q_out = r.q; a.k_cache = r.kcache + (size_t)l * kv_layer; a.v_cache = r.vcache + (size_t)l * kv_layer; a.kv_stride = kv_stride;
      a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.pos = r.pos;
      a.heads = d.heads; a.kv_heads = d.kv_heads; a.hd = hd; a.max_ctx = d.max_ctx;
      a.raw_qk = d.qk_norm ? 1 : 0; a.k_raw = r.k_raw;
      if (pf_active(c, R, resid, rv)) a.pf.t[0] = pf_target(c, l, TGX_KERNEL_OPROJ, c->pf_oproj_kb);
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
        n.q_stride = qd; n.kraw_stride = kvd; n.kv_stride = kv_stride;
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
      a.q_stride = qd; a.kv_stride = kv_stride; a.part_stride = (long long)c->attn_part_row; a.dbg = c->debug_attn;
      if (pf_active(c, R, resid, rv)) { a.pf.t[0] = pf_target(c, l, TGX_KERNEL_GATEUP, c->pf_gu_kb); a.pf_comb.t[0] = pf_target(c, l, TGX_KERNEL_GATEUP, c->pf_comb_kb, c->pf_gu_kb); }
      if (qk_fused(c, kv_stride)) {
        a.k_raw = r.k_raw; a.kraw_stride = kvd; a.q_norm_w = w.q_norm; a.k_norm_w = w.k_norm;
        a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.eps = d.norm_eps;
      }
      launch_attn(c, a, R);
      break;
    }
    case TGX_KERNEL_OPROJ: { // o_proj + residual                                     (Attention.h:90, DecoderLayer.h:40)
      tgx::GemvArgs a{};
      fill_strides(c, a);
      a.W = w.wo; a.bias = w.bo; a.x = r.attn; a.x_stride = qd; a.N = H; a.K = qd; a.units = (H + 1) / 2; a.out = resid; a.out_stride = H; a.hd = 2;
      if (attn_fold_ok(c, R)) {   // split-form attention: the input vector is built from the split partials inside this launch
        a.attn_part = r.attn_part; a.part_in_stride = (long long)c->attn_part_row; a.pos = r.pos;
        a.attn_nsplit = c->attn_nsplit; a.attn_hd = hd; a.attn_step = 4 * (64 / (hd / 8)) * 4;   // attn_decode_kernel: NW waves x TPW tokens x UNR wave-loads per block
        launch_gemv<tgx::PRO_ATTNCOMB, tgx::EPI_RESIDUAL>(c, a, TGX_KERNEL_OPROJ, R);
        break;
      }
      if (pf_active(c, R, resid, rv)) a.pf.t[0] = pf_target(c, l, TGX_KERNEL_GATEUP, c->pf_oproj_gu_kb, c->pf_gu_kb + c->pf_comb_kb);
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
      if (pf_active(c, R, resid, rv)) a.pf.t[0] = pf_target(c, l, TGX_KERNEL_DOWN, c->pf_dn_kb);
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
      if (pf_active(c, R, resid, rv)) a.pf.t[0] = pf_target(c, l + 1, TGX_KERNEL_QKV, c->pf_qkv_kb);
      launch_gemv<tgx::PRO_PLAIN, tgx::EPI_RESIDUAL>(c, a, TGX_KERNEL_DOWN, R);
      break;
    }
    default: break;
  }
}

// All decoder layers for rows [row0, row0+R): their current tokens' embeddings sit in slab_x, positions in slab_pos.
// == for (auto& layer : layers_) x = layer->forward(x)  (GPTModel.h:53-55)
// ---- persistent engine (kernels/engine.h) ----------------------------------------------------------------------------------------
int engine_ring_slots(const tgx_ctx* c, int xb0, int xb1) {
  int ns = (int)((160 * 1024 - (size_t)xb0 - (size_t)xb1 - tgx::ENG_RGS * tgx::ENG_KCMAX * 32 - 1024) / tgx::ENG_SLOT);
  if (ns > 15) ns = 15;
  if (c->engine_ns > 0 && c->engine_ns < ns) ns = c->engine_ns;
  return ns;
}
// Geometries the engine's tiles cover: every K a multiple of 1024 (8 rows x 1024 k tiles), row counts multiples of 8, head_dim 64 / 128,
// a normed input of at most 4096 values, and at least 4 ring slots next to the staged inputs.
bool engine_ok(const tgx_ctx* c, int R, long long kv_stride) {
  const tgx_model_desc& d = c->d;
  if (!c->engine_mode || R != 1 || kv_stride == 0 || c->gpt2 || d.qk_norm || c->dt == tgx::DT_F32 || !c->eng_gh) return false;
  const int H = d.hidden, I = d.inter, qd = d.heads * d.head_dim, kvd = d.kv_heads * d.head_dim;
  if (H % 1024 || I % 1024 || qd % 1024 || (qd + 2 * kvd) % 8 || H > 4096 || I > 1024 * tgx::ENG_KCMAX || qd > 8192) return false;
  if (d.head_dim != 64 && d.head_dim != 128) return false;
  if ((H / 8 + c->num_cus - 1) / c->num_cus * 8 > tgx::ENG_RES_MAX) return false;
  return engine_ring_slots(c, std::max(qd, I) * 4, H * 4) >= 4;
}

// One engine launch for layer l of row r: mode 1 = {gate_up, down}; mode 2 = {o_proj, gate_up, down, qkv of layer l + 1 (if any)}
void launch_engine(tgx_ctx* c, RowState& r, int l) {
  const tgx_model_desc& d = c->d;
  const int H = d.hidden, I = d.inter, hd = d.head_dim, qd = d.heads * hd, kvd = d.kv_heads * hd;
  const size_t kv_layer = (size_t)d.kv_heads * d.max_ctx * hd * c->esz;
  const LayerW& w = c->L[(size_t)l];
  tgx::EngArgs a{};
  a.thin = c->engine_thin; a.depth = c->engine_depth;
  a.x_in = r.x; a.pos = r.pos; a.heads = d.heads; a.kv_heads = d.kv_heads; a.hd = hd; a.max_ctx = d.max_ctx; a.eps = d.norm_eps;
  a.epoch = c->eng_epoch; a.err = c->eng_err;
  a.stats = (c->engine_stats && c->eng_stats) ? c->eng_stats + (size_t)l * c->num_cus * tgx::ENG_NSTAT : nullptr;
  int n = 0, edge = 0;
  if (c->engine_mode >= 2) {
    tgx::EngOp& o = a.op[n++];
    o.W = w.wo; o.bias = w.bo; o.N = H; o.K = qd; o.epi = tgx::EOP_RESID; o.in_plain = r.attn; o.out_gran = c->eng_gx1; o.out_tag = edge;
  }
  {
    tgx::EngOp& o = a.op[n++];
    o.W = w.wgu; o.norm_w = w.post_norm; o.N = 2 * I; o.K = H; o.epi = tgx::EOP_SILU;
    if (c->engine_mode >= 2) { o.in_gran = c->eng_gx1; o.in_tag = edge++; } else o.in_plain = r.x;
    o.out_gran = c->eng_gh; o.out_tag = edge;
  }
  {
    tgx::EngOp& o = a.op[n++];
    o.W = w.wdown; o.bias = w.bdown; o.N = H; o.K = I; o.epi = tgx::EOP_RESID; o.in_gran = c->eng_gh; o.in_tag = edge++; o.out_plain = r.x;
  }
  if (c->engine_mode >= 2 && l + 1 < d.layers) {
    const LayerW& nw = c->L[(size_t)l + 1];
    a.op[n - 1].out_gran = c->eng_gx2; a.op[n - 1].out_tag = edge;
    tgx::EngOp& o = a.op[n++];
    o.W = nw.wqkv; o.bias = nw.bqkv; o.norm_w = nw.in_norm; o.N = qd + 2 * kvd; o.K = H; o.epi = tgx::EOP_QKV; o.in_gran = c->eng_gx2; o.in_tag = edge++;
    a.q_out = r.q; a.k_cache = r.kcache + (size_t)(l + 1) * kv_layer; a.v_cache = r.vcache + (size_t)(l + 1) * kv_layer;
    a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin;
  }
  a.nops = n;
  // the ops' inputs alternate between two LDS staging buffers: sizes by the widest input of each parity
  int xb[2] = {0, 0};
  for (int k = 0; k < n; k++) { xb[k & 1] = std::max(xb[k & 1], a.op[k].K * 4); tgx::eng_plan_op(a.op[k], c->num_cus); }
  a.xb_bytes[0] = xb[0]; a.xb_bytes[1] = xb[1];
  a.ns = engine_ring_slots(c, xb[0], xb[1]);
  const size_t lds = tgx::eng_lds_bytes(a.ns, xb[0], xb[1]);
  const dim3 g(c->num_cus), b(tgx::ENG_THREADS);
  if (c->dt == tgx::DT_F16) {
    if (a.stats) hipLaunchKernelGGL((tgx::engine_kernel<tgx::DT_F16, true>), g, b, lds, c->stream, a);
    else hipLaunchKernelGGL((tgx::engine_kernel<tgx::DT_F16, false>), g, b, lds, c->stream, a);
  } else {
    if (a.stats) hipLaunchKernelGGL((tgx::engine_kernel<tgx::DT_BF16, true>), g, b, lds, c->stream, a);
    else hipLaunchKernelGGL((tgx::engine_kernel<tgx::DT_BF16, false>), g, b, lds, c->stream, a);
  }
}

void launch_layers(tgx_ctx* c, RowState* rv, int R, long long kv_stride) {
  if (engine_ok(c, R, kv_stride)) {
    // == the same layer sequence (GPTModel.h:53-55) with the Linears between two attentions in ONE persistent launch
    for (int l = 0; l < c->d.layers; l++) {
      if (c->engine_mode == 1 || l == 0) launch_layer_kernel(c, rv, R, l, TGX_KERNEL_QKV, rv[0].x, kv_stride);
      launch_layer_kernel(c, rv, R, l, TGX_KERNEL_ATTN, rv[0].x, kv_stride);
      if (c->engine_mode == 1) launch_layer_kernel(c, rv, R, l, TGX_KERNEL_OPROJ, rv[0].x, kv_stride);
      launch_engine(c, rv[0], l);
    }
    return;
  }
  for (int l = 0; l < c->d.layers; l++) {
    for (int cls = TGX_KERNEL_QKV; cls <= TGX_KERNEL_DOWN; cls++) launch_layer_kernel(c, rv, R, l, cls, rv[0].x, kv_stride);
    for (int i = 0; i < c->debug_nops; i++) hipLaunchKernelGGL(tgx::nop_kernel, dim3(1), dim3(64), 0, c->stream, c->nop_word);
  }
}
void launch_layers(tgx_ctx* c, int row0, int R) { launch_layers(c, &c->rows[(size_t)row0], R, (long long)c->kv_row_elems); }

// ---- batched prefill (kernels/prefill.h) -------------------------------------------------------------------------
bool prefill_shapes_ok(const tgx_model_desc& d) {
  return d.hidden % 64 == 0 && (d.heads * d.head_dim) % 64 == 0 && d.inter % 64 == 0;
}

void drop_step_graphs(tgx_ctx* c);

int ensure_prefill_ws(tgx_ctx* c, int S) {
  if (S <= c->ws_rows) return TGX_OK;
  const tgx_model_desc& d = c->d;
  const size_t H = (size_t)d.hidden, qd = (size_t)d.heads * d.head_dim, kvd = (size_t)d.kv_heads * d.head_dim, I = (size_t)d.inter;
  const size_t wout = qd + 2 * kvd, wa = std::max(H, qd);   // the gate_up product leaves no fp32 intermediate (GEMM_SILU)
  drop_step_graphs(c);                                       // a captured batched decode step holds pointers into the old workspace
  HIP_OK(c, hipStreamSynchronize(c->stream));
  auto fr = [](void* p) { if (p) (void)hipFree(p); };
  fr(c->ws_x); fr(c->ws_out); fr(c->ws_ah); fr(c->ws_al); fr(c->ws_al2); fr(c->ws_qh); fr(c->ws_ql); fr(c->ws_hh); fr(c->ws_hl);
  c->ws_al2 = nullptr;
  c->ws_x = nullptr; c->ws_out = nullptr; c->ws_ah = c->ws_al = c->ws_qh = c->ws_ql = c->ws_hh = c->ws_hl = nullptr; c->ws_rows = 0;
  const size_t rows = (size_t)S;
  // fp32 storage: ws_ah / ws_qh / ws_hh hold fp32 rows (the fp32 GEMM's A operands, the rotated queries); the lo terms are unused
  const size_t te = c->dt == tgx::DT_F32 ? 4 : 2, lo = c->dt == tgx::DT_F32 ? 0 : 1;
  HIP_OK(c, hipMalloc((void**)&c->ws_x, rows * H * 4));
  HIP_OK(c, hipMalloc((void**)&c->ws_out, rows * wout * 4));
  HIP_OK(c, hipMalloc((void**)&c->ws_ah, rows * wa * te));
  HIP_OK(c, hipMalloc((void**)&c->ws_al, rows * wa * 2 * lo + 16));
  HIP_OK(c, hipMalloc((void**)&c->ws_al2, rows * H * 2 * lo + 16));
  HIP_OK(c, hipMalloc((void**)&c->ws_qh, rows * qd * te));
  HIP_OK(c, hipMalloc((void**)&c->ws_ql, rows * qd * 2 * lo + 16));
  HIP_OK(c, hipMalloc((void**)&c->ws_hh, rows * I * te));
  HIP_OK(c, hipMalloc((void**)&c->ws_hl, rows * I * 2 * lo + 16));
  if (c->dt == tgx::DT_F32) {
    if (c->ws_pos) (void)hipFree(c->ws_pos);
    c->ws_pos = nullptr;
    HIP_OK(c, hipMalloc((void**)&c->ws_pos, rows * 4));
    if (!c->ws_attn_part) HIP_OK(c, hipMalloc((void**)&c->ws_attn_part, (size_t)F32_ATTN_ROWS * c->attn_part_row * 4));
  }
  c->ws_rows = S;
  return TGX_OK;
}

// defer (optional, RESIDUAL / STORE products): when the product is split over K, leave the slabs in ws_part for the consumer kernel to sum
// (rmsnorm_split_kernel / rope_kv_split_kernel: same z order, one launch and one pass over the rows less) and report the slab count; 1 = done here.
void launch_gemm(tgx_ctx* c, int epi, const ebyte* B_, const ebyte* bias_, float* C, int M, int N, int K, int ldc, bool three_terms = false,
                 const bf16_t* a_hi = nullptr, const bf16_t* a_lo = nullptr, int three_from = 0, int* defer = nullptr) {
  if (defer) *defer = 1;
  const bf16_t* B = reinterpret_cast<const bf16_t*>(B_);   // 16-bit storage (bf16 or fp16 bit patterns); fp32 storage never gets here
  const bf16_t* bias = reinterpret_cast<const bf16_t*>(bias_);
  tgx::GemmArgs g{};
  g.A_hi = a_hi ? a_hi : c->ws_ah; g.A_lo = a_lo ? a_lo : c->ws_al; g.A_lo2 = three_terms ? c->ws_al2 : nullptr;
  g.inter = N / 2; g.out_hi = c->ws_hh; g.out_lo = c->ws_hl;
  g.B = B; g.bias = bias; g.C = C; g.M = M; g.N = N; g.K = K; g.ldc = ldc; g.three_from = three_from;
  // few column tiles (N = hidden) -> 64-row tiles, so that at least two workgroups share a CU
  const bool few = ((N + tgx::GBN - 1) / tgx::GBN) * ((M + tgx::GBM - 1) / tgx::GBM) < 2 * c->num_cus;
  // measured (tools/prefill_bench.py --gemm-tm, Llama-3.2-1B, S = 2048): this policy 15.0 ms, 64-row tiles also for the three-term
  // QKV product 15.3, 128-row tiles everywhere 15.85, 64-row tiles everywhere 15.9
  const bool small = (epi == tgx::GEMM_SILU || epi == tgx::GEMM_GELU) ? false : (c->gemm_tm ? c->gemm_tm == 64 : (few && !three_terms));
  const int tm = small ? 64 : tgx::GBM;
  const dim3 grid((N + tgx::GBN - 1) / tgx::GBN, (M + tm - 1) / tm), blk(256);
  const size_t dyn = three_terms ? (size_t)tm * tgx::GLD * 2 : 0;      // LDS tile of the third term
  // few row tiles (a short prompt): the tiles alone cannot stream the weights at rate (S <= 96 cost a flat 4.7 ms on Llama-3.2-1B) —
  // split K over blockIdx.z until ~2 workgroups per CU exist; the slabs are summed in z order by a second launch (deterministic)
  const int ntiles = (int)(grid.x * grid.y), ktiles = K / tgx::GBK;
  int nsplit = 1;
  if (c->gemm_splitk && ntiles < c->num_cus && K % tgx::GBK == 0) nsplit = std::min(std::min(16, ktiles), (2 * c->num_cus + ntiles - 1) / ntiles);
  if (nsplit > 1) {
    const size_t need = (size_t)nsplit * M * N * 4;
    if (need > c->ws_part_bytes) {
      drop_step_graphs(c);              // a captured batched decode step points into the old slab buffer
      (void)hipStreamSynchronize(c->stream);
      if (c->ws_part) (void)hipFree(c->ws_part);
      c->ws_part = nullptr; c->ws_part_bytes = 0;
      if (hipMalloc((void**)&c->ws_part, need) == hipSuccess) c->ws_part_bytes = need; else nsplit = 1;
    }
  }
  if (nsplit > 1) {
    g.part = c->ws_part; g.nsplit = nsplit; g.interleave = epi == tgx::GEMM_SILU ? 1 : 0;
    g.k_per = ((ktiles + nsplit - 1) / nsplit) * tgx::GBK;
    const dim3 gz(grid.x, grid.y, nsplit);
    const size_t nout = (size_t)M * (epi == tgx::GEMM_SILU ? N / 2 : N);
    const dim3 rg((unsigned)((nout + 255) / 256));
    // the slabs' GEMM: operand tiles by LDS-DMA (round 3; option prefill.splitk_dma) — the register-staged kernel streamed a short prompt's weights at
    // 1-2 TB/s (S = 48: gate_up 32 us for 67 MB); 64-row tiles whenever the prompt fits them, k = 64 per stage (32 for the 128-row three-term tile)
    // measured (Llama-3.2-1B, ms per prompt, DMA vs register-staged slabs): S = 40 1.61 / 1.79, 48 1.65 / 1.76, 64 1.71 / 1.87; 96 2.08 / 1.99, 128 2.12 / 2.09,
    // 256 2.65 / 2.68; Mistral-7B S = 48 5.40 / 6.23 — the 64-row tile wins, the 128-row one does not: prompts of <= 64 rows only (value 2 = always)
    const bool dma_part = c->splitk_dma && (M <= 64 || c->splitk_dma == 2) && (c->gemm_dma & 3) && g.k_per % 64 == 0 && K % 64 == 0;
    if (dma_part) {
      const int mi = (small || M <= 64) ? 1 : 2;
      const int dbk = (mi == 2 && three_terms) ? 32 : 64;
      const dim3 gd((N + tgx::GBN - 1) / tgx::GBN, (M + 64 * mi - 1) / (64 * mi), nsplit);
      const size_t lds = tgx::gemm_dma_lds_bytes(mi, three_terms, dbk, 2);
      TGX_DT16_SWITCH(c->dt,
        if (mi == 1) hipLaunchKernelGGL((tgx::gemm_dma_kernel<DT, tgx::GEMM_PARTIAL, 1, 64, 2>), gd, blk, lds, c->stream, g);
        else if (dbk == 64) hipLaunchKernelGGL((tgx::gemm_dma_kernel<DT, tgx::GEMM_PARTIAL, 2, 64, 2>), gd, blk, lds, c->stream, g);
        else hipLaunchKernelGGL((tgx::gemm_dma_kernel<DT, tgx::GEMM_PARTIAL, 2, 32, 2>), gd, blk, lds, c->stream, g);)
    }
    TGX_DT16_SWITCH(c->dt,
      if (dma_part) {}
      else if (small) hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_PARTIAL, 1>), gz, blk, dyn, c->stream, g);
      else hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_PARTIAL, 2>), gz, blk, dyn, c->stream, g);
      if (defer && c->defer_reduce && M >= c->defer_min_rows && (epi == tgx::GEMM_RESIDUAL || epi == tgx::GEMM_STORE)) { *defer = nsplit; }   // with few rows the row-wise consumers are too few workgroups to sum 16 slabs quickly (round 2, S = 64: 1.82 -> 1.87 ms; S = 256: 2.80 -> 2.70)
      else if (epi == tgx::GEMM_SILU) hipLaunchKernelGGL((tgx::gemm_splitk_reduce_kernel<DT, tgx::GEMM_SILU>), rg, blk, 0, c->stream, g);
      else if (epi == tgx::GEMM_GELU) hipLaunchKernelGGL((tgx::gemm_splitk_reduce_kernel<DT, tgx::GEMM_GELU>), rg, blk, 0, c->stream, g);
      else if (epi == tgx::GEMM_RESIDUAL) hipLaunchKernelGGL((tgx::gemm_splitk_reduce_kernel<DT, tgx::GEMM_RESIDUAL>), rg, blk, 0, c->stream, g);
      else hipLaunchKernelGGL((tgx::gemm_splitk_reduce_kernel<DT, tgx::GEMM_STORE>), rg, blk, 0, c->stream, g);)
    return;
  }
  if ((c->gemm_dma & 4) && K % 64 == 0 && !three_terms && (epi == tgx::GEMM_SILU || epi == tgx::GEMM_GELU) && ((N + 255) / 256) * ((M + 255) / 256) >= c->num_cus) {
    // the wide product (gate_up / c_fc) with enough 256 x 256 tiles to fill the chip: 8 waves, three-stage LDS-DMA ring
    const dim3 g8((N + 255) / 256, (M + 255) / 256), b8(512);
    const size_t lds8 = (size_t)3 * 3 * 256 * 32 * 2;
    TGX_DT16_SWITCH(c->dt,
      if (epi == tgx::GEMM_SILU) hipLaunchKernelGGL((tgx::gemm_dma8_kernel<DT, tgx::GEMM_SILU>), g8, b8, lds8, c->stream, g);
      else hipLaunchKernelGGL((tgx::gemm_dma8_kernel<DT, tgx::GEMM_GELU>), g8, b8, lds8, c->stream, g);)
    return;
  }
  if ((c->gemm_dma & 4) && c->hidden_256 && K % 64 == 0 && !three_terms && (epi == tgx::GEMM_RESIDUAL || epi == tgx::GEMM_STORE) &&
      ((N + 255) / 256) * ((M + 255) / 256) >= c->num_cus) {
    // the N = hidden products of a prompt long enough to give every CU a 256 x 256 tile (Llama-3.2-1B from 8192 rows, Mistral-7B from 4096): the wide
    // product's kernel with the plain fp32 epilogue (option prefill.hidden_256)
    const dim3 g8((N + 255) / 256, (M + 255) / 256), b8(512);
    const size_t lds8 = (size_t)3 * 3 * 256 * 32 * 2;
    TGX_DT16_SWITCH(c->dt,
      if (epi == tgx::GEMM_RESIDUAL) hipLaunchKernelGGL((tgx::gemm_dma8_kernel<DT, tgx::GEMM_RESIDUAL>), g8, b8, lds8, c->stream, g);
      else hipLaunchKernelGGL((tgx::gemm_dma8_kernel<DT, tgx::GEMM_STORE>), g8, b8, lds8, c->stream, g);)
    return;
  }
  if ((c->gemm_dma & 8) && c->wide_8k && K % 64 == 0 && !three_terms && epi == tgx::GEMM_SILU) {
    // the wide product of a prompt too short for 256 x 256 tiles (129-384 rows: 128-384 tiles of 128 x 128): the eight-wave kernel with the K step split between
    // wave pairs instead of the four-wave one (option prefill.wide_8k: Llama-3.2-1B S = 256 gate_up 61 us per layer)
    const int t128 = ((N + 127) / 128) * ((M + 127) / 128);
    if (2 * t128 >= c->num_cus && 2 * t128 <= c->wide_8k_max * c->num_cus) {
      const dim3 g8((N + 127) / 128, (M + 127) / 128), b8(512);
      const size_t lds8 = (size_t)3 * 3 * 128 * 64 * 2;
      TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_SILU>), g8, b8, lds8, c->stream, g))
      return;
    }
  }
  if ((c->gemm_dma & 8) && K % 64 == 0 && !three_terms && (epi == tgx::GEMM_RESIDUAL || epi == tgx::GEMM_STORE)) {
    // N = hidden products whose 128 x 128 tiles number between half a chip and a chip and a half: eight waves per tile, K step split between wave pairs
    const int t128 = ((N + 127) / 128) * ((M + 127) / 128);
    if (2 * t128 >= c->num_cus && 2 * t128 <= 3 * c->num_cus) {
      const dim3 g8((N + 127) / 128, (M + 127) / 128), b8(512);
      const size_t lds8 = (size_t)3 * 3 * 128 * 64 * 2;
      TGX_DT16_SWITCH(c->dt,
        if (epi == tgx::GEMM_RESIDUAL) hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_RESIDUAL>), g8, b8, lds8, c->stream, g);
        else hipLaunchKernelGGL((tgx::gemm_dma8k_kernel<DT, tgx::GEMM_STORE>), g8, b8, lds8, c->stream, g);)
      return;
    }
  }
  if ((c->gemm_dma & 3) && c->qkv_balanced && three_terms && epi == tgx::GEMM_STORE && three_from > 0 && three_from % tgx::GBN == 0 && N > three_from && K % 64 == 0 && M >= 128) {
    // the QKV product of a bf16 prompt: Q columns as two-term 128-row tiles, K / V columns as three-term 64-row tiles, ONE launch with
    // equal work per workgroup pair (kernels/gemm_dma.h gemm_dma_qkv_kernel)
    const int nq = (three_from / tgx::GBN) * ((M + 127) / 128), nkv = ((N - three_from + tgx::GBN - 1) / tgx::GBN) * ((M + 63) / 64);
    const size_t ldsq = std::max(tgx::gemm_dma_lds_bytes(2, false, 32, 2), tgx::gemm_dma_lds_bytes(1, true, 32, 2));
    TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::gemm_dma_qkv_kernel<DT, 32, 2>), dim3(nq + nkv), blk, ldsq, c->stream, g))
    return;
  }
  if ((c->gemm_dma & 3) && K % 64 == 0) {     // operand tiles by LDS-DMA into a ring of stages (kernels/gemm_dma.h): one barrier per K step
    // geometry per tile height (option prefill.gemm_dma bits 4-7 / 8-11 override: value = BK/32 + 4*(stages-2)): 128-row tiles k = 32 x 3 stages,
    // 64-row tiles k = 64 x 2 stages
    int sel = small ? ((c->gemm_dma >> 8) & 15) : ((c->gemm_dma >> 4) & 15);
    if (!sel) sel = small ? 2 : 5;
    const int dbk = (sel & 3) == 1 ? 32 : 64, ns = 2 + (sel >> 2);
    const size_t lds = tgx::gemm_dma_lds_bytes(small ? 1 : 2, three_terms, dbk, ns);
#define TGX_DMA2(EPI_, MI_, BK_) do { if (ns == 2) hipLaunchKernelGGL((tgx::gemm_dma_kernel<DT, EPI_, MI_, BK_, 2>), grid, blk, lds, c->stream, g); \
                                      else if (ns == 3) hipLaunchKernelGGL((tgx::gemm_dma_kernel<DT, EPI_, MI_, BK_, 3>), grid, blk, lds, c->stream, g); \
                                      else hipLaunchKernelGGL((tgx::gemm_dma_kernel<DT, EPI_, MI_, BK_, 4>), grid, blk, lds, c->stream, g); } while (0)
#define TGX_DMA(EPI_, MI_) do { if (dbk == 32) TGX_DMA2(EPI_, MI_, 32); else TGX_DMA2(EPI_, MI_, 64); } while (0)
    if (lds <= 160 * 1024) {
      TGX_DT16_SWITCH(c->dt,
        if (epi == tgx::GEMM_SILU) TGX_DMA(tgx::GEMM_SILU, 2);
        else if (epi == tgx::GEMM_GELU) TGX_DMA(tgx::GEMM_GELU, 2);
        else if (epi == tgx::GEMM_RESIDUAL) { if (small) TGX_DMA(tgx::GEMM_RESIDUAL, 1); else TGX_DMA(tgx::GEMM_RESIDUAL, 2); }
        else { if (small) TGX_DMA(tgx::GEMM_STORE, 1); else TGX_DMA(tgx::GEMM_STORE, 2); })
      return;
    }
#undef TGX_DMA
#undef TGX_DMA2
  }
  TGX_DT16_SWITCH(c->dt,
    if (epi == tgx::GEMM_SILU) hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_SILU, 2>), grid, blk, dyn, c->stream, g);
    else if (epi == tgx::GEMM_GELU) hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_GELU, 2>), grid, blk, dyn, c->stream, g);
    else if (small) {
      if (epi == tgx::GEMM_RESIDUAL) hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_RESIDUAL, 1>), grid, blk, dyn, c->stream, g);
      else hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_STORE, 1>), grid, blk, dyn, c->stream, g);
    } else {
      if (epi == tgx::GEMM_RESIDUAL) hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_RESIDUAL, 2>), grid, blk, dyn, c->stream, g);
      else hipLaunchKernelGGL((tgx::gemm_x2_kernel<DT, tgx::GEMM_STORE, 2>), grid, blk, dyn, c->stream, g);
    })
}

// All layers for S prompt positions of one row at once; leaves the last position's hidden state in row.x.
// == CausalLM::forward on [1,S] ids with an empty cache (GPTModel.h:51-56)
// NB batch rows [row0, row0 + NB) are stacked into ONE [NB*S] row block for the row-wise kernels and the GEMMs (the weights stream
// once for all of them); RoPE / cache append and attention run per batch row on its slice and its own cache.
void launch_prefill(tgx_ctx* c, int row0, int NB, int S) {
  const tgx_model_desc& d = c->d;
  const int H = d.hidden, I = d.inter, hd = d.head_dim, qd = d.heads * hd, kvd = d.kv_heads * hd;
  const size_t kv_layer = (size_t)d.kv_heads * d.max_ctx * hd;
  const int M = NB * S;
  const size_t wout = (size_t)qd + 2 * kvd;
  // GPT-2 (ModelGPT2.h:23-208): wte + wpe rows, LayerNorm with bias ahead of both products, a bias on every Conv1D, c_fc -> gelu_new;
  // its rotation tables are the identity, so the RoPE / cache-append kernel and the attention are the Llama family's
  TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::embed_rows_any_kernel<DT>, dim3(M), dim3(256), 0, c->stream, (const long long*)c->rows[(size_t)row0].prompt, (const void*)c->embed, (const void*)(c->gpt2 ? c->wpe : nullptr), c->ws_x, H, S, (long long)d.max_ctx, (int)c->past))
  int pend = 1;                     // slabs of the previous layer's down product still to be added to ws_x (1: none)
  const bf16_t* pend_bias = nullptr;
  for (int l = 0; l < d.layers; l++) {
    const LayerW& w = c->L[(size_t)l];
    // the QKV product feeds a second rounding (the KV cache): bf16 needs three split terms to reproduce the step path's cache
    // entries (two leave 1-8 % of them one ulp off); fp16's two terms already carry 22 bits
    const bool three = c->dt == tgx::DT_BF16;
    if (c->gpt2) { TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::norm_rows_kernel<DT, 1, 1>), dim3(M), dim3(256), 0, c->stream, (const float*)c->ws_x, (const void*)w.in_norm, (const void*)w.in_norm_b, d.norm_eps, H, (float*)nullptr, c->ws_ah, c->ws_al, three ? c->ws_al2 : (bf16_t*)nullptr)) }
    else { TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::rmsnorm_split_kernel<DT>, dim3(M), dim3(256), 0, c->stream, c->ws_x, (const bf16_t*)w.in_norm, d.norm_eps, H, c->ws_ah, c->ws_al, three ? c->ws_al2 : (bf16_t*)nullptr,
                                                  (const float*)(pend > 1 ? c->ws_part : nullptr), pend, (long long)M * H, pend_bias)) }
    pend = 1;
    int qsl = 1;
    launch_gemm(c, tgx::GEMM_STORE, w.wqkv, w.bqkv, c->ws_out, M, qd + 2 * kvd, H, qd + 2 * kvd, /*three_terms=*/three, nullptr, nullptr, /*three_from=*/qd, &qsl);   // Q columns: two terms
    for (int b = 0; b < NB; b++) {
      RowState& r = c->rows[(size_t)(row0 + b)];
      bf16_t* kc = reinterpret_cast<bf16_t*>(r.kcache);
      bf16_t* vc = reinterpret_cast<bf16_t*>(r.vcache);
      const size_t ro = (size_t)b * S;             // first workspace row of this batch row
      tgx::RopeKvArgs a{};
      a.QKV = c->ws_out + ro * wout; a.q_hi = c->ws_qh + ro * qd; a.q_lo = c->ws_ql + ro * qd;
      if (qsl > 1) { a.QKV = nullptr; a.part = c->ws_part + ro * wout; a.nsplit = qsl; a.slab = (long long)M * (long long)wout; a.bias = reinterpret_cast<const bf16_t*>(w.bqkv); }
      a.k_cache = kc + (size_t)l * kv_layer; a.v_cache = vc + (size_t)l * kv_layer;
      a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin;
      a.heads = d.heads; a.kv_heads = d.kv_heads; a.hd = hd; a.max_ctx = d.max_ctx; a.past = (int)c->past;
      a.q_norm_w = d.qk_norm ? (const bf16_t*)w.q_norm : nullptr; a.k_norm_w = d.qk_norm ? (const bf16_t*)w.k_norm : nullptr; a.eps = d.norm_eps;
      TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::rope_kv_split_kernel<DT>, dim3(S), dim3(256), 0, c->stream, a))
    }
    for (int b = 0; b < NB; b++) {
      RowState& r = c->rows[(size_t)(row0 + b)];
      bf16_t* kc = reinterpret_cast<bf16_t*>(r.kcache);
      bf16_t* vc = reinterpret_cast<bf16_t*>(r.vcache);
      const size_t ro = (size_t)b * S;
      tgx::AttnPrefillArgs a{};
      a.q_hi = c->ws_qh + ro * qd; a.q_lo = c->ws_ql + ro * qd; a.k_cache = kc + (size_t)l * kv_layer; a.v_cache = vc + (size_t)l * kv_layer;
      a.o_hi = c->ws_ah + ro * qd; a.o_lo = c->ws_al + ro * qd; a.S = S; a.heads = d.heads; a.kv_heads = d.kv_heads; a.max_ctx = d.max_ctx; a.past = (int)c->past;
      a.scale = 1.0f / sqrtf((float)hd); a.qblk_mirror = c->attn_mirror;
      const dim3 grid((S + 127) / 128, d.heads), blk(256);
      // head_dim 64 with three or more workgroups per CU: the one-tile look-ahead form at three waves per SIMD (prefill.h)
      const bool lean = hd == 64 && (int)(grid.x * grid.y) >= 3 * c->num_cus;
      TGX_DT16_SWITCH(c->dt, if (lean) hipLaunchKernelGGL((tgx::attn_prefill_kernel<DT, 64, 1>), grid, blk, 0, c->stream, a);
                             else if (hd == 64) hipLaunchKernelGGL((tgx::attn_prefill_kernel<DT, 64>), grid, blk, 0, c->stream, a);
                             else hipLaunchKernelGGL((tgx::attn_prefill_kernel<DT, 128, 1>), grid, blk, 0, c->stream, a))     // head_dim 128: two waves per SIMD only in this form (95 vs 138 µs per layer at S = 2048)
    }
    int osl = 1;
    launch_gemm(c, tgx::GEMM_RESIDUAL, w.wo, w.bo, c->ws_x, M, H, qd, H, false, nullptr, nullptr, 0, c->gpt2 ? nullptr : &osl);
    if (c->gpt2) {
      TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::norm_rows_kernel<DT, 1, 1>), dim3(M), dim3(256), 0, c->stream, (const float*)c->ws_x, (const void*)w.post_norm, (const void*)w.post_norm_b, d.norm_eps, H, (float*)nullptr, c->ws_ah, c->ws_al, (bf16_t*)nullptr))
      launch_gemm(c, tgx::GEMM_GELU, w.wgu, w.bfc, nullptr, M, I, H, I);             // c_fc + bias + gelu_new -> ws_hh / ws_hl
    } else {
      TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::rmsnorm_split_kernel<DT>, dim3(M), dim3(256), 0, c->stream, c->ws_x, (const bf16_t*)w.post_norm, d.norm_eps, H, c->ws_ah, c->ws_al, (bf16_t*)nullptr,
                                                (const float*)(osl > 1 ? c->ws_part : nullptr), osl, (long long)M * H, reinterpret_cast<const bf16_t*>(w.bo)))
      launch_gemm(c, tgx::GEMM_SILU, w.wgu, nullptr, nullptr, M, 2 * I, H, 2 * I);      // gate_up + siluMul -> ws_hh / ws_hl
    }
    // the down product's slabs wait for the next layer's input norm (the last layer, and GPT-2's LayerNorm path, finish them here)
    const bool can_defer = !c->gpt2 && l + 1 < d.layers;
    launch_gemm(c, tgx::GEMM_RESIDUAL, w.wdown, w.bdown, c->ws_x, M, H, I, H, false, c->ws_hh, c->ws_hl, 0, can_defer ? &pend : nullptr);
    pend_bias = reinterpret_cast<const bf16_t*>(w.bdown);
  }
  for (int b = 0; b < NB; b++)     // the last position of every batch row feeds lm_head
    (void)hipMemcpyAsync(c->rows[(size_t)(row0 + b)].x, c->ws_x + ((size_t)(b + 1) * S - 1) * H, (size_t)H * 4, hipMemcpyDeviceToDevice, c->stream);
}

tgx::FinalizeArgs make_finalize_args(tgx_ctx* c, int row, bool advance_pos, bool log_step);

// model.norm -> lm_head on the current position + per-workgroup argmax partials   (GPTModel.h:56-57)
// fuse_greedy: the launch's last-arriving workgroup also does the greedy finalize of its R rows (argmax over the partials, token publish,
// pastLength + 1, next embedding row) — decode steps with a greedy sampler; no finalize launch follows.
void launch_lm_head(tgx_ctx* c, int row0, int R, bool fuse_greedy = false) {
  const tgx_model_desc& d = c->d;
  RowState& r = c->rows[(size_t)row0];
  tgx::GemvArgs a{};
  fill_strides(c, a);
  a.W = d.tied ? c->embed : c->lm_head; a.x = r.x; a.x_stride = d.hidden; a.norm_w = c->final_norm; a.eps = d.norm_eps;
  a.N = d.vocab; a.K = d.hidden; a.units = (d.vocab + 1) / 2; a.hd = 2;
  a.logits = r.logits; a.part_val = r.part_val; a.part_idx = r.part_idx;
  if (fuse_greedy) {
    a.ticket = c->lm_ticket;
    for (int k = 0; k < R; k++)      // the rows' finalize arguments go to device memory ahead of the launch (a captured node like any other)
      hipLaunchKernelGGL(tgx::write_finalize_args_kernel, dim3(1), dim3(64), 0, c->stream, make_finalize_args(c, row0 + k, /*advance_pos=*/true, /*log_step=*/true), c->fin_dev + k);
    a.fin = c->fin_dev;
  }
  if (c->gpt2) {   // ln_f -> wte^T (tied head, ModelGPT2.h:170-176)
    a.norm_b = c->final_norm_b;
    launch_gemv<tgx::PRO_LAYERNORM, tgx::EPI_LOGITS>(c, a, TGX_KERNEL_LMHEAD, R);
    return;
  }
  launch_gemv<tgx::PRO_RMSNORM, tgx::EPI_LOGITS>(c, a, TGX_KERNEL_LMHEAD, R);
}

bool is_greedy(const tgx_sampler_cfg* s) {   // Sampler.cpp:15-21
  return !(s->temperature > 0.f || s->top_k > 0 || s->top_p < 1.f || s->min_p > 0.f);
}

tgx::FinalizeArgs make_finalize_args(tgx_ctx* c, int row, bool advance_pos, bool log_step) {
  RowState& r = c->rows[(size_t)row];
  tgx::FinalizeArgs a{};
  a.part_val = r.part_val; a.part_idx = r.part_idx; a.n_part = c->lm_grid;
  a.tok = r.tok; a.pos = r.pos; a.step = c->step; a.tok_log = c->tok_log; a.host_ring = c->mirror_to_host ? c->host_ring_dev : nullptr;
  a.log_cap = c->log_cap; a.ring_cap = HOST_RING;
  a.row = row; a.rows = c->batch;
  a.log = log_step ? 1 : 0; a.bump_step = (row == c->batch - 1) ? 1 : 0;
  a.embed = c->embed; a.x = r.x; a.H = c->d.hidden; a.V = c->d.vocab; a.advance_pos = advance_pos ? 1 : 0;
  a.wpe = c->gpt2 ? c->wpe : nullptr; a.n_pos = c->d.n_positions > 0 ? c->d.n_positions : 1;
  return a;
}

// == Sampler::sample on the logits of rows [row0, row0+R) (Sampler.cpp:23-79) + token publish / pastLength / next embedding.
// Greedy: one finalize launch per row.  Otherwise the staged sampler of kernels/sampler.h: ceil(V/1024) workgroups per row
// (rows on blockIdx.y), one launch per digit level of each active filter, the partial-sum stages, then one pick per row.
void launch_sample(tgx_ctx* c, int row0, int R, const tgx_sampler_cfg& cfg, bool advance_pos, bool log_step) {
  if (is_greedy(&cfg)) {
    for (int b = row0; b < row0 + R; b++) {
      const tgx::FinalizeArgs a = make_finalize_args(c, b, advance_pos, log_step);
      TGX_DT_SWITCH(c->dt, hipLaunchKernelGGL(tgx::finalize_greedy_kernel<DT>, dim3(1), dim3(256), 0, c->stream, a))
    }
    return;
  }
  const int V = c->d.vocab;
  RowState& r = c->rows[(size_t)row0];
  tgx::SampArgs a{};
  a.logits = r.logits; a.logits_stride = V;
  a.part_val = r.part_val; a.part_stride = c->lm_grid; a.n_part = c->lm_grid;
  a.sc = c->samp_scratch + row0;
  a.probs_out = r.probs; a.probs_stride = V;
  a.V = V; a.idx_bits = 1;
  while ((1 << a.idx_bits) < V) a.idx_bits++;
  a.temperature = cfg.temperature; a.top_k = cfg.top_k; a.top_p = cfg.top_p; a.min_p = cfg.min_p;
  const bool setK = cfg.top_k > 0, setP = cfg.top_p < 1.f, setM = cfg.min_p > 0.f;
  const int nwg = (V + tgx::SAMP_TILE - 1) / tgx::SAMP_TILE;
  const dim3 grid(nwg, R), blk(tgx::SAMP_WG);
  if (setK) for (int l = 0; l < tgx::SAMP_LEVELS; l++) { a.level = l; hipLaunchKernelGGL(tgx::samp_level_kernel<0>, grid, blk, 0, c->stream, a); }
  if (setP) for (int l = 0; l < tgx::SAMP_LEVELS; l++) { a.level = l; hipLaunchKernelGGL(tgx::samp_level_kernel<1>, grid, blk, 0, c->stream, a); }
  // the first stage after a filter's last level derives that filter's threshold from the level-4 histogram; later stages read it
  a.k_from_hist = (setK && !setP) ? 1 : 0;      // with top-p on, its first level already derived the top-k threshold
  a.p_from_hist = setP ? 1 : 0;
  if (setM) { hipLaunchKernelGGL(tgx::samp_sum_kernel<0>, grid, blk, 0, c->stream, a); a.k_from_hist = 0; a.p_from_hist = 0; }
  hipLaunchKernelGGL(tgx::samp_sum_kernel<1>, grid, blk, 0, c->stream, a);
  a.k_from_hist = 0; a.p_from_hist = 0;
  hipLaunchKernelGGL(tgx::samp_sum_kernel<2>, grid, blk, 0, c->stream, a);
  for (int b = row0; b < row0 + R; b++) {
    tgx::SampPickArgs pa{};
    pa.s = a;
    pa.s.logits = c->rows[(size_t)b].logits; pa.s.part_val = c->rows[(size_t)b].part_val; pa.s.sc = c->samp_scratch;   // the pick kernel indexes sc by fin.row
    pa.s.logits_stride = 0; pa.s.part_stride = 0;
    pa.nwg = nwg; pa.seed = c->seed_dev;
    pa.fin = make_finalize_args(c, b, advance_pos, log_step);
    TGX_DT_SWITCH(c->dt, hipLaunchKernelGGL(tgx::samp_pick_kernel<DT>, dim3(1), dim3(tgx::SAMP_WG), 0, c->stream, pa))
  }
}

// One decode step for all active rows: layers at pos, lm_head, then {sample, pos+=1, next embedding}.
// == nextToken = genNextToken(nextToken)  (GPTEngine.cpp:94-99,165-168)
void launch_decode_step_mfma(tgx_ctx* c, int row0, int M, const tgx_sampler_cfg& cfg);
// Context from which the MFMA decode attention (kernels/attn_decode_mfma.h) beats the VALU kernel, measured per geometry class
// (profiles/r02_attn_long.txt): head_dim 64 with 8 kv heads from ~6k keys (Llama-3.2-1B: 11.0 -> 9.9 µs per layer at 6k, 23.3 -> 18.3 at 30k);
// two kv heads (Qwen2.5-0.5B: few workgroups) and head_dim 128 (Mistral-7B, Llama-3.2-3B) from ~14k (Mistral-7B: 25.3 -> 21.4 at 16k, 39.5 -> 31.4 at 30k).
static int attn_mfma_threshold(const tgx_ctx* c) {
  if (c->attn_mfma_min >= 0) return c->attn_mfma_min;
  return (c->d.head_dim == 64 && c->d.kv_heads >= 8) ? 6000 : 14000;
}
bool decode_mfma_ok(const tgx_ctx* c);
// The attention form of the launches about to be issued / captured, from the context the call ends at: direct (one workgroup per head, no
// combine) for short contexts, the MFMA decode attention for long ones, the VALU split form in between.  One place for all callers
// (the prefill-by-steps branch used to leave attn_mfma at whatever the previous decode call had chosen).
void update_attn_modes(tgx_ctx* c, int n_positions, int rows_per_launch = -1) {   // rows_per_launch: batch rows that share an attention launch (-1: the batch)
  // batches (round 3): the rows themselves fill the chip, so the one-workgroup-per-(kv head, row) form stays ahead of the split form far beyond the
  // batch-1 crossover — Llama-3.2-1B at context 2k: B = 8 1.250 -> 1.105 ms/step, B = 32 2.360 -> 1.680; Mistral-7B at 600: B = 32 6.91 -> 5.47
  const int rpl = rows_per_launch < 0 ? c->batch : rows_per_launch;
  const long long direct_lim = (long long)c->attn_direct_max * (rpl >= 4 ? rpl : 1);
  c->attn_direct = c->past + n_positions <= direct_lim;
  c->attn_nw4 = c->attn_direct && rpl < 4 && c->attn_direct_nw4 > 0 && c->past + n_positions <= c->attn_direct_nw4;
  c->attn_mfma = !c->attn_direct && c->past >= attn_mfma_threshold(c) && c->dt != tgx::DT_F32 && !(c->d.qk_norm && c->d.head_dim == 128 && c->qk_fuse);
}

void launch_decode_step(tgx_ctx* c, const tgx_sampler_cfg& cfg) {
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
    const bool fuse = c->lm_fuse && is_greedy(&cfg);     // greedy: argmax + token publish ride in the lm_head launch (arrival ticket)
    launch_lm_head(c, row0, R, fuse);
    if (!fuse) launch_sample(c, row0, R, cfg, /*advance_pos=*/true, /*log_step=*/true);
    row0 += R;
  }
}


// ---- batched decode on the matrix cores (kernels/skinny.h) ---------------------------------------------------------------------------
// the (epilogue, terms, activation source) combinations the batched step uses; every one exists for 2 dtypes x MB 1,2 x NBW 1,2
#define TGX_SKINNY_COMBOS(X)                                                                                                   \
  X(tgx::GEMM_PARTIAL, 3, 2) X(tgx::GEMM_STORE, 3, 2) X(tgx::GEMM_PARTIAL, 2, 2) X(tgx::GEMM_STORE, 2, 2) X(tgx::GEMM_SILU, 2, 2) \
  X(tgx::GEMM_PARTIAL, 2, 1) X(tgx::GEMM_RESIDUAL, 2, 1) X(tgx::GEMM_PARTIAL, 2, 0) X(tgx::GEMM_RESIDUAL, 2, 0) X(tgx::GEMM_SILU, 2, 0) X(tgx::GEMM_STORE, 2, 0) \
  X(tgx::GEMM_PARTIAL, 3, 0) X(tgx::GEMM_STORE, 3, 0)

template <int DT, int EPI, int NT, int ASRC>
int skinny_set_attr_dt(tgx_ctx* c) {
#define TGX_SK_A(MB_, CFG_) HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::skinny_gemm_kernel<DT, EPI, MB_, NT, CFG_, ASRC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::skinny_lds_bytes(MB_, NT, CFG_)));
  TGX_SK_A(1, 0) TGX_SK_A(1, 1) TGX_SK_A(1, 2) TGX_SK_A(2, 0) TGX_SK_A(2, 1) TGX_SK_A(2, 2)
  if constexpr (ASRC == 0) { TGX_SK_A(4, 0) TGX_SK_A(4, 1) TGX_SK_A(4, 2) }
  if constexpr (ASRC == 1) { TGX_SK_A(4, 1) }
#undef TGX_SK_A
  return TGX_OK;
}
// the skinny GEMM's LDS image (weight tiles + activation panels) exceeds the 64 KB default for 32 rows
int skinny_set_attrs(tgx_ctx* c) {
  int rc;
#define X(E, N, A) if ((rc = skinny_set_attr_dt<tgx::DT_BF16, E, N, A>(c)) || (rc = skinny_set_attr_dt<tgx::DT_F16, E, N, A>(c))) return rc;
  TGX_SKINNY_COMBOS(X)
#undef X
  return TGX_OK;
}

template <int EPI, int NT, int ASRC>
void skinny_dispatch(tgx_ctx* c, dim3 grid, int mb, int cfg, const tgx::GemmArgs& g) {
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
void skinny_dma_dispatch(tgx_ctx* c, dim3 grid, int mb, int nbw, const tgx::GemmArgs& g) {
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
int skinny_dma_set_attr_dt(tgx_ctx* c) {
#define TGX_SKD_A(MB_, NBW_) HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::skinny_dma_kernel<DT, EPI, MB_, NBW_, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::skd_lds_bytes(MB_, NBW_, NT)));
  TGX_SKD_A(1, 1) TGX_SKD_A(2, 1) TGX_SKD_A(4, 1) TGX_SKD_A(1, 2) TGX_SKD_A(2, 2) TGX_SKD_A(4, 2) TGX_SKD_A(8, 1)
#undef TGX_SKD_A
  return TGX_OK;
}
int skinny_dma_set_attrs(tgx_ctx* c) {
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
int launch_skinny(tgx_ctx* c, const SkinnyCall& k) {
  tgx::GemmArgs g{};
  g.A_hi = k.a_hi; g.A_lo = k.a_lo; g.A_lo2 = k.a_lo2; g.A_f32 = k.a_f32; g.lda = k.lda;
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
void launch_reduce_rows(tgx_ctx* c, int epi, int nsplit, const ebyte* bias, float* C, int ldc, int M, int N, float* ssq_out) {
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
bool ksplit_ok(const tgx_ctx* c, int M, int N, int K) {
  // 17-32 rows (two activation blocks, two weight slots), ms/step panel / K-split kernel (round 3 closing build): Llama-3.2-1B (K = 2048) B = 17 1.314 / 1.253,
  // 24 1.275 / 1.237, 32 1.316 / 1.312; Llama-3.2-3B (K = 3072) 2.834 / 2.853, 2.891 / 2.984, 3.015 / 3.199; Mistral-7B (K = 4096) B = 32 5.11 / 5.38: at K = 2048
  // only (option value 2: always)
  const int max_rows = c->skinny_ksplit >= 2 || K == 2048 ? 32 : 16;
  return c->skinny_ksplit && M <= max_rows && K % 256 == 0 && K >= 768 && N >= 64 * c->num_cus;
}
void launch_ksplit(tgx_ctx* c, int epi, const ebyte* W, float* C, int ldc, int M, int N, int K) {
  tgx::GemmArgs g{};
  g.A_hi = c->ws_ah; g.A_lo = c->ws_al; g.B = reinterpret_cast<const bf16_t*>(W); g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.inter = N / 2; g.out_hi = c->ws_hh; g.out_lo = c->ws_hl;
  const dim3 grid((N + 63) / 64), blk(256);
#define TGX_KS(E_, K_) do { if (M > 16) hipLaunchKernelGGL((tgx::skinny_ksplit_kernel<DT, E_, K_, 2>), grid, blk, 0, c->stream, g); \
                            else hipLaunchKernelGGL((tgx::skinny_ksplit_kernel<DT, E_, K_, 1>), grid, blk, 0, c->stream, g); } while (0)
#define TGX_KS_K(E_) do { if (K == 2048) TGX_KS(E_, 2048); else if (K == 3072) TGX_KS(E_, 3072); else if (K == 4096) TGX_KS(E_, 4096); else TGX_KS(E_, 0); } while (0)
  TGX_DT16_SWITCH(c->dt, if (epi == tgx::GEMM_SILU) TGX_KS_K(tgx::GEMM_SILU); else TGX_KS_K(tgx::GEMM_STORE);)
#undef TGX_KS_K
#undef TGX_KS
}
// RMSNorm of the rows of x into 16-bit terms (ws_ah / ws_al), first adding a pending split-K residual (nsplit > 1: the slabs in ws_part)
void launch_norm_terms(tgx_ctx* c, float* x, const ebyte* norm_w, int M, int H, int nsplit, bool third = false) {
  TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::rmsnorm_split_kernel<DT>, dim3(M), dim3(256), 0, c->stream, x, reinterpret_cast<const bf16_t*>(norm_w), c->d.norm_eps, H,
                                          c->ws_ah, c->ws_al, third ? c->ws_al2 : (bf16_t*)nullptr, (const float*)(nsplit > 1 ? c->ws_part : nullptr), nsplit, (long long)M * H, (const bf16_t*)nullptr))
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
  const size_t kv_layer = (size_t)d.kv_heads * d.max_ctx * hd * c->esz;
  RowState& r = c->rows[(size_t)row0];
  const int nt_qkv = c->dt == tgx::DT_BF16 ? 3 : 2;
  float* ssq = c->ws_ssq;
  const bool lm_ks = ksplit_ok(c, M, V, H);
  // 33-64 rows (round 3): four activation blocks; every RMSNorm-fused product takes its activations as 16-bit terms prepared once per product by the
  // row-wise launch that also adds the pending split-K residual (the RMSNorm-on-the-way staging runs out of registers at four blocks)
  const bool terms = M > 32 || ((c->skinny_terms >= 2 || (c->skinny_dma && c->skinny_dma_qkv && M >= c->skinny_dma_rows && H % 64 == 0)) && M > (c->skinny_dma_qkv >= 2 ? 4 : 16));
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
      a.kv_stride = (long long)c->kv_row_elems; a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.pos = r.pos;
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
      a.q_stride = qd; a.kv_stride = (long long)c->kv_row_elems; a.part_stride = (long long)c->attn_part_row; a.dbg = c->debug_attn;
      if (raw_fuse) {
        if (qs > 1) { a.raw_part = c->ws_part; a.raw_nsplit = qs; a.raw_bias = w.bqkv; } else a.raw_qkv = c->ws_out;
        a.raw_rows = M; a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.eps = d.norm_eps;
        a.q_norm_w = d.qk_norm ? w.q_norm : nullptr; a.k_norm_w = d.qk_norm ? w.k_norm : nullptr;
      }
      if (attn_terms) { a.out_hi = c->ws_ah; a.out_lo = c->ws_al; }
      const int fold = c->attn_fold; c->attn_fold = 0;          // the o_proj product here reads the merged output
      launch_attn(c, a, M);
      c->attn_fold = fold;
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
    if (gs > 1) {                                                          // slabs -> siluMul -> terms (z-ordered sums)
      tgx::GemmArgs g{};
      g.part = c->ws_part; g.nsplit = gs; g.M = M; g.N = 2 * I; g.inter = I; g.out_hi = c->ws_hh; g.out_lo = c->ws_hl;
      const dim3 rg((unsigned)(((size_t)M * I + 255) / 256));
      TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::gemm_splitk_reduce_kernel<DT, tgx::GEMM_SILU>), rg, dim3(256), 0, c->stream, g))
    }
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
  hipLaunchKernelGGL(tgx::argmax_partials_rows_kernel, dim3(c->lm_grid, M), dim3(256), 0, c->stream, (const float*)r.logits, (long long)V, V, r.part_val, r.part_idx, (long long)c->lm_grid);
  if (is_greedy(&cfg)) {
    tgx::FinalizeRowsArgs fa{};
    fa.f = make_finalize_args(c, row0, /*advance_pos=*/true, /*log_step=*/true);
    fa.part_stride = c->lm_grid; fa.x_stride = H;
    // rows are finalized concurrently: the step counter moves afterwards, once, when this group holds the batch's last row
    TGX_DT_SWITCH(c->dt, hipLaunchKernelGGL(tgx::finalize_rows_kernel<DT>, dim3(M), dim3(256), 0, c->stream, fa))
    if (row0 + M == c->batch) hipLaunchKernelGGL(tgx::bump_step_kernel, dim3(1), dim3(64), 0, c->stream, c->step);
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
  const size_t kv_layer = (size_t)d.kv_heads * d.max_ctx * hd;
  const int M = NB * S, nq = qd + 2 * kvd;
  const int nt_qkv = c->dt == tgx::DT_BF16 ? 3 : 2;
  float* ssq = c->ws_ssq;
  TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::embed_rows_kernel<DT>, dim3(M), dim3(256), 0, c->stream, (const long long*)c->rows[(size_t)row0].prompt, (const bf16_t*)c->embed, c->ws_x, H, S, (long long)d.max_ctx))
  // 33-64 rows (four activation blocks): RMSNorm + the 16-bit terms once per product in a row-wise launch (which also takes the pending split-K
  // residual), the panel kernel stages stored terms — its RMSNorm-on-the-way form runs out of registers at four blocks
  const bool terms = M > 32;
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
    if (qs > 1) launch_reduce_rows(c, tgx::GEMM_STORE, qs, w.bqkv, c->ws_out, nq, M, nq, nullptr);
    for (int b = 0; b < NB; b++) {
      RowState& r = c->rows[(size_t)(row0 + b)];
      const size_t ro = (size_t)b * S;
      tgx::RopeKvArgs a{};
      a.QKV = c->ws_out + ro * nq; a.q_hi = c->ws_qh + ro * qd; a.q_lo = c->ws_ql + ro * qd;
      a.k_cache = reinterpret_cast<bf16_t*>(r.kcache) + (size_t)l * kv_layer; a.v_cache = reinterpret_cast<bf16_t*>(r.vcache) + (size_t)l * kv_layer;
      a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin;
      a.heads = d.heads; a.kv_heads = d.kv_heads; a.hd = hd; a.max_ctx = d.max_ctx; a.past = (int)c->past;
      a.q_norm_w = d.qk_norm ? (const bf16_t*)w.q_norm : nullptr; a.k_norm_w = d.qk_norm ? (const bf16_t*)w.k_norm : nullptr; a.eps = d.norm_eps;
      TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL(tgx::rope_kv_split_kernel<DT>, dim3(S), dim3(256), 0, c->stream, a))
    }
    for (int b = 0; b < NB; b++) {
      RowState& r = c->rows[(size_t)(row0 + b)];
      const size_t ro = (size_t)b * S;
      tgx::AttnPrefillArgs a{};
      a.q_hi = c->ws_qh + ro * qd; a.q_lo = c->ws_ql + ro * qd;
      a.k_cache = reinterpret_cast<bf16_t*>(r.kcache) + (size_t)l * kv_layer; a.v_cache = reinterpret_cast<bf16_t*>(r.vcache) + (size_t)l * kv_layer;
      a.o_hi = c->ws_ah + ro * qd; a.o_lo = c->ws_al + ro * qd; a.S = S; a.heads = d.heads; a.kv_heads = d.kv_heads; a.max_ctx = d.max_ctx; a.past = (int)c->past;
      a.scale = 1.0f / sqrtf((float)hd); a.qblk_mirror = c->attn_mirror;
      const dim3 grid((S + 127) / 128, d.heads), blk(256);
      TGX_DT16_SWITCH(c->dt, if (hd == 64) hipLaunchKernelGGL((tgx::attn_prefill_kernel<DT, 64>), grid, blk, 0, c->stream, a);
                             else hipLaunchKernelGGL((tgx::attn_prefill_kernel<DT, 128, 1>), grid, blk, 0, c->stream, a))
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
    if (gs > 1) {
      tgx::GemmArgs g{};
      g.part = c->ws_part; g.nsplit = gs; g.M = M; g.N = 2 * I; g.inter = I; g.out_hi = c->ws_hh; g.out_lo = c->ws_hl;
      const dim3 rg((unsigned)(((size_t)M * I + 255) / 256));
      TGX_DT16_SWITCH(c->dt, hipLaunchKernelGGL((tgx::gemm_splitk_reduce_kernel<DT, tgx::GEMM_SILU>), rg, dim3(256), 0, c->stream, g))
    }
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


// ---- batched prefill for fp32 storage (kernels/gemm_f32.h): every product on v_mfma_f32_32x32x2_f32 (exact fp32 products), the
// row-wise ops in fp32, attention = the decode attention kernel with the prompt positions as its rows (they share the sequence's
// cache: kv_stride 0, row r attends the keys up to past + r).  All families incl. GPT-2 — BASELINE.json configs[0] is GPT-2 fp32.
void launch_gemm_f32(tgx_ctx* c, int epi, const ebyte* B, const ebyte* bias, const float* A, float* C, int M, int N, int K, int ldc) {
  tgx::GemmF32Args g{};
  g.A = A; g.B = reinterpret_cast<const float*>(B); g.bias = reinterpret_cast<const float*>(bias); g.C = C; g.M = M; g.N = N; g.K = K; g.ldc = ldc; g.inter = N / 2;
  const bool few = ((N + tgx::FBN - 1) / tgx::FBN) * ((M + 127) / 128) < 2 * c->num_cus;
  const bool small = epi == tgx::F32_SILU ? false : few;
  const int tm = small ? 64 : 128;
  dim3 grid((N + tgx::FBN - 1) / tgx::FBN, (M + tm - 1) / tm), blk(256);
  // few tiles (a short prompt, or N = hidden): split K over blockIdx.z until ~2 workgroups per CU exist; slabs summed in z order
  const int ntiles = (int)(grid.x * grid.y), ksteps = (K + tgx::FBK - 1) / tgx::FBK;
  int nsplit = 1;
  if (c->gemm_splitk && ntiles < 2 * c->num_cus) nsplit = std::max(1, std::min(std::min(16, ksteps / 4), (2 * c->num_cus + ntiles - 1) / ntiles));
  if (nsplit > 1 && (size_t)nsplit * M * N * 4 > c->ws_part_bytes) nsplit = 1;      // sized in tgx_forward (ensure_f32_part)
  if (nsplit > 1) {
    g.k_per = ((ksteps + nsplit - 1) / nsplit) * tgx::FBK;
    nsplit = (K + g.k_per - 1) / g.k_per;
  }
  g.part = c->ws_part; g.nsplit = nsplit;
  grid.z = nsplit > 1 ? nsplit : 1;
  switch (epi) {
    case tgx::F32_SILU: hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_SILU, 2>), grid, blk, 0, c->stream, g); break;
    case tgx::F32_GELU: if (small) hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_GELU, 1>), grid, blk, 0, c->stream, g);
                        else hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_GELU, 2>), grid, blk, 0, c->stream, g); break;
    case tgx::F32_RESIDUAL: if (small) hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_RESIDUAL, 1>), grid, blk, 0, c->stream, g);
                            else hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_RESIDUAL, 2>), grid, blk, 0, c->stream, g); break;
    default: if (small) hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_STORE, 1>), grid, blk, 0, c->stream, g);
             else hipLaunchKernelGGL((tgx::gemm_f32_kernel<tgx::F32_STORE, 2>), grid, blk, 0, c->stream, g); break;
  }
  if (nsplit > 1) {
    const size_t nout = (size_t)M * (epi == tgx::F32_SILU ? N / 2 : N);
    const dim3 rg((unsigned)((nout + 255) / 256));
    switch (epi) {
      case tgx::F32_SILU: hipLaunchKernelGGL((tgx::gemm_f32_reduce_kernel<tgx::F32_SILU>), rg, blk, 0, c->stream, g); break;
      case tgx::F32_GELU: hipLaunchKernelGGL((tgx::gemm_f32_reduce_kernel<tgx::F32_GELU>), rg, blk, 0, c->stream, g); break;
      case tgx::F32_RESIDUAL: hipLaunchKernelGGL((tgx::gemm_f32_reduce_kernel<tgx::F32_RESIDUAL>), rg, blk, 0, c->stream, g); break;
      default: hipLaunchKernelGGL((tgx::gemm_f32_reduce_kernel<tgx::F32_STORE>), rg, blk, 0, c->stream, g); break;
    }
  }
}

// split-K slabs of the fp32 products: up to 16 splits of the widest [rows][N] output
int ensure_f32_part(tgx_ctx* c, int rows) {
  const tgx_model_desc& d = c->d;
  const size_t widest = std::max<size_t>(std::max<size_t>((size_t)d.heads * d.head_dim + 2 * (size_t)d.kv_heads * d.head_dim, (size_t)(c->gpt2 ? 1 : 2) * d.inter), (size_t)d.hidden);
  // splits shrink as the tile count grows: nsplit * tiles stays near two per CU, so nsplit * rows * N is bounded by ~2 CUs x one 128 x 128 tile x 16
  const size_t need = std::min<size_t>((size_t)16 * rows * widest * 4, (size_t)64 << 20);
  if (need > c->ws_part_bytes) {
    drop_step_graphs(c);
    HIP_OK(c, hipStreamSynchronize(c->stream));
    if (c->ws_part) (void)hipFree(c->ws_part);
    c->ws_part = nullptr; c->ws_part_bytes = 0;
    HIP_OK(c, hipMalloc((void**)&c->ws_part, need));
    c->ws_part_bytes = need;
  }
  return TGX_OK;
}

void launch_prefill_f32(tgx_ctx* c, int row0, int NB, int S) {
  const tgx_model_desc& d = c->d;
  const int H = d.hidden, I = d.inter, hd = d.head_dim, qd = d.heads * hd, kvd = d.kv_heads * hd, nq = qd + 2 * kvd;
  const size_t kv_layer = (size_t)d.kv_heads * d.max_ctx * hd * c->esz;
  const int M = NB * S;
  float* xn = reinterpret_cast<float*>(c->ws_ah);      // [M][max(H, qd)]: normalised rows, later the attention output
  float* qrows = reinterpret_cast<float*>(c->ws_qh);   // [M][qd] rotated queries
  float* hrows = reinterpret_cast<float*>(c->ws_hh);   // [M][I]
  hipLaunchKernelGGL((tgx::embed_rows_any_kernel<tgx::DT_F32>), dim3(M), dim3(256), 0, c->stream, (const long long*)c->rows[(size_t)row0].prompt, (const void*)c->embed, (const void*)(c->gpt2 ? c->wpe : nullptr), c->ws_x, H, S, (long long)d.max_ctx, (int)c->past);
  hipLaunchKernelGGL(tgx::iota_pos_kernel, dim3((S + 255) / 256), dim3(256), 0, c->stream, c->ws_pos, (int)c->past, S);
  auto norm = [&](const ebyte* w, const ebyte* b) {
    if (c->gpt2) hipLaunchKernelGGL((tgx::norm_rows_kernel<tgx::DT_F32, 1, 0>), dim3(M), dim3(256), 0, c->stream, (const float*)c->ws_x, (const void*)w, (const void*)b, d.norm_eps, H, xn, (bf16_t*)nullptr, (bf16_t*)nullptr, (bf16_t*)nullptr);
    else hipLaunchKernelGGL((tgx::norm_rows_kernel<tgx::DT_F32, 0, 0>), dim3(M), dim3(256), 0, c->stream, (const float*)c->ws_x, (const void*)w, (const void*)nullptr, d.norm_eps, H, xn, (bf16_t*)nullptr, (bf16_t*)nullptr, (bf16_t*)nullptr);
  };
  c->attn_direct = c->past + S <= c->attn_direct_max;
  for (int l = 0; l < d.layers; l++) {
    const LayerW& w = c->L[(size_t)l];
    norm(w.in_norm, w.in_norm_b);
    launch_gemm_f32(c, tgx::F32_STORE, w.wqkv, w.bqkv, xn, c->ws_out, M, nq, H, nq);
    for (int b = 0; b < NB; b++) {
      RowState& r = c->rows[(size_t)(row0 + b)];
      const size_t ro = (size_t)b * S;
      tgx::RopeRowsArgs a{};
      a.QKV = c->ws_out + ro * nq; a.rows = S; a.q_out = qrows + ro * qd; a.q_stride = qd;
      a.k_cache = r.kcache + (size_t)l * kv_layer; a.v_cache = r.vcache + (size_t)l * kv_layer; a.kv_stride = 0;     // the rows are positions of ONE sequence
      a.rope_cos = c->rope_cos; a.rope_sin = c->rope_sin; a.pos = c->ws_pos;
      a.heads = d.heads; a.kv_heads = d.kv_heads; a.hd = hd; a.max_ctx = d.max_ctx;
      a.q_norm_w = d.qk_norm ? w.q_norm : nullptr; a.k_norm_w = d.qk_norm ? w.k_norm : nullptr; a.eps = d.norm_eps;
      hipLaunchKernelGGL((tgx::rope_kv_rows_kernel<tgx::DT_F32>), dim3(S, d.heads + 2 * d.kv_heads), dim3(64), 0, c->stream, a);
    }
    for (int b = 0; b < NB; b++) {
      RowState& r = c->rows[(size_t)(row0 + b)];
      const size_t ro = (size_t)b * S;
      if (c->f32_flash) {        // causal flash attention on the f32-input MFMA (K / V tiles shared by 128 queries)
        tgx::AttnPrefillF32Args a{};
        a.q = qrows + ro * qd; a.k_cache = reinterpret_cast<const float*>(r.kcache + (size_t)l * kv_layer); a.v_cache = reinterpret_cast<const float*>(r.vcache + (size_t)l * kv_layer);
        a.out = xn + ro * qd; a.S = S; a.heads = d.heads; a.kv_heads = d.kv_heads; a.max_ctx = d.max_ctx; a.past = (int)c->past;
        a.scale = 1.0f / sqrtf((float)hd); a.qblk_mirror = c->attn_mirror;
        const dim3 grid((S + 127) / 128, d.heads), blk(256);
        if (hd == 64) hipLaunchKernelGGL((tgx::attn_prefill_f32_kernel<64>), grid, blk, 0, c->stream, a);
        else hipLaunchKernelGGL((tgx::attn_prefill_f32_kernel<128>), grid, blk, 0, c->stream, a);
        continue;
      }
      for (int s0 = 0; s0 < S; s0 += F32_ATTN_ROWS) {       // option prefill.f32_flash = 0: the decode attention kernel, blocks of rows
        const int R = std::min(F32_ATTN_ROWS, S - s0);
        tgx::AttnArgs a{};
        a.q = qrows + (ro + s0) * qd; a.k_cache = r.kcache + (size_t)l * kv_layer; a.v_cache = r.vcache + (size_t)l * kv_layer;
        a.pos = c->ws_pos + s0; a.part = c->ws_attn_part; a.out = xn + (ro + s0) * qd;
        a.heads = d.heads; a.kv_heads = d.kv_heads; a.max_ctx = d.max_ctx; a.nsplit = c->attn_nsplit;
        a.scale = 1.0f / sqrtf((float)hd);
        a.q_stride = qd; a.kv_stride = 0; a.part_stride = (long long)c->attn_part_row; a.dbg = c->debug_attn;
        const int fold = c->attn_fold; c->attn_fold = 0;
        launch_attn(c, a, R);
        c->attn_fold = fold;
      }
    }
    launch_gemm_f32(c, tgx::F32_RESIDUAL, w.wo, w.bo, xn, c->ws_x, M, H, qd, H);
    norm(w.post_norm, w.post_norm_b);
    if (c->gpt2) launch_gemm_f32(c, tgx::F32_GELU, w.wgu, w.bfc, xn, hrows, M, I, H, I);
    else launch_gemm_f32(c, tgx::F32_SILU, w.wgu, nullptr, xn, hrows, M, 2 * I, H, I);
    launch_gemm_f32(c, tgx::F32_RESIDUAL, w.wdown, w.bdown, hrows, c->ws_x, M, H, I, H);
  }
  for (int b = 0; b < NB; b++)
    (void)hipMemcpyAsync(c->rows[(size_t)(row0 + b)].x, c->ws_x + ((size_t)(b + 1) * S - 1) * H, (size_t)H * 4, hipMemcpyDeviceToDevice, c->stream);
}

bool same_cfg(const tgx_sampler_cfg& a, const tgx_sampler_cfg& b) {
  return a.temperature == b.temperature && a.top_k == b.top_k && a.top_p == b.top_p && a.min_p == b.min_p;
}

// The decode step as a hipGraph, captured once per (batch, sampler config): `steps` consecutive steps per graph — token,
// position and step counter live on the device, so a multi-step graph is the same launch sequence repeated.
int capture_steps(tgx_ctx* c, const tgx_sampler_cfg& cfg, int steps, hipGraphExec_t* out) {
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

int ensure_step_graph(tgx_ctx* c, const tgx_sampler_cfg& cfg, bool want_multi) {
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
    if (g.step || g.multi) {      // evict the least recently used set (its replays may still be in flight)
      HIP_OK(c, hipStreamSynchronize(c->stream));
      if (g.step) (void)hipGraphExecDestroy(g.step);
      if (g.multi) (void)hipGraphExecDestroy(g.multi);
      g = tgx_ctx::GraphSet{};
    }
    int rc = capture_steps(c, cfg, 1, &g.step);
    if (rc) return rc;
    g.batch = c->batch; g.cfg = cfg; g.direct = c->attn_direct; g.mfma = c->attn_mfma; g.nw4 = c->attn_nw4;
    hit = victim;
  }
  tgx_ctx::GraphSet& g = c->graph_cache[hit];
  if (want_multi && !g.multi && c->graph_steps > 1) { int rc = capture_steps(c, cfg, c->graph_steps, &g.multi); if (rc) return rc; }
  g.used = ++c->graph_clock;
  c->graph_cur = hit; c->step_graph = g.step; c->multi_graph = g.multi;
  return TGX_OK;
}

int run_decode_steps(tgx_ctx* c, const tgx_sampler_cfg& cfg, uint64_t seed, int n) {
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
    c->have_probs = true;
  }
  if (decode_mfma_ok(c)) {   // the batched step's workspace must exist before the step is captured
    int rc = ensure_skinny_ws(c, std::min(c->decode_step_rows, c->batch));
    if (rc) return rc;
  }
  // The attention form depends on the context (four-wave direct / sixteen-wave direct / split + combine / matrix cores): a call that crosses a limit is
  // issued in chunks, each on the form of its own contexts, from the cache of captured graphs
  int remaining = n;
  while (remaining > 0) {
    int m = remaining;
    const long long lims[2] = {c->batch < 4 ? (long long)c->attn_direct_nw4 : 0LL, (long long)c->attn_direct_max * (c->batch >= 4 ? c->batch : 1)};
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
      LAUNCH_OK(c);
    }
    c->past += m;
    c->steps_issued += m;
    remaining -= m;
  }
  return TGX_OK;
}

template <typename T>
int dev_alloc(tgx_ctx* c, T** p, size_t n) {
  HIP_OK(c, hipMalloc((void**)p, n * sizeof(T)));
  return TGX_OK;
}

}  // namespace

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
  c->attn_direct_max = d.head_dim == 64 ? 768 : 384;
  // four waves per head up to 256 keys at head_dim 64 (Qwen2.5-0.5B 16-token prompt 0.579 -> 0.565 ms/token, Llama-3.2-1B 0.648 -> 0.634; from ~256 keys and at head_dim 128 the sixteen-wave form is ahead)
  c->attn_direct_nw4 = d.head_dim == 64 ? 256 : 0;
  // very short prompts: ONE pass through the batched decode kernels (4 positions) still beats the skinny MFMA prefill on small models
  // (Llama-3.2-1B: S = 4 1.00 vs 1.07 ms, S = 5 1.55 vs 1.07; Mistral-7B S = 4 4.65 vs 4.09) — tools/prefill_crossover.py, profiles/r02_prefill_short.txt
  c->prefill_min_rows = d.hidden > 2048 ? 4 : 5;
  c->tune[TGX_KERNEL_DOWN].ks = 4;   // K = intermediate_size: 4 waves split each row pair
  // qkv is the most latency-bound launch (few rows): 4 waves per row pair shorten every wave's load -> reduce chain; measured
  // ks 1 -> 4: Llama-3.2-1B 1395 -> 1411 tok/s, 3B 628 -> 637, Mistral-7B 341 -> 347; hidden 896 (Qwen2.5-0.5B) loses 2 %
  c->tune[TGX_KERNEL_QKV].ks = d.hidden >= 2048 ? 4 : 1;

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
  const size_t kv_elems = (size_t)d.layers * d.kv_heads * d.max_ctx * hd;
  c->kv_row_elems = kv_elems;
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
  if ((rc = dev_alloc(c, &c->slab_k, B * kv_elems * c->esz))) return rc;
  if ((rc = dev_alloc(c, &c->slab_v, B * kv_elems * c->esz))) return rc;
  HIP_OK(c, hipMemset(c->slab_tok, 0, B * 4));
  HIP_OK(c, hipMemset(c->slab_pos, 0, B * 4));
  HIP_OK(c, hipMemset(c->slab_k, 0, B * kv_elems * c->esz));
  HIP_OK(c, hipMemset(c->slab_v, 0, B * kv_elems * c->esz));
  for (size_t b = 0; b < B; b++) {
    RowState& r = c->rows[b];
    r.x = c->slab_x + b * H; r.q = c->slab_q + b * qd; r.k_raw = c->slab_kraw + b * kvd; r.attn = c->slab_attn + b * qd;
    r.h = c->slab_h + b * I; r.logits = c->slab_logits + b * V; r.probs = c->slab_probs + b * V;
    r.part_val = c->slab_part_val + b * c->lm_grid; r.part_idx = c->slab_part_idx + b * c->lm_grid;
    r.attn_part = c->slab_attn_part + b * c->attn_part_row;
    r.tok = c->slab_tok + b; r.pos = c->slab_pos + b; r.prompt = c->slab_prompt + b * d.max_ctx;
    r.kcache = c->slab_k + b * kv_elems * c->esz; r.vcache = c->slab_v + b * kv_elems * c->esz;
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
  if ((rc = dev_alloc(c, &c->step, 1))) return rc;
  if ((rc = dev_alloc(c, &c->seed_dev, 1))) return rc;
  if ((rc = dev_alloc(c, &c->samp_scratch, (size_t)d.max_batch))) return rc;
  HIP_OK(c, hipMemset(c->samp_scratch, 0, sizeof(tgx::SampScratch) * (size_t)d.max_batch));
  if ((V + tgx::SAMP_TILE - 1) / tgx::SAMP_TILE > tgx::SAMP_MAX_WG) return set_err(c, TGX_ERR_UNSUPPORTED, "vocabulary %d exceeds the sampler's %d entries", V, tgx::SAMP_MAX_WG * tgx::SAMP_TILE);
  if ((rc = dev_alloc(c, &c->nop_word, 1))) return rc;
  if ((rc = dev_alloc(c, &c->lm_ticket, 1))) return rc;
  if ((rc = dev_alloc(c, &c->fin_dev, 4))) return rc;
  HIP_OK(c, hipMemset(c->lm_ticket, 0, 4));
  if ((rc = dev_alloc(c, &c->scratch_x, (size_t)H))) return rc;
  HIP_OK(c, hipMemset(c->scratch_x, 0, (size_t)H * 4));
  if ((rc = dev_alloc(c, &c->tok_log, (size_t)c->log_cap * d.max_batch))) return rc;
  HIP_OK(c, hipMemset(c->step, 0, 4));
  HIP_OK(c, hipHostMalloc((void**)&c->host_ring, (size_t)HOST_RING * d.max_batch * 4, hipHostMallocMapped));
  HIP_OK(c, hipHostGetDevicePointer((void**)&c->host_ring_dev, c->host_ring, 0));
  for (int i = 0; i < MAX_TICKET_EVENTS; i++) HIP_OK(c, hipEventCreateWithFlags(&c->ticket_ev[i], hipEventDisableTiming));
  for (int i = 0; i < 2; i++) HIP_OK(c, hipEventCreate(&c->prof.ev[i]));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_x2_kernel<tgx::DT_BF16, tgx::GEMM_STORE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, tgx::GBM * tgx::GLD * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_x2_kernel<tgx::DT_BF16, tgx::GEMM_RESIDUAL, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, tgx::GBM * tgx::GLD * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_x2_kernel<tgx::DT_F16, tgx::GEMM_STORE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, tgx::GBM * tgx::GLD * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_x2_kernel<tgx::DT_F16, tgx::GEMM_RESIDUAL, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, tgx::GBM * tgx::GLD * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_x2_kernel<tgx::DT_BF16, tgx::GEMM_PARTIAL, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, tgx::GBM * tgx::GLD * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_x2_kernel<tgx::DT_F16, tgx::GEMM_PARTIAL, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, tgx::GBM * tgx::GLD * 2));
  if ((rc = skinny_set_attrs(c))) return rc;
  if ((rc = skinny_dma_set_attrs(c))) return rc;
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_decode_mfma_kernel<tgx::DT_BF16, 64, 8, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::attn_mfma_raw_lds_bytes<64, 8>()));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_decode_mfma_kernel<tgx::DT_F16, 64, 8, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::attn_mfma_raw_lds_bytes<64, 8>()));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_decode_mfma_kernel<tgx::DT_BF16, 128, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::attn_mfma_raw_lds_bytes<128>()));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_decode_mfma_kernel<tgx::DT_F16, 128, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::attn_mfma_raw_lds_bytes<128>()));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_decode_mfma_kernel<tgx::DT_BF16, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::attn_mfma_lds_bytes<128>()));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::attn_decode_mfma_kernel<tgx::DT_F16, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tgx::attn_mfma_lds_bytes<128>()));
#define TGX_DMA_ATTR1(DT_, EPI_, MI_, BK_, NS_) HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma_kernel<DT_, EPI_, MI_, BK_, NS_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::min<size_t>(160 * 1024, tgx::gemm_dma_lds_bytes(MI_, true, BK_, NS_))));
#define TGX_DMA_ATTR(DT_, EPI_, MI_) TGX_DMA_ATTR1(DT_, EPI_, MI_, 64, 2) TGX_DMA_ATTR1(DT_, EPI_, MI_, 64, 3) TGX_DMA_ATTR1(DT_, EPI_, MI_, 64, 4) TGX_DMA_ATTR1(DT_, EPI_, MI_, 32, 2) TGX_DMA_ATTR1(DT_, EPI_, MI_, 32, 3) TGX_DMA_ATTR1(DT_, EPI_, MI_, 32, 4)
#define TGX_DMA_ATTR_D(DT_) TGX_DMA_ATTR(DT_, tgx::GEMM_SILU, 2) TGX_DMA_ATTR(DT_, tgx::GEMM_GELU, 2) TGX_DMA_ATTR(DT_, tgx::GEMM_RESIDUAL, 1) TGX_DMA_ATTR(DT_, tgx::GEMM_RESIDUAL, 2) TGX_DMA_ATTR(DT_, tgx::GEMM_STORE, 1) TGX_DMA_ATTR(DT_, tgx::GEMM_STORE, 2)
  TGX_DMA_ATTR_D(tgx::DT_BF16) TGX_DMA_ATTR_D(tgx::DT_F16)
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_BF16, tgx::GEMM_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_F16, tgx::GEMM_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_BF16, tgx::GEMM_RESIDUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_BF16, tgx::GEMM_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_F16, tgx::GEMM_RESIDUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8k_kernel<tgx::DT_F16, tgx::GEMM_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 128 * 64 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_BF16, tgx::GEMM_RESIDUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_BF16, tgx::GEMM_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_F16, tgx::GEMM_RESIDUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_F16, tgx::GEMM_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_BF16, tgx::GEMM_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_BF16, tgx::GEMM_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_F16, tgx::GEMM_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
  HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::gemm_dma8_kernel<tgx::DT_F16, tgx::GEMM_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 3 * 256 * 32 * 2));
#define TGX_DMA_ATTR_P(DT_) TGX_DMA_ATTR1(DT_, tgx::GEMM_PARTIAL, 1, 64, 2) TGX_DMA_ATTR1(DT_, tgx::GEMM_PARTIAL, 2, 64, 2) TGX_DMA_ATTR1(DT_, tgx::GEMM_PARTIAL, 2, 32, 2)
  TGX_DMA_ATTR_P(tgx::DT_BF16) TGX_DMA_ATTR_P(tgx::DT_F16)
#undef TGX_DMA_ATTR_P
#undef TGX_DMA_ATTR_D
#undef TGX_DMA_ATTR
#undef TGX_DMA_ATTR1
  if ((rc = dev_alloc(c, &c->attn_tickets, (size_t)d.max_batch * d.kv_heads * 8))) return rc;
  HIP_OK(c, hipMemset(c->attn_tickets, 0, (size_t)d.max_batch * d.kv_heads * 8 * 4));
  // persistent engine (option engine.mode): granule buffers of its in-launch edges, the tag epoch, the give-up word
  if (!c->gpt2 && c->dt != tgx::DT_F32) {
    if ((rc = dev_alloc(c, &c->eng_gx1, (size_t)H))) return rc;
    if ((rc = dev_alloc(c, &c->eng_gh, (size_t)I))) return rc;
    if ((rc = dev_alloc(c, &c->eng_gx2, (size_t)H))) return rc;
    if ((rc = dev_alloc(c, &c->eng_epoch, 1))) return rc;
    if ((rc = dev_alloc(c, &c->eng_err, 1))) return rc;
    HIP_OK(c, hipMemset(c->eng_gx1, 0, (size_t)H * 8)); HIP_OK(c, hipMemset(c->eng_gh, 0, (size_t)I * 8)); HIP_OK(c, hipMemset(c->eng_gx2, 0, (size_t)H * 8));
    const unsigned one = 1, zero = 0;
    HIP_OK(c, hipMemcpy(c->eng_epoch, &one, 4, hipMemcpyHostToDevice)); HIP_OK(c, hipMemcpy(c->eng_err, &zero, 4, hipMemcpyHostToDevice));
    HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::engine_kernel<tgx::DT_BF16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::engine_kernel<tgx::DT_BF16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::engine_kernel<tgx::DT_F16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_OK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&tgx::engine_kernel<tgx::DT_F16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  c->past = 0;
  c->finalized = true;
  return TGX_OK;
}

void tgx_destroy(tgx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  drop_step_graphs(c);
  auto fr = [](void* p) { if (p) (void)hipFree(p); };
  fr(c->embed); fr(c->lm_head); fr(c->final_norm); fr(c->wpe); fr(c->final_norm_b); fr(c->rope_cos); fr(c->rope_sin); fr(c->step); fr(c->tok_log); fr(c->nop_word); fr(c->lm_ticket); fr(c->fin_dev); fr(c->scratch_x); fr(c->seed_dev); fr(c->samp_scratch);
  fr(c->attn_tickets); fr(c->eng_gx1); fr(c->eng_gh); fr(c->eng_gx2); fr(c->eng_epoch); fr(c->eng_err); fr(c->eng_stats);
  fr(c->ch_x); fr(c->ch_q); fr(c->ch_kraw); fr(c->ch_attn); fr(c->ch_h); fr(c->ch_part); fr(c->ch_pos);
  fr(c->ws_x); fr(c->ws_out); fr(c->ws_ah); fr(c->ws_al); fr(c->ws_al2); fr(c->ws_qh); fr(c->ws_ql); fr(c->ws_hh); fr(c->ws_hl); fr(c->ws_part); fr(c->ws_ssq); fr(c->ws_pos); fr(c->ws_attn_part);
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
  if (seq > 1 && c->past > 0) return set_err(c, TGX_ERR_INVALID, "seq>1 with pastLength>0");
  if (c->past + seq > c->d.max_ctx) return set_err(c, TGX_ERR_CONTEXT, "context size exceeded: %lld + %d > %d", (long long)c->past, seq, c->d.max_ctx);
  for (int64_t i = 0; i < (int64_t)batch * seq; i++)
    if (ids[i] < 0 || ids[i] >= c->d.vocab) return set_err(c, TGX_ERR_INVALID, "token id out of range");
  HIP_OK(c, hipSetDevice(c->device));
  c->batch = batch;
  // matrix-core prefill: 16-bit storage through the split-term GEMMs (every family incl. GPT-2), fp32 storage through the f32-input MFMA
  const bool f32_path = c->dt == tgx::DT_F32 && seq >= c->prefill_f32_min_rows && c->prefill_mfma;
  const bool mfma_path = f32_path || (seq >= c->prefill_min_rows && seq >= 4 && c->prefill_mfma && c->dt != tgx::DT_F32 && prefill_shapes_ok(c->d));
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
      else if (skinny) launch_prefill_skinny(c, row0, nb, seq); else launch_prefill(c, row0, nb, seq);
      for (int b = row0; b < row0 + nb;) {
        const int rem = row0 + nb - b, R = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
        launch_lm_head(c, b, R);
        b += R;
      }
      for (int b = row0; b < row0 + nb; b++) hipLaunchKernelGGL(tgx::add_pos_kernel, dim3(1), dim3(64), 0, c->stream, c->rows[(size_t)b].pos, seq);
    }
  }
  for (int b = 0; b < batch && !mfma_path; b++) {
    RowState& r = c->rows[(size_t)b];
    // prefill by steps (fp32 storage, GPT-2, prompts shorter than 4 tokens, shapes the GEMM tile does not cover): up to 4 consecutive
    // positions per pass through the decode kernels — the chunk rows share this row's cache (kv_stride 0), each attends the
    // keys up to its own position, so the result equals position-by-position passes at a quarter of the weight traffic
    update_attn_modes(c, seq, 1);      // the chunk rows of a pass are positions of ONE sequence
    for (int s0 = 0; s0 < seq;) {
      const int rem = seq - s0, R = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
      tgx::EmbedChunkArgs e{};
      e.ids = r.prompt + s0; e.embed = c->embed; e.x = c->ch_x; e.H = c->d.hidden; e.pos = c->ch_pos; e.pos0 = (int)c->past + s0; e.wpe = c->gpt2 ? c->wpe : nullptr;
      TGX_DT_SWITCH(c->dt, hipLaunchKernelGGL(tgx::embed_chunk_kernel<DT>, dim3(R), dim3(256), 0, c->stream, e))
      for (int k = 0; k < R; k++) { c->chunk[k].kcache = r.kcache; c->chunk[k].vcache = r.vcache; }
      launch_layers(c, c->chunk, R, 0);
      s0 += R;
      if (s0 == seq) {                                   // the last position's hidden state feeds lm_head; publish token and length
        (void)hipMemcpyAsync(r.x, c->chunk[R - 1].x, (size_t)c->d.hidden * 4, hipMemcpyDeviceToDevice, c->stream);
        hipLaunchKernelGGL(tgx::add_pos_kernel, dim3(1), dim3(64), 0, c->stream, r.pos, seq);
        launch_lm_head(c, b, 1);
      }
    }
  }
  HIP_OK(c, hipGetLastError());
  LAUNCH_OK(c);
  HIP_OK(c, hipStreamSynchronize(c->stream));   // host `ids` may be pageable and reused by the caller
  c->past += seq;
  c->have_logits = true;
  c->have_token = false;
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
    c->have_probs = true;
  }
  launch_sample(c, 0, c->batch, *cfg, /*advance_pos=*/false, /*log_step=*/false);
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipStreamSynchronize(c->stream));
  for (int b = 0; b < c->batch; b++) {
    int t = 0;
    HIP_OK(c, hipMemcpy(&t, c->rows[(size_t)b].tok, 4, hipMemcpyDeviceToHost));
    if (out_ids) out_ids[b] = t;
    if (b == 0) c->last_sampled0 = t;
  }
  c->have_token = true;
  return TGX_OK;
}

int tgx_decode(tgx_ctx* c, const tgx_sampler_cfg* cfg, uint64_t seed, int n_steps, int64_t* out_ids) {
  if (!c || !cfg || n_steps < 0) return TGX_ERR_INVALID;
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
  HIP_OK(c, hipStreamSynchronize(c->stream));
  c->past = 0;
  c->have_logits = c->have_token = false;
  return TGX_OK;
}

int64_t tgx_past_length(const tgx_ctx* c) { return c ? c->past : -1; }
int64_t tgx_context_size(const tgx_ctx* c) { return c ? c->d.max_ctx : -1; }
int32_t tgx_num_layers(const tgx_ctx* c) { return c ? c->d.layers : -1; }

int tgx_synchronize(tgx_ctx* c) {
  if (!c) return TGX_ERR_INVALID;
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  if (c->engine_mode && c->eng_err) {   // a bounded spin of the persistent engine gave up (kernels/engine.h): the step's results are invalid
    unsigned e = 0;
    HIP_OK(c, hipMemcpy(&e, c->eng_err, 4, hipMemcpyDeviceToHost));
    if (e) { const unsigned z = 0; (void)hipMemcpy(c->eng_err, &z, 4, hipMemcpyHostToDevice); return set_err(c, TGX_ERR_DEVICE, "decode engine gave up a wait (code 0x%x)", e); }
  }
  return TGX_OK;
}

int tgx_engine_read_stats(tgx_ctx* c, uint64_t* out, int64_t capacity, int32_t* out_layers, int32_t* out_cus, int32_t* out_fields) {
  if (!c || !c->finalized) return TGX_ERR_INVALID;
  if (out_layers) *out_layers = c->d.layers;
  if (out_cus) *out_cus = c->num_cus;
  if (out_fields) *out_fields = tgx::ENG_NSTAT;
  if (!out) return TGX_OK;
  if (!c->eng_stats) return set_err(c, TGX_ERR_STATE, "no engine statistics: set option engine.stats = 1 first");
  const int64_t n = (int64_t)c->d.layers * c->num_cus * tgx::ENG_NSTAT;
  if (capacity < n) return set_err(c, TGX_ERR_INVALID, "engine statistics need %lld words", (long long)n);
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  HIP_OK(c, hipMemcpy(out, c->eng_stats, (size_t)n * 8, hipMemcpyDeviceToHost));
  return TGX_OK;
}

int tgx_read_kv(tgx_ctx* c, int row, int layer, float* k_out, float* v_out) {
  if (!c || !c->finalized || row < 0 || row >= c->d.max_batch || layer < 0 || layer >= c->d.layers) return TGX_ERR_INVALID;
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  const tgx_model_desc& d = c->d;
  const size_t hd = (size_t)d.head_dim, per_head = (size_t)d.max_ctx * hd, T = (size_t)c->past;
  std::vector<unsigned char> tmp(per_head * c->esz);
  for (int which = 0; which < 2; which++) {
    float* out = which ? v_out : k_out;
    if (!out) continue;
    const ebyte* base = (which ? c->rows[(size_t)row].vcache : c->rows[(size_t)row].kcache) + (size_t)layer * d.kv_heads * per_head * c->esz;
    for (int h = 0; h < d.kv_heads; h++) {
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
  if (!c || !c->finalized || row < 0 || row >= c->d.max_batch || layer < 0 || layer >= c->d.layers || n_rows < 0 || n_rows > c->past) return c ? set_err(c, TGX_ERR_INVALID, "write_kv: row / layer / n_rows out of range") : TGX_ERR_INVALID;
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  const tgx_model_desc& d = c->d;
  const size_t hd = (size_t)d.head_dim, per_head = (size_t)d.max_ctx * hd, T = (size_t)n_rows;
  std::vector<unsigned char> tmp(T * hd * c->esz);
  for (int which = 0; which < 2; which++) {
    const float* in = which ? v_in : k_in;
    if (!in || !T) continue;
    ebyte* base = (which ? c->rows[(size_t)row].vcache : c->rows[(size_t)row].kcache) + (size_t)layer * d.kv_heads * per_head * c->esz;
    for (int h = 0; h < d.kv_heads; h++) {
      for (size_t t = 0; t < T; t++)
        for (size_t k = 0; k < hd; k++) {   // BSHD view in, head-major cache out; one round-to-nearest-even into the storage dtype
          const float v = in[(t * d.kv_heads + h) * hd + k];
          const size_t i = t * hd + k;
          if (c->dt == tgx::DT_F32) memcpy(tmp.data() + 4 * i, &v, 4);
          else { const uint16_t u = c->dt == tgx::DT_BF16 ? host_f32_to_bf16(v) : host_f32_to_half(v); memcpy(tmp.data() + 2 * i, &u, 2); }
        }
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
      else for (int l = 0; l < c->d.layers; l++, n++) launch_layer_kernel(c, &c->rows[0], 1, c->prof_same_layer ? 0 : l, cls, c->scratch_x, (long long)c->kv_row_elems);
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
  if (!c->have_probs) return set_err(c, TGX_ERR_STATE, "no probabilities: the last sample was greedy or none was taken");
  HIP_OK(c, hipSetDevice(c->device));
  HIP_OK(c, hipStreamSynchronize(c->stream));
  const size_t V = (size_t)c->d.vocab;
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
    hipLaunchKernelGGL(tgx::argmax_partials_kernel, dim3(c->lm_grid), dim3(256), 0, c->stream, r.logits, (int)V, r.part_val, r.part_idx);
  }
  HIP_OK(c, hipGetLastError());
  HIP_OK(c, hipStreamSynchronize(c->stream));
  c->batch = batch;
  c->have_logits = true;
  return TGX_OK;
}

int tgx_set_option(tgx_ctx* c, const char* key, int value) {
  if (!c || !key) return TGX_ERR_INVALID;
  static const char* cls_names[TGX_KERNEL_COUNT] = {"qkv", "attn", "oproj", "gateup", "down", "lmhead"};
  drop_step_graphs(c);
  if (!strcmp(key, "graph")) { c->use_graph = value != 0; return TGX_OK; }
  if (!strcmp(key, "graph.steps")) { if (value < 1 || value > 64) return set_err(c, TGX_ERR_INVALID, "graph.steps out of range"); c->graph_steps = value; return TGX_OK; }
  if (!strcmp(key, "debug.nops")) { c->debug_nops = value; return TGX_OK; }
  if (!strcmp(key, "debug.skip")) { c->debug_skip = value; return TGX_OK; }
  if (!strcmp(key, "debug.attn")) { c->debug_attn = value; return TGX_OK; }
  if (!strcmp(key, "attn.gmax")) {   // query heads per attention workgroup: the kernel is instantiated for 1..4 (0 = default)
    if (value < 0 || value > 4) return set_err(c, TGX_ERR_INVALID, "attn.gmax must be 0 (default) or 1..4");
    c->attn_gmax = value; return TGX_OK;
  }
  if (!strcmp(key, "attn.direct_max")) { c->attn_direct_max = value; return TGX_OK; }
  if (!strcmp(key, "attn.direct_nw4")) { drop_step_graphs(c); c->attn_direct_nw4 = value; return TGX_OK; }
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
  if (!strcmp(key, "prefill.f32_flash")) { c->f32_flash = value; return TGX_OK; }
  if (!strcmp(key, "prefill.f32_min_rows")) { c->prefill_f32_min_rows = value; return TGX_OK; }
  if (!strcmp(key, "prefill.splitk")) { c->gemm_splitk = value; return TGX_OK; }
  if (!strcmp(key, "prefill.qkv_balanced")) { c->qkv_balanced = value != 0; return TGX_OK; }
  if (!strcmp(key, "decode.step_rows")) { if (value != 32 && value != 64 && value != 128) return set_err(c, TGX_ERR_INVALID, "decode.step_rows is 32, 64 or 128"); drop_step_graphs(c); c->decode_step_rows = value; return TGX_OK; }
  if (!strcmp(key, "prefill.skinny_hidden_max_wide")) { c->prefill_skinny_hidden_max_wide = value; return TGX_OK; }
  if (!strcmp(key, "prefill.skinny_hidden_max")) { c->prefill_skinny_hidden_max = value; return TGX_OK; }
  if (!strcmp(key, "prefill.skinny_rows")) { if (value < 0 || value > 128) return set_err(c, TGX_ERR_INVALID, "prefill.skinny_rows is 0..128"); c->prefill_skinny_rows = value; return TGX_OK; }
  if (!strcmp(key, "prefill.wide_8k_max")) { c->wide_8k_max = value; return TGX_OK; }
  if (!strcmp(key, "prefill.wide_8k")) { c->wide_8k = value != 0; return TGX_OK; }
  if (!strcmp(key, "prefill.hidden_256")) { c->hidden_256 = value != 0; return TGX_OK; }
  if (!strcmp(key, "prefill.splitk_dma")) { if (value < 0 || value > 2) return set_err(c, TGX_ERR_INVALID, "prefill.splitk_dma is 0, 1 (<= 64 rows) or 2 (always)"); c->splitk_dma = value; return TGX_OK; }
  if (!strcmp(key, "prefill.attn_mirror")) { c->attn_mirror = value; return TGX_OK; }
  if (!strcmp(key, "attn.qk_fuse")) { c->qk_fuse = value; return TGX_OK; }
  if (!strcmp(key, "attn.fold_combine")) { c->attn_fold = value != 0; return TGX_OK; }
  if (!strcmp(key, "attn.fold_ticket")) { drop_step_graphs(c); c->attn_ticket = value != 0; return TGX_OK; }
  if (!strcmp(key, "lmhead.fuse_finalize")) { c->lm_fuse = value != 0; return TGX_OK; }
  if (!strcmp(key, "skinny.wgs")) { if (value < 1) return set_err(c, TGX_ERR_INVALID, "skinny.wgs must be >= 1"); c->skinny_wgs = value; return TGX_OK; }
  if (!strcmp(key, "prefill.skinny")) { c->prefill_skinny = value; return TGX_OK; }
  if (!strcmp(key, "skinny.gu_split")) { c->skinny_gu_split = value; return TGX_OK; }
  if (!strcmp(key, "skinny.cfg_mid")) { if (value < 0 || value > 2) return set_err(c, TGX_ERR_INVALID, "skinny.cfg_mid is 0..2"); c->skinny_cfg_mid = value; return TGX_OK; }
  if (!strcmp(key, "skinny.cfg")) { if (value < -1 || value > 2) return set_err(c, TGX_ERR_INVALID, "skinny.cfg is -1..2"); c->skinny_cfg_force = value; return TGX_OK; }
  if (!strcmp(key, "decode.mfma_min_batch")) { if (value < 1) return set_err(c, TGX_ERR_INVALID, "decode.mfma_min_batch must be >= 1"); c->decode_mfma_min = value; return TGX_OK; }
  if (!strcmp(key, "debug.profile_same_layer")) { c->prof_same_layer = value; return TGX_OK; }
  if (!strncmp(key, "pf.", 3)) {   // L2 prefetch chaining (kernels/l2_prefetch.h); the captured step graphs hold the old launch geometry
    int* dst = !strcmp(key, "pf.mode") ? &c->pf_mode : !strcmp(key, "pf.wgs") ? &c->pf_wgs : !strcmp(key, "pf.stride") ? &c->pf_stride : !strcmp(key, "pf.oproj_kb") ? &c->pf_oproj_kb
             : !strcmp(key, "pf.gu_kb") ? &c->pf_gu_kb : !strcmp(key, "pf.dn_kb") ? &c->pf_dn_kb : !strcmp(key, "pf.qkv_kb") ? &c->pf_qkv_kb : !strcmp(key, "pf.comb_kb") ? &c->pf_comb_kb : !strcmp(key, "pf.oproj_gu_kb") ? &c->pf_oproj_gu_kb : nullptr;
    if (!dst) return set_err(c, TGX_ERR_INVALID, "unknown option %s", key);
    if (value < 0 || (dst == &c->pf_wgs && (value < 8 || value > 1024)) || (dst == &c->pf_stride && value != 64 && value != 128 && value != 256)) return set_err(c, TGX_ERR_INVALID, "%s out of range", key);
    drop_step_graphs(c);
    *dst = value;
    return TGX_OK;
  }
  if (!strncmp(key, "engine.", 7)) {   // the persistent decode engine (kernels/engine.h); captured step graphs hold the old launch sequence
    if (!strcmp(key, "engine.mode")) { if (value < 0 || value > 2) return set_err(c, TGX_ERR_INVALID, "engine.mode is 0 (off), 1 (gate_up + down) or 2 (o_proj .. next qkv)"); drop_step_graphs(c); c->engine_mode = value; return TGX_OK; }
    if (!strcmp(key, "engine.ns")) { if (value < 0 || value > 15) return set_err(c, TGX_ERR_INVALID, "engine.ns is 0 (auto) .. 15"); drop_step_graphs(c); c->engine_ns = value; return TGX_OK; }
    if (!strcmp(key, "engine.depth")) { if (value < 2 || value > 4) return set_err(c, TGX_ERR_INVALID, "engine.depth is 2 .. 4"); drop_step_graphs(c); c->engine_depth = value; return TGX_OK; }
    if (!strcmp(key, "engine.thin")) { if (value < 0 || value > 2) return set_err(c, TGX_ERR_INVALID, "engine.thin is 0 .. 2"); drop_step_graphs(c); c->engine_thin = value; return TGX_OK; }
    if (!strcmp(key, "engine.stats")) {
      drop_step_graphs(c);
      if (value && !c->eng_stats) {
        if (!c->finalized) return set_err(c, TGX_ERR_STATE, "engine.stats needs a finalized context");
        int rc;
        if ((rc = dev_alloc(c, &c->eng_stats, (size_t)c->d.layers * c->num_cus * tgx::ENG_NSTAT))) return rc;
        HIP_OK(c, hipMemset(c->eng_stats, 0, (size_t)c->d.layers * c->num_cus * tgx::ENG_NSTAT * 8));
      }
      c->engine_stats = value != 0;
      return TGX_OK;
    }
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
