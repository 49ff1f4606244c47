/*
 * tgx.h — the drop-in boundary of the MI355X decode path (C ABI, no C++/torch types).
 *
 * TinyGPT has no FFI of its own: the device is a tinytorch::Device value threaded through
 * GPTConfig -> ModelLoader::load -> every nn::Module (reference: src/engine/GPTEngine.h:25-32,
 * src/huggingface/ModelLoader.cpp:25-89) and CPU-vs-CUDA dispatch happens inside TinyTorch.
 * This header defines the boundary at the seams the reference does expose — the virtual
 * GPTModel interface (src/model/GPTModel.h:80-106), KVCacheManager (src/engine/CacheManager.h:18-55),
 * Sampler::sample (src/engine/Sampler.h:30, Sampler.cpp:23-79) and the AsyncTokenPipeline hook
 * (src/engine/GPTEngine.cpp:17-35).  Each entry point names the reference interface it replaces.
 *
 * Conventions (mirroring the reference's: bool returns + LOGE, exceptions disabled,
 * src/CMakeLists.txt:63-67; single engine thread, server/HttpServer.cpp:118-163):
 *   - every function returns a tgx_status (0 = ok) and never throws across the ABI;
 *   - one opaque context per GPU; all calls on a context come from one host thread;
 *   - the caller owns host buffers; the library owns device memory and one HIP stream;
 *   - host buffers may be pageable; ids are int64 like the reference's token tensors.
 */
#ifndef TGX_H
#define TGX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 5): per-row sequence lifecycle (tgx_reset_row / tgx_forward_row / tgx_sample_row / tgx_past_length_row); tgx_get_option added and
 * tgx_engine_read_stats + the options engine.*, pf.*, attn.fold_*, lmhead.fuse_finalize removed since 2. */
#define TGX_ABI_VERSION 3

#if defined(__GNUC__)
#define TGX_API __attribute__((visibility("default")))
#else
#define TGX_API
#endif

typedef struct tgx_ctx tgx_ctx;

typedef enum tgx_status {
  TGX_OK = 0,
  TGX_ERR_INVALID = 1,     /* bad argument (null pointer, batch/seq out of range, ...)        */
  TGX_ERR_UNSUPPORTED = 2, /* family / dtype / head_dim this backend does not implement       */
  TGX_ERR_DEVICE = 3,      /* a HIP call failed; tgx_last_error() has the HIP error string     */
  TGX_ERR_STATE = 4,       /* call order violated (forward before finalize, missing tensors)  */
  TGX_ERR_NOMEM = 5,       /* host or device allocation failed                                */
  TGX_ERR_NAME = 6,        /* tensor name not part of this model ("Unexpected key")           */
  TGX_ERR_SHAPE = 7,       /* "shape not equal for tensor" (SafeTensors.cpp:187-193)          */
  TGX_ERR_CONTEXT = 8      /* pastLength + seq would exceed contextSize()                     */
} tgx_status;

/* GPTModelType (src/model/GPTModel.h:71-78) */
typedef enum tgx_family {
  TGX_FAMILY_GPT2 = 1,
  TGX_FAMILY_LLAMA = 2,
  TGX_FAMILY_QWEN2 = 3,
  TGX_FAMILY_QWEN3 = 4,
  TGX_FAMILY_MISTRAL = 5
} tgx_family;

/* tinytorch::DType values the CLI accepts (examples/inference/main.cpp:82-88) */
typedef enum tgx_dtype { TGX_F32 = 0, TGX_BF16 = 1, TGX_F16 = 2 } tgx_dtype;

/*
 * What the family factories derive from config.json (src/huggingface/ModelConfig.cpp:63-122,
 * src/model/ModelLlama.h:21-53, ModelQwen2.h:23-45, ModelMistral.h:23-40, ModelGPT2.h:226-230).
 */
typedef struct tgx_model_desc {
  int32_t family;         /* tgx_family                                                          */
  int32_t hidden;         /* hidden_size / n_embd                                                */
  int32_t layers;         /* num_hidden_layers / n_layer                                         */
  int32_t heads;          /* num_attention_heads / n_head                                        */
  int32_t kv_heads;       /* num_key_value_heads (must divide heads, Attention.h:38)             */
  int32_t head_dim;       /* hidden/heads for llama/qwen2/mistral (ModelLlama.h:37)              */
  int32_t inter;          /* intermediate_size (4*n_embd for GPT-2, ModelGPT2.h:96)              */
  int32_t vocab;          /* vocab_size                                                          */
  int32_t max_ctx;        /* GPTModel::contextSize(): KV capacity and RoPE table length          */
  int32_t qkv_bias;       /* 1 for Qwen2 (ModelQwen2.h:26-31) and GPT-2                          */
  int32_t tied;           /* tie_word_embeddings: lm_head aliases embed_tokens (GPTModel.h:39-41)*/
  int32_t compute_dtype;  /* tgx_dtype the model is cast to after load (ModelLoader.cpp:84)      */
  float norm_eps;         /* rms_norm_eps / layer_norm_epsilon                                   */
  float rope_theta;       /* rope_theta                                                          */
  float rope_factor;      /* llama3 RopeScalingConfig.factor; 0 = std::nullopt (no scaling)      */
  float rope_low_freq;    /* .lowFreqFactor                                                      */
  float rope_high_freq;   /* .highFreqFactor                                                     */
  int32_t rope_orig_ctx;  /* .originalMaxPositionEmbeddings                                      */
  int32_t n_positions;    /* GPT-2 wpe rows; 0 otherwise                                         */
  int32_t max_batch;      /* rows of independent KV state to allocate (>= 1)                     */
  int32_t qk_norm;        /* 1: per-head RMSNorm on q and k before RoPE (AttentionWithQKNorm,    */
                          /*    src/layer/Attention.h:128-167; Qwen3, ModelQwen3.h:23-40)        */
} tgx_model_desc;

/* SamplerConfig (src/engine/Sampler.h:13-22).  Greedy iff temperature<=0 && top_k<=0 &&
 * top_p>=1 && min_p<=0 (Sampler.cpp:15-21). */
typedef struct tgx_sampler_cfg {
  float temperature;
  int64_t top_k;
  float top_p;
  float min_p;
} tgx_sampler_cfg;

/* ---- lifetime ---------------------------------------------------------------------------- */

/* Number of visible MI355X devices (the `--device mi355x` probe; reference branch:
 * examples/inference/main.cpp:76-80). */
TGX_API int tgx_device_count(int* out_count);

/* == Model{Llama,Qwen2,Mistral}::Model*(config, device) (e.g. src/model/ModelLlama.h:57-65):
 * validates the description, binds the GPU, allocates parameter storage in compute_dtype. */
TGX_API int tgx_create(const tgx_model_desc* desc, int device_ordinal, tgx_ctx** out_ctx);

/* == SafeTensors::loadInternal -> Storage::copyOnDevice for ONE named tensor
 * (src/util/SafeTensors.cpp:157-215).  `hf_name` is the checkpoint key; q/k/v and gate/up land in
 * the row slices of the merged weights exactly like MergedLinear's LinearRef views
 * (src/layer/Linear.h:64-79).  Shape must match (TGX_ERR_SHAPE); an unknown key is TGX_ERR_NAME
 * (the reference warns "Unexpected key" and continues — callers may ignore that status).
 * ndim < 0 is a name probe: nothing is copied; TGX_ERR_NAME = unknown key, TGX_OK = a key this path knows and
 * ignores, TGX_ERR_SHAPE = a parameter the model needs (the loader uses it for tensors whose file dtype it cannot convert).
 * `src_dtype` is the dtype of `host`; conversion to compute_dtype happens on upload
 * (bf16->fp32 exact, fp32->bf16 round-to-nearest-even), == model().to(dtype), ModelLoader.cpp:84. */
TGX_API int tgx_upload(tgx_ctx* ctx, const char* hf_name, const void* host, const int64_t* shape, int ndim,
               int src_dtype);

/* == model().eval() + GPTModel::init(): checks every tensor arrived (TGX_ERR_STATE names the
 * first missing key in tgx_last_error), builds the RoPE tables (nn::RoPE ctor, ModelLlama.h:41-42),
 * allocates the KV cache for max_batch x max_ctx tokens and instantiates the decode graph. */
TGX_API int tgx_finalize(tgx_ctx* ctx);

TGX_API void tgx_destroy(tgx_ctx* ctx);

/* ---- the hot path ------------------------------------------------------------------------ */

/* == GPTModel::forward(inputIds[B,S]) + narrow-to-last-position (src/model/GPTModel.h:86,
 * src/engine/GPTEngine.cpp:96-97).  `ids` is a row-major host array [batch][seq] (left-padded by
 * the caller, no mask — GPTEngine.cpp:95).  Appends seq positions to every row's KV cache
 * (KVCacheManager::append, CacheManager.h:24-42) and leaves the last-position logits [batch][vocab]
 * on the device for tgx_sample / tgx_read_logits.  seq>1 with pastLength>0 is rejected with
 * TGX_ERR_INVALID (the reference would run it non-causally, Attention.h:108; its engine never does). */
TGX_API int tgx_forward(tgx_ctx* ctx, const int64_t* ids, int batch, int seq);

/* Copies the logits of the last tgx_forward / decode step to `out` [batch*vocab] as fp32.
 * rounded=1: values as the reference's logits tensor holds them (rounded to compute_dtype);
 * rounded=0: the fp32 accumulators before that rounding (for tolerance checks). */
TGX_API int tgx_read_logits(tgx_ctx* ctx, float* out, int rounded);

/* == Sampler::sample(logits[B,V]) (src/engine/Sampler.cpp:23-79) on the current logits.
 * The sampled ids become the device-resident "next token" of every row; if out_ids != NULL they
 * are also copied to the host ([batch], int64).  `seed` drives the multinomial draw (the
 * reference's RNG stream is not reproducible; greedy ignores it). */
TGX_API int tgx_sample(tgx_ctx* ctx, const tgx_sampler_cfg* cfg, uint64_t seed, int64_t* out_ids);

/* == the decode loop body of GPTEngine::generateSync (src/engine/GPTEngine.cpp:165-168), n_steps
 * times: nextToken = sample(forward(nextToken)).  Token ids and positions stay on the GPU between
 * steps; out_ids (may be NULL) receives [n_steps][batch] int64. */
TGX_API int tgx_decode(tgx_ctx* ctx, const tgx_sampler_cfg* cfg, uint64_t seed, int n_steps, int64_t* out_ids);

/* == AsyncTokenPipeline (src/engine/GPTEngine.cpp:17-35) for generateAsync's one-step lookahead
 * (GPTEngine.cpp:196-217), batch row 0.  tgx_step_async enqueues one decode step and returns
 * immediately with a ticket; tgx_fetch_token blocks until the step with that ticket has sampled
 * and returns its id.  Ticket 0 refers to the token produced by the last tgx_sample. */
TGX_API int tgx_step_async(tgx_ctx* ctx, const tgx_sampler_cfg* cfg, uint64_t seed, int64_t* out_ticket);
TGX_API int tgx_fetch_token(tgx_ctx* ctx, int64_t ticket, int32_t* out_id);

/* == GPTModel::resetCache() (src/model/GPTModel.h:91-94): pastLength back to 0 for all rows. */
TGX_API int tgx_reset_cache(tgx_ctx* ctx);

/* == KVCacheManager::pastLength (src/engine/CacheManager.h:44-51).  With rows of different lengths (see the per-row calls below): the LONGEST
 * row of the batch — the length every capacity check uses. */
TGX_API int64_t tgx_past_length(const tgx_ctx* ctx);

/* ---- per-row sequence lifecycle (ABI 3) ---------------------------------------------------
 * The kernel-contract half of the reference's continuous-batching TODO (README.md:33-34): the reference's KVCacheManager holds ONE pastLength for
 * the whole batch (src/engine/CacheManager.h:44-51) and its engine rebuilds the batch per request (src/engine/GPTEngine.cpp:67-84,180-232), so a
 * finished sequence blocks its slot until the longest one ends.  Here every row owns its cache slab and its device-resident position, and the
 * step kernels read the position per row: a row can be retired and another prompt prefilled into it while the other rows keep their state.
 * What a row keeps is the semantics of a solo sequence (Attention.h:71-112 over its own keys [0, pastLength_row]): its logits equal those of the
 * same prompt run alone up to the summation-order differences between kernel paths (tests/test_hip_rows.py).
 *
 * tgx_reset_row      == resetCache() for ONE row: its pastLength back to 0; the other rows and the batch size are untouched.  The row is now
 *                       RETIRED: tgx_decode / tgx_step_async keep stepping the live rows without waiting for it (a finished sequence with no queued
 *                       prompt stalls nobody).  It still rides in the steps (it decodes from position 0 on its stale token; its ids and logits in
 *                       tgx_decode's output / tgx_read_logits are meaningless and harmless) and counts for neither tgx_past_length nor the context
 *                       check; tgx_past_length_row reports 0 for it.  With every row retired there is nothing to step: tgx_decode refuses.
 * tgx_forward_row    == GPTModel::forward(inputIds[1,S]) for ONE row of the live batch: a retired `row` < batch (refill) or `row` == batch (the
 *                       batch grows by one row, up to max_batch); a live row must be retired first (tgx_reset_row).  Leaves the row's last-position
 *                       logits in its slot of tgx_read_logits; the row is live again but has no current token until tgx_sample_row (or tgx_sample)
 *                       ran — tgx_decode refuses until then.
 * tgx_sample_row     == Sampler::sample on ONE row's logits; the id becomes that row's device-resident next token.
 * tgx_past_length_row   the row's own pastLength. */
TGX_API int tgx_reset_row(tgx_ctx* ctx, int row);
TGX_API int tgx_forward_row(tgx_ctx* ctx, int row, const int64_t* ids, int seq);
TGX_API int tgx_sample_row(tgx_ctx* ctx, int row, const tgx_sampler_cfg* cfg, uint64_t seed, int64_t* out_id);
TGX_API int64_t tgx_past_length_row(const tgx_ctx* ctx, int row);

/* == GPTModel::contextSize() / numLayers() (src/model/GPTModel.h:97-98). */
TGX_API int64_t tgx_context_size(const tgx_ctx* ctx);
TGX_API int32_t tgx_num_layers(const tgx_ctx* ctx);

/* ---- diagnostics ------------------------------------------------------------------------- */

/* Last error text for this context (or for tgx_create when ctx == NULL). */
TGX_API const char* tgx_last_error(const tgx_ctx* ctx);

/* Blocks until all work enqueued on the context's stream has finished. */
TGX_API int tgx_synchronize(tgx_ctx* ctx);

/* Reads back the KV cache of (row, layer) as fp32 into k_out/v_out, each [pastLength of that row][kv_heads][head_dim]
 * (the BSHD view KVCacheManager::append returns, Attention.h:106).  Test/diagnostic use. */
TGX_API int tgx_read_kv(tgx_ctx* ctx, int row, int layer, float* k_out, float* v_out);

/* The inverse of tgx_read_kv: overwrites cache rows [0, n_rows) of (row, layer), n_rows <= pastLength, from fp32 k_in / v_in
 * [n_rows][kv_heads][head_dim] (either may be NULL), rounded once to the cache's storage dtype.  Test/diagnostic use: with the CPU path's
 * cache rows injected, a decode step is compared free of the bf16 KV-rounding floor (the reference's KVCacheManager holds the tensors
 * it was given, CacheManager.h:24-51 — there is nothing to overwrite there). */
TGX_API int tgx_write_kv(tgx_ctx* ctx, int row, int layer, const float* k_in, const float* v_in, int64_t n_rows);

/* Per-kernel-class timing with HIP events on the context's stream (bench.py's roofline leg).  For each
 * class (TGX_KERNEL_*) the kernels of ALL layers are launched back-to-back between two events, n_reps times,
 * at the current context length; returns launch counts and summed milliseconds.  Every launch streams a
 * different layer's weights (nothing repeats out of the Infinity Cache); residual outputs go to a scratch
 * vector so pastLength, the KV cache and the current token are unchanged (the logits buffer is consumed). */
#define TGX_KERNEL_QKV 0
#define TGX_KERNEL_ATTN 1
#define TGX_KERNEL_OPROJ 2
#define TGX_KERNEL_GATEUP 3
#define TGX_KERNEL_DOWN 4
#define TGX_KERNEL_LMHEAD 5
#define TGX_KERNEL_COUNT 6
TGX_API int tgx_profile_decode(tgx_ctx* ctx, int n_reps, int64_t* launches /*[TGX_KERNEL_COUNT]*/,
                       double* total_ms /*[TGX_KERNEL_COUNT]*/);

/* Final probability vector(s) [batch*vocab] of the last non-greedy tgx_sample / decode step: what the
 * reference passes to multinomial (Sampler.cpp:77) — zero where top-k/top-p/min-p removed a token.  A step
 * does not materialise the vector (the draw needs per-tile sums only): this call evaluates it from what the
 * step left on the device — valid until the next tgx_forward / tgx_forward_row / tgx_set_logits replaces the
 * logits (TGX_ERR_STATE then); rows whose last step was greedy read as zeros. */
TGX_API int tgx_read_probs(tgx_ctx* ctx, float* out);

/* Injects logits [batch][vocab] as if a forward had produced them, so that Sampler::sample can be
 * exercised on its own (the reference's Sampler takes any [B,V] tensor, Sampler.h:30). */
TGX_API int tgx_set_logits(tgx_ctx* ctx, const float* logits, int batch);

/* Launch-geometry knobs for tuning sweeps (results stay within the parity tolerances; summation order may change):
 *   "<class>.ks"   waves sharing a row pair's K range (1/2/4), class in {qkv,oproj,gateup,down,lmhead}
 *   "<class>.bpc"  grid cap in workgroups per CU ("lmhead.bpc" only before tgx_finalize)
 *   "attn.nsplit"  KV splits per kv head (<= 32, before tgx_finalize); "attn.gmax" query heads per attention workgroup
 *   "attn.direct_max"  contexts up to this many keys run attention as one workgroup per head group, without the combine launch (default 768 at head_dim 64, 384 at 128; 0 = never)
 *   "graph" 0/1    hipGraph replay vs eager launches; "graph.steps" decode steps per graph for long tgx_decode calls
 *   "prefill.min_rows"  prompts shorter than this take passes through the decode kernels instead of the batched prefill (default 7)
 *   "prefill.splitk" 0/1  split K over workgroups when a prompt gives the GEMMs few row tiles (default 1)
 *   "prefill.mfma" 0/1  batched matrix-core prefill vs passes through the decode kernels; "prefill.gemm_tm" 64/128 row tile
 *   "act.round16" 0/1  (default 0) numerics contract: 1 = the input of every Linear is rounded to the storage dtype (round-to-nearest-even) before the
 *                  product — what a module constructed in config.torch_dtype sees (ModelLlama.h:62) — instead of entering in fp32.  The matrix-core
 *                  products of the prefill and of batched steps then take one 16-bit term per activation instead of two or three;
 *                  fp32 storage: no effect.  Set it before the first tgx_forward of a sequence: the cache of a sequence must be filled under one contract
 *   "kv.budget_tokens"  (default 0; set BEFORE tgx_finalize, which sizes the caches; 16-bit storage dtypes) PAGED KV — the kernel contract of the reference's
 *                  "Paged Attention" TODO (README.md:32-34).  0: every row owns a max_ctx slab ([layer][kv_head][max_ctx][hd], max_batch x max_ctx tokens of
 *                  cache whatever the rows hold).  N > 0: the caches are pools of 128-token blocks worth N tokens in all, shared by the rows; a row's blocks are
 *                  assigned as its sequence grows (a device-resident block table per row, read by the attention and cache-append kernels) and returned by
 *                  tgx_reset_row / tgx_reset_cache; max_ctx stays the per-row limit.  A call that would need a block when none is free returns TGX_ERR_CONTEXT and
 *                  changes nothing (retire a row, retry).  Results are those of the unpaged cache, bit for bit on the same kernels (tests/test_hip_paged.py);
 *                  tgx_get_option "kv.free_tokens" = the unassigned blocks' worth of tokens (-1 when unpaged)
 *   "debug.*"      experiment switches (tools/gemv_dissect.py, tools/attn_dissect.py; live only in a -DTGX_DISSECT=1 build) */
TGX_API int tgx_set_option(tgx_ctx* ctx, const char* key, int value);

/* Reads back what the library would do for the CURRENT batch size (no reference counterpart; callers that time a decode region use it to keep the
 * region on one attention form instead of mirroring the defaults):
 *   "attn.direct_limit"  decode steps at contexts up to this many keys run the direct attention form (batch-1 steps at head_dim 64: with the
 *                        o_proj product in the same launch), beyond it the split form
 *   "attn.nw4_limit"     ... and up to this many keys its four-wave variant (0 = never)
 *   "graph.steps"        decode steps per captured multi-step graph */
TGX_API int tgx_get_option(const tgx_ctx* ctx, const char* key, int* out_value);

/* Algorithmic HBM bytes one decoded token streams at context length T (SURVEY.md §8d formula). */
TGX_API int64_t tgx_bytes_per_token(const tgx_ctx* ctx, int64_t T);

TGX_API int tgx_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TGX_H */
