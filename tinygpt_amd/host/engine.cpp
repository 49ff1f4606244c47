#include "engine.h"

#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

namespace tgxh {

namespace {

std::string self_dir() {
  Dl_info info;
  if (dladdr(reinterpret_cast<void*>(&self_dir), &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    size_t k = p.find_last_of('/');
    if (k != std::string::npos) return p.substr(0, k);
  }
  return ".";
}

tgx_sampler_cfg to_c(const SamplerConfig& s) { tgx_sampler_cfg c; c.temperature = s.temperature; c.top_k = s.topK; c.top_p = s.topP; c.min_p = s.minP; return c; }

}  // namespace

GPTEngine::GPTEngine(GPTConfig config) : config_(std::move(config)) {}

GPTEngine::~GPTEngine() {
  if (model_.ctx && be_.destroy) be_.destroy(model_.ctx);
  be_.close();
}

bool GPTEngine::fail(const std::string& what) {
  err_ = what;
  fprintf(stderr, "[tgx] %s\n", what.c_str());
  return false;
}

bool GPTEngine::prepare() {
  // device -> shim.  The reference maps everything that is not "cpu" to CUDA (examples/inference/main.cpp:76-80); here
  // "mi355x" binds the HIP library and any other value is refused — this engine has no CPU execution path of its own.
  std::string lib = self_dir() + "/libtgx_mi355x.so", prefix = "tgx_";
  bool hooked = false;
#ifdef TGXH_TEST_HOOKS
  if (!config_.backendLib.empty()) { hooked = true; lib = config_.backendLib; prefix = config_.backendPrefix; }
#endif
  if (config_.device != "mi355x" && !hooked)
    return fail("device '" + config_.device + "' is not provided by this engine (only --device mi355x); the reference's cpu path lives in TinyTorch");
  if (!be_.open(lib, prefix)) return fail("cannot bind device shim " + lib + ": " + be_.error);

  if (!config_.synthetic.empty()) {
    if (!known_config(config_.synthetic, config_.dtype, config_.maxBatch, model_.config)) return fail("unknown synthetic config: " + config_.synthetic);
    if (!load_synthetic(be_, model_.config, config_.deviceOrdinal, 1234, 0.02, &model_.ctx, err_)) return fail("Prepare failed: " + err_);
  } else {
    if (!load_model_dir(be_, config_.modelDir, config_.deviceOrdinal, config_.dtype, config_.maxBatch, model_, err_)) return fail("Prepare failed: " + err_);
  }
  // tokenizer: optional for the id entry points, required for the text ones (ModelLoader.cpp:60-69 always loads it)
  const std::string tdir = !config_.tokenizerDir.empty() ? config_.tokenizerDir : config_.modelDir;
  if (!tdir.empty()) {
    const std::string tj = tdir + "/tokenizer.json", tc = tdir + "/tokenizer_config.json";
    if (FILE* f = fopen(tj.c_str(), "rb")) {
      fclose(f);
      if (!tokenizer_.initWithConfig(tj, tc)) return fail("load tokenizer failed: " + tokenizer_.lastError());
      tokenizerOk_ = true;
    } else if (!config_.tokenizerDir.empty()) return fail("Cannot open file: " + tj);
  }
  // EOS ids: generation_config first, else the tokenizer's eos, else the model config's (:50-61)
  for (int64_t id : model_.generation.eos_token_ids) baseEosTokenIds_.push_back((int32_t)id);
  if (baseEosTokenIds_.empty() && tokenizerOk_ && tokenizer_.eosTokenId() >= 0) baseEosTokenIds_.push_back(tokenizer_.eosTokenId());
  if (baseEosTokenIds_.empty() && model_.config.eos_token_id >= 0) baseEosTokenIds_.push_back((int32_t)model_.config.eos_token_id);
  eosTokenIds_ = baseEosTokenIds_;
  prepared_ = true;
  return true;
}

void GPTEngine::reconfigure(const SamplerConfig& samplerConfig, int64_t maxNewTokens, const std::vector<int32_t>& extraStopTokenIds) {
  config_.samplerConfig = samplerConfig;
  config_.maxNewTokens = maxNewTokens;
  eosTokenIds_ = baseEosTokenIds_;
  for (int32_t id : extraStopTokenIds) if (!isEosToken(id)) eosTokenIds_.push_back(id);
  if (model_.ctx) be_.reset_cache(model_.ctx);           // context_.model->resetCache()  (:83)
}

bool GPTEngine::isEosToken(int32_t id) const { return std::find(eosTokenIds_.begin(), eosTokenIds_.end(), id) != eosTokenIds_.end(); }

int64_t GPTEngine::contextSize() const { return model_.ctx ? be_.context_size(model_.ctx) : 0; }

std::vector<int64_t> GPTEngine::alignPrompts(const std::vector<std::vector<int32_t>>& prompts, int32_t padToken, int64_t& maxLen) const {
  maxLen = 0;
  for (const auto& p : prompts) maxLen = std::max<int64_t>(maxLen, (int64_t)p.size());
  maxLen = std::min<int64_t>(maxLen, contextSize());
  std::vector<int64_t> ids((size_t)(prompts.size() * maxLen));
  for (size_t b = 0; b < prompts.size(); b++) {
    const auto& t = prompts[b];
    int64_t* row = ids.data() + b * maxLen;
    if ((int64_t)t.size() > maxLen) {
      for (int64_t i = 0; i < maxLen; i++) row[i] = t[t.size() - (size_t)maxLen + (size_t)i];     // keep the tail (:127-129)
    } else {
      const int64_t pad = maxLen - (int64_t)t.size();
      for (int64_t i = 0; i < pad; i++) row[i] = padToken;                                         // left pad (:130-138)
      for (size_t i = 0; i < t.size(); i++) row[pad + (int64_t)i] = t[i];
    }
  }
  return ids;
}

GPTOutput GPTEngine::generateSync(const std::vector<std::vector<int32_t>>& prompts, int32_t padToken) {
  GPTOutput out;
  if (!prepared_ || prompts.empty()) { fail("generateSync: engine not prepared or empty batch"); return out; }
  const int B = (int)prompts.size();
  int64_t S = 0;
  std::vector<int64_t> ids = alignPrompts(prompts, padToken, S);
  if (S == 0) { fail("generateSync: every prompt is empty (nothing to prefill)"); return out; }
  const tgx_sampler_cfg sc = to_c(config_.samplerConfig);
  const int64_t n_new = std::max<int64_t>(1, config_.maxNewTokens);

  // prefill (mask ignored, like the reference: "TODO padding mask", GPTEngine.cpp:95)
  const auto t0 = std::chrono::steady_clock::now();
  if (be_.forward(model_.ctx, ids.data(), B, (int)S) != TGX_OK) { fail(std::string("forward: ") + be_.last_error(model_.ctx)); return out; }
  std::vector<int64_t> first((size_t)B), rest((size_t)(B * (n_new - 1)));
  if (be_.sample(model_.ctx, &sc, config_.seed, first.data()) != TGX_OK) { fail(std::string("sample: ") + be_.last_error(model_.ctx)); return out; }
  const auto t1 = std::chrono::steady_clock::now();
  // decode: maxNewTokens-1 iterations, no EOS check (:165-172)
  if (n_new > 1 && be_.decode(model_.ctx, &sc, config_.seed, (int)(n_new - 1), rest.data()) != TGX_OK) {
    fail(std::string("decode: ") + be_.last_error(model_.ctx));
    return out;
  }
  out.firstTokenMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
  out.decodeMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
  out.batch = B;
  out.newTokens = n_new;
  out.tokenIds.resize((size_t)(B * (S + n_new)));
  for (int b = 0; b < B; b++) {
    int32_t* row = out.tokenIds.data() + (size_t)b * (size_t)(S + n_new);
    for (int64_t i = 0; i < S; i++) row[i] = (int32_t)ids[(size_t)(b * S + i)];
    row[S] = (int32_t)first[(size_t)b];
    for (int64_t i = 0; i + 1 < n_new; i++) row[S + 1 + i] = (int32_t)rest[(size_t)(i * B + b)];
  }
  out.finishReason = FinishReason::Length;
  return out;
}

GPTOutput GPTEngine::generateAsync(const std::vector<int32_t>& prompt, const GenerateCallback& callback) {
  GPTOutput out;
  if (!prepared_) { fail("generateAsync: engine not prepared"); return out; }
  int64_t S = 0;
  std::vector<int64_t> ids = alignPrompts({prompt}, 0, S);
  if (S == 0) { fail("generateAsync: the prompt is empty (nothing to prefill)"); return out; }
  const tgx_sampler_cfg sc = to_c(config_.samplerConfig);
  const auto t0 = std::chrono::steady_clock::now();
  if (be_.forward(model_.ctx, ids.data(), 1, (int)S) != TGX_OK) { fail(std::string("forward: ") + be_.last_error(model_.ctx)); return out; }
  int64_t cur = 0;
  if (be_.sample(model_.ctx, &sc, config_.seed, &cur) != TGX_OK) { fail(std::string("sample: ") + be_.last_error(model_.ctx)); return out; }
  const auto t1 = std::chrono::steady_clock::now();
  std::vector<int32_t> tokens;
  for (int64_t i = 0; i < S; i++) tokens.push_back((int32_t)ids[(size_t)i]);
  tokens.push_back((int32_t)cur);

  bool hitEos = false, aborted = false, broke = false;
  const bool pipelined = be_.step_async && be_.fetch_token;
  // ticket 0 names the token the last tgx_sample produced (T1); ticket k the token of the k-th step issued since
  int64_t ticket_cur = 0, pending = cur;
  bool have_pending_ticket = false;
  for (int64_t i = 1; i < config_.maxNewTokens; i++) {
    // submitToken(cur); futureToken = genNextToken(cur); tokenId = fetchTokenId()   (GPTEngine.cpp:197-200):
    // the next step is enqueued BEFORE the current id is read back, so the 4-byte read overlaps its compute.
    int32_t tokenId;
    int64_t ticket_next = 0, future = 0;
    if (pipelined) {
      if (be_.step_async(model_.ctx, &sc, config_.seed, &ticket_next) != TGX_OK) { fail(std::string("step: ") + be_.last_error(model_.ctx)); broke = true; break; }
      if (be_.fetch_token(model_.ctx, ticket_cur, &tokenId) != TGX_OK) { fail(std::string("fetch: ") + be_.last_error(model_.ctx)); broke = true; break; }
    } else {
      tokenId = (int32_t)pending;
      if (be_.decode(model_.ctx, &sc, config_.seed, 1, &future) != TGX_OK) { fail(std::string("decode: ") + be_.last_error(model_.ctx)); broke = true; break; }
    }
    if (i > 1) tokens.push_back(tokenId);      // the reference appended it at the end of the previous iteration (:215-216)
    if (isEosToken(tokenId)) { hitEos = true; break; }
    if (callback && !callback(tokenId)) { aborted = true; break; }
    ticket_cur = ticket_next;
    pending = future;
    have_pending_ticket = true;
  }
  if (!hitEos && !aborted && !broke && have_pending_ticket) {
    // loop ended by length: the last futureToken is part of the token list but was never reported
    int32_t last = (int32_t)pending;
    if (pipelined && be_.fetch_token(model_.ctx, ticket_cur, &last) != TGX_OK) fail(std::string("fetch: ") + be_.last_error(model_.ctx));
    tokens.push_back(last);
  }
  out.firstTokenMs = std::chrono::duration<double, std::milli>(t1 - t0).count();
  out.decodeMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
  out.batch = 1;
  out.newTokens = (int64_t)tokens.size() - S;
  out.tokenIds = std::move(tokens);
  out.finishReason = (hitEos || aborted) ? FinishReason::Stop : FinishReason::Length;
  return out;
}

// ---- text entry points ---------------------------------------------------------------------------------------------
int32_t GPTEngine::padTokenId() const {
  if (tokenizerOk_ && tokenizer_.padTokenId() >= 0) return tokenizer_.padTokenId();
  if (tokenizerOk_ && tokenizer_.eosTokenId() >= 0) return tokenizer_.eosTokenId();
  return 0;                                              // default pad token id (GPTEngine.cpp:112)
}

GPTOutput GPTEngine::generateSync(const std::vector<std::string>& texts) {
  if (!tokenizerOk_) { fail("generateSync(texts): no tokenizer loaded"); return GPTOutput(); }
  const std::vector<std::vector<int32_t>> prompts = tokenizer_.encodeBatch(texts);
  GPTOutput out = generateSync(prompts, padTokenId());
  if (out.batch > 0) out.texts = tokenizer_.decodeBatch(out.tokenIds, (uint32_t)out.batch, (uint32_t)(out.tokenIds.size() / (size_t)out.batch - (size_t)out.newTokens));
  return out;
}

GPTOutput GPTEngine::generateAsync(const std::string& text, const TextCallback& callback) {
  if (!tokenizerOk_) { fail("generateAsync(text): no tokenizer loaded"); return GPTOutput(); }
  tokenizer_.decodeStreamFlush();                        // a fresh stream
  bool aborted = false;
  const std::vector<int32_t> prompt = tokenizer_.encode(text);
  GPTOutput out = generateAsync(prompt, [&](int32_t id) {
    const std::string chunk = tokenizer_.decodeStream({id});
    if (!chunk.empty() && callback && !callback(chunk)) { aborted = true; return false; }
    return true;
  });
  if (!aborted) {                                        // flush bytes of an unfinished character (:219-227)
    const std::string rest = tokenizer_.decodeStreamFlush();
    if (!rest.empty() && callback) callback(rest);
  }
  if (out.batch > 0) out.texts = tokenizer_.decodeBatch(out.tokenIds, 1, (uint32_t)(out.tokenIds.size() - (size_t)out.newTokens));
  return out;
}

// ---------------------------------------------------------------------------------------------------------------
// synthetic checkpoints (bit-identical to tinygpt_amd/synth.py)
// ---------------------------------------------------------------------------------------------------------------
namespace {

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline uint64_t fnv1a64(const std::string& s) {
  uint64_t h = 0xCBF29CE484222325ull;
  for (unsigned char b : s) h = (h ^ b) * 0x100000001B3ull;
  return h;
}
inline uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
bool ends_with(const std::string& s, const char* suf) { size_t n = strlen(suf); return s.size() >= n && !s.compare(s.size() - n, n, suf); }

}  // namespace

void synth_tensor_bf16(uint64_t seed, const std::string& name, size_t n, double std_dev, uint16_t* out) {
  const uint64_t base = fnv1a64(name) ^ (seed * 0xD1342543DE82EF95ull);
  // leaf module name decides the distribution: norm weights are 1 + U(-0.1, 0.1)
  std::string leaf, last;
  {
    size_t k = name.find_last_of('.');
    last = k == std::string::npos ? name : name.substr(k + 1);
    std::string rest = k == std::string::npos ? "" : name.substr(0, k);
    size_t k2 = rest.find_last_of('.');
    leaf = k2 == std::string::npos ? rest : rest.substr(k2 + 1);
  }
  const bool is_norm = last == "weight" && (ends_with(leaf, "norm") || ends_with(leaf, "layernorm") || leaf == "ln_1" || leaf == "ln_2" || leaf == "ln_f");
  const float lo = is_norm ? 1.0f : 0.0f;
  const float a = is_norm ? 0.1f : (float)(std_dev * 1.7320508);
  const float two_a = 2.0f * a;
  const unsigned nthreads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
  auto work = [&](size_t s, size_t e) {
    for (size_t i = s; i < e; i++) {
      const uint64_t z = splitmix64(base + (uint64_t)i * 0x9E3779B97F4A7C15ull);
      const float u24 = (float)(z >> 40);
      const float f = (u24 * 5.9604644775390625e-08f - 0.5f) * two_a + lo;
      out[i] = f32_to_bf16(f);
    }
  };
  if (n < (1u << 20) || nthreads == 1) { work(0, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n + nthreads - 1) / nthreads;
  for (unsigned t = 0; t < nthreads; t++) { size_t s = t * per, e = std::min(n, s + per); if (s < e) th.emplace_back(work, s, e); }
  for (auto& t : th) t.join();
}

bool known_config(const std::string& key, int compute_dtype, int max_batch, ModelConfig& out) {
  struct K { const char* name; int fam, H, L, nh, nkv, I, V, tied, bias, ctx; float eps, theta; int scaled; };
  static const K table[] = {
      {"llama-3.2-1b", TGX_FAMILY_LLAMA, 2048, 16, 32, 8, 8192, 128256, 1, 0, 131072, 1e-5f, 500000.f, 1},
      {"llama-3.2-3b", TGX_FAMILY_LLAMA, 3072, 28, 24, 8, 8192, 128256, 1, 0, 131072, 1e-5f, 500000.f, 1},
      {"qwen2.5-0.5b", TGX_FAMILY_QWEN2, 896, 24, 14, 2, 4864, 151936, 1, 1, 32768, 1e-6f, 1000000.f, 0},
      {"mistral-7b-v0.3", TGX_FAMILY_MISTRAL, 4096, 32, 32, 8, 14336, 32768, 0, 0, 32768, 1e-5f, 1000000.f, 0},
      {"qwen2.5-3b", TGX_FAMILY_QWEN2, 2048, 36, 16, 2, 11008, 151936, 1, 1, 32768, 1e-6f, 1000000.f, 0},
      {"llama-3.1-70b", TGX_FAMILY_LLAMA, 8192, 80, 64, 8, 28672, 128256, 0, 0, 131072, 1e-5f, 500000.f, 2},
  };
  if (key == "gpt2") {   // GPT-2 124M (BASELINE.json configs[0]): head = wte, n_ctx = n_positions = 1024
    out = ModelConfig();
    tgx_model_desc& d = out.desc;
    d.family = TGX_FAMILY_GPT2; d.hidden = 768; d.layers = 12; d.heads = d.kv_heads = 12; d.head_dim = 64; d.inter = 3072; d.vocab = 50257;
    d.max_ctx = 1024; d.n_positions = 1024; d.qkv_bias = 1; d.tied = 1; d.compute_dtype = compute_dtype; d.norm_eps = 1e-5f;
    d.max_batch = max_batch < 1 ? 1 : max_batch;
    out.model_type = "gpt2";
    return true;
  }
  if (key == "qwen3-1.7b") {
    out = ModelConfig();
    tgx_model_desc& d = out.desc;
    d.family = TGX_FAMILY_QWEN3; d.hidden = 2048; d.layers = 28; d.heads = 16; d.kv_heads = 8; d.head_dim = 128; d.inter = 6144; d.vocab = 151936;
    d.max_ctx = 40960; d.tied = 1; d.qk_norm = 1; d.compute_dtype = compute_dtype; d.norm_eps = 1e-6f; d.rope_theta = 1000000.f;
    d.max_batch = max_batch < 1 ? 1 : max_batch;
    out.model_type = "qwen3";
    return true;
  }
  if (key == "qwen3-0.6b") {   // explicit head_dim (q_dim 2048 != hidden 1024), per-head q/k RMSNorm (ModelQwen3.h:23-40)
    out = ModelConfig();
    tgx_model_desc& d = out.desc;
    d.family = TGX_FAMILY_QWEN3; d.hidden = 1024; d.layers = 28; d.heads = 16; d.kv_heads = 8; d.head_dim = 128; d.inter = 3072; d.vocab = 151936;
    d.max_ctx = 40960; d.tied = 1; d.qk_norm = 1; d.compute_dtype = compute_dtype; d.norm_eps = 1e-6f; d.rope_theta = 1000000.f;
    d.max_batch = max_batch < 1 ? 1 : max_batch;
    out.model_type = "qwen3";
    return true;
  }
  for (const K& k : table) {
    if (key != k.name) continue;
    out = ModelConfig();
    tgx_model_desc& d = out.desc;
    d.family = k.fam; d.hidden = k.H; d.layers = k.L; d.heads = k.nh; d.kv_heads = k.nkv; d.head_dim = k.H / k.nh;
    d.inter = k.I; d.vocab = k.V; d.max_ctx = k.ctx; d.qkv_bias = k.bias; d.tied = k.tied; d.compute_dtype = compute_dtype;
    d.norm_eps = k.eps; d.rope_theta = k.theta; d.max_batch = max_batch < 1 ? 1 : max_batch;
    if (k.scaled) { d.rope_factor = k.scaled == 2 ? 8.f : 32.f; d.rope_high_freq = 4.f; d.rope_low_freq = 1.f; d.rope_orig_ctx = 8192; d.max_ctx = 8192; }
    out.model_type = k.fam == TGX_FAMILY_LLAMA ? "llama" : k.fam == TGX_FAMILY_QWEN2 ? "qwen2" : "mistral";
    return true;
  }
  return false;
}

bool load_synthetic(const Backend& be, const ModelConfig& cfg, int device_ordinal, uint64_t seed, double std_dev, tgx_ctx** ctx, std::string& err) {
  const tgx_model_desc& d = cfg.desc;
  if (be.create(&d, device_ordinal, ctx) != TGX_OK) { err = std::string("create failed: ") + be.last_error(*ctx); if (*ctx) be.destroy(*ctx); *ctx = nullptr; return false; }
  const int64_t H = d.hidden, I = d.inter, V = d.vocab, qd = (int64_t)d.heads * d.head_dim, kvd = (int64_t)d.kv_heads * d.head_dim;
  std::vector<uint16_t> buf;
  auto put = [&](const std::string& name, int64_t r, int64_t c) -> bool {
    const size_t n = (size_t)(c < 0 ? r : r * c);
    buf.resize(n);
    synth_tensor_bf16(seed, name, n, std_dev, buf.data());
    int64_t shape[2] = {r, c};
    if (be.upload(*ctx, name.c_str(), buf.data(), shape, c < 0 ? 1 : 2, TGX_BF16) != TGX_OK) { err = be.last_error(*ctx); return false; }
    return true;
  };
  if (d.family == TGX_FAMILY_GPT2) {   // hub layout, Conv1D weights [in][out] (ModelGPT2.h:26,226); same names and shapes as desc.py
    bool ok = put("wte.weight", V, H) && put("wpe.weight", d.n_positions, H);
    for (int l = 0; ok && l < d.layers; l++) {
      const std::string p = "h." + std::to_string(l) + ".";
      ok = put(p + "ln_1.weight", H, -1) && put(p + "ln_1.bias", H, -1) && put(p + "attn.c_attn.weight", H, 3 * H) && put(p + "attn.c_attn.bias", 3 * H, -1) &&
           put(p + "attn.c_proj.weight", H, H) && put(p + "attn.c_proj.bias", H, -1) && put(p + "ln_2.weight", H, -1) && put(p + "ln_2.bias", H, -1) &&
           put(p + "mlp.c_fc.weight", H, I) && put(p + "mlp.c_fc.bias", I, -1) && put(p + "mlp.c_proj.weight", I, H) && put(p + "mlp.c_proj.bias", H, -1);
    }
    ok = ok && put("ln_f.weight", H, -1) && put("ln_f.bias", H, -1);
    if (ok && be.finalize(*ctx) != TGX_OK) { err = be.last_error(*ctx); ok = false; }
    if (!ok) { be.destroy(*ctx); *ctx = nullptr; }
    return ok;
  }
  bool ok = put("model.embed_tokens.weight", V, H);
  for (int l = 0; ok && l < d.layers; l++) {
    const std::string p = "model.layers." + std::to_string(l) + ".";
    ok = put(p + "input_layernorm.weight", H, -1) && put(p + "self_attn.q_proj.weight", qd, H) && put(p + "self_attn.k_proj.weight", kvd, H) &&
         put(p + "self_attn.v_proj.weight", kvd, H);
    if (ok && d.qkv_bias) ok = put(p + "self_attn.q_proj.bias", qd, -1) && put(p + "self_attn.k_proj.bias", kvd, -1) && put(p + "self_attn.v_proj.bias", kvd, -1);
    if (ok && d.qk_norm) ok = put(p + "self_attn.q_norm.weight", d.head_dim, -1) && put(p + "self_attn.k_norm.weight", d.head_dim, -1);
    ok = ok && put(p + "self_attn.o_proj.weight", H, qd) && put(p + "post_attention_layernorm.weight", H, -1) &&
         put(p + "mlp.gate_proj.weight", I, H) && put(p + "mlp.up_proj.weight", I, H) && put(p + "mlp.down_proj.weight", H, I);
  }
  ok = ok && put("model.norm.weight", H, -1);
  if (ok && !d.tied) ok = put("lm_head.weight", V, H);
  if (ok && be.finalize(*ctx) != TGX_OK) { err = be.last_error(*ctx); ok = false; }
  if (!ok) { be.destroy(*ctx); *ctx = nullptr; }
  return ok;
}

}  // namespace tgxh
