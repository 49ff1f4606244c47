// engine_c.cpp — a C view of the host engine so that tests (ctypes) and other hosts can drive it.
#include <cstring>

#include "engine.h"
#include "tokenizer.h"

#define TGXE_API extern "C" __attribute__((visibility("default")))

using tgxh::GPTEngine;

struct tgxe_engine {
  GPTEngine* e = nullptr;
  std::string err;
};

TGXE_API tgxe_engine* tgxe_create2(const char* model_dir, const char* synthetic, const char* device, const char* backend_lib,
                                   const char* prefix, int device_ordinal, int dtype, int max_batch, const char* tokenizer_dir);
TGXE_API tgxe_engine* tgxe_create(const char* model_dir, const char* synthetic, const char* device, const char* backend_lib,
                                  const char* prefix, int device_ordinal, int dtype, int max_batch) {
  return tgxe_create2(model_dir, synthetic, device, backend_lib, prefix, device_ordinal, dtype, max_batch, nullptr);
}
TGXE_API tgxe_engine* tgxe_create2(const char* model_dir, const char* synthetic, const char* device, const char* backend_lib,
                                   const char* prefix, int device_ordinal, int dtype, int max_batch, const char* tokenizer_dir) {
  tgxh::GPTConfig c;
  c.modelDir = model_dir ? model_dir : "";
  c.synthetic = synthetic ? synthetic : "";
  c.device = device ? device : "mi355x";
#ifdef TGXH_TEST_HOOKS
  c.backendLib = backend_lib ? backend_lib : "";
  if (prefix && *prefix) c.backendPrefix = prefix;
#else
  (void)prefix;
  if (backend_lib && *backend_lib) return nullptr;     // the shipped library binds libtgx_mi355x.so only
#endif
  c.deviceOrdinal = device_ordinal;
  c.dtype = dtype;
  c.maxBatch = max_batch;
  if (tokenizer_dir) c.tokenizerDir = tokenizer_dir;
  auto* h = new tgxe_engine();
  h->e = new GPTEngine(c);
  return h;
}
TGXE_API void tgxe_destroy(tgxe_engine* h) { if (h) { delete h->e; delete h; } }
TGXE_API int tgxe_prepare(tgxe_engine* h) { return h && h->e->prepare() ? 0 : 1; }
TGXE_API const char* tgxe_last_error(tgxe_engine* h) { return h ? h->e->lastError().c_str() : "null engine"; }
TGXE_API int64_t tgxe_context_size(tgxe_engine* h) { return h ? h->e->contextSize() : -1; }
TGXE_API int tgxe_eos_ids(tgxe_engine* h, int32_t* out, int cap) {
  const auto& v = h->e->eosTokenIds();
  for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[(size_t)i];
  return (int)v.size();
}
TGXE_API void tgxe_reconfigure(tgxe_engine* h, float temperature, int64_t top_k, float top_p, float min_p, int64_t max_new,
                               const int32_t* extra_stop, int n_extra) {
  tgxh::SamplerConfig s; s.temperature = temperature; s.topK = top_k; s.topP = top_p; s.minP = min_p;
  std::vector<int32_t> ex(extra_stop, extra_stop + (extra_stop ? n_extra : 0));
  h->e->reconfigure(s, max_new, ex);
}
// prompts: flat ids + per-row lengths.  out_ids capacity must be >= batch * (min(maxlen, ctx) + max_new).
TGXE_API int tgxe_generate_sync(tgxe_engine* h, const int32_t* flat, const int32_t* lens, int batch, int32_t pad,
                                int32_t* out_ids, int64_t cap, int64_t* out_n, int64_t* out_new, int* out_finish) {
  std::vector<std::vector<int32_t>> p((size_t)batch);
  size_t o = 0;
  for (int b = 0; b < batch; b++) { p[(size_t)b].assign(flat + o, flat + o + lens[b]); o += (size_t)lens[b]; }
  tgxh::GPTOutput r = h->e->generateSync(p, pad);
  if (r.batch == 0) return 1;
  if ((int64_t)r.tokenIds.size() > cap) return 2;
  memcpy(out_ids, r.tokenIds.data(), r.tokenIds.size() * 4);
  *out_n = (int64_t)r.tokenIds.size(); *out_new = r.newTokens; *out_finish = r.finishReason == tgxh::FinishReason::Stop ? 0 : 1;
  return 0;
}
typedef int (*tgxe_token_cb)(int32_t token, void* user);
TGXE_API int tgxe_generate_async(tgxe_engine* h, const int32_t* ids, int len, tgxe_token_cb cb, void* user,
                                 int32_t* out_ids, int64_t cap, int64_t* out_n, int64_t* out_new, int* out_finish) {
  std::vector<int32_t> p(ids, ids + len);
  tgxh::GPTOutput r = h->e->generateAsync(p, [&](int32_t t) { return cb ? cb(t, user) != 0 : true; });
  if (r.batch == 0) return 1;
  if ((int64_t)r.tokenIds.size() > cap) return 2;
  memcpy(out_ids, r.tokenIds.data(), r.tokenIds.size() * 4);
  *out_n = (int64_t)r.tokenIds.size(); *out_new = r.newTokens; *out_finish = r.finishReason == tgxh::FinishReason::Stop ? 0 : 1;
  return 0;
}
// text entry points.  texts: `batch` NUL-terminated UTF-8 strings.  out_text receives the decoded new tokens of every row joined
// by '\x1e' (record separator); out_ids as in tgxe_generate_sync.
TGXE_API int tgxe_generate_sync_text(tgxe_engine* h, const char* const* texts, int batch, int32_t* out_ids, int64_t cap, int64_t* out_n,
                                     int64_t* out_new, char* out_text, int64_t text_cap, int64_t* out_text_len) {
  std::vector<std::string> t;
  for (int b = 0; b < batch; b++) t.emplace_back(texts[b]);
  tgxh::GPTOutput r = h->e->generateSync(t);
  if (r.batch == 0) return 1;
  if ((int64_t)r.tokenIds.size() > cap) return 2;
  memcpy(out_ids, r.tokenIds.data(), r.tokenIds.size() * 4);
  *out_n = (int64_t)r.tokenIds.size(); *out_new = r.newTokens;
  std::string joined;
  for (size_t b = 0; b < r.texts.size(); b++) { if (b) joined += '\x1e'; joined += r.texts[b]; }
  *out_text_len = (int64_t)joined.size();
  if ((int64_t)joined.size() > text_cap) return 2;
  memcpy(out_text, joined.data(), joined.size());
  return 0;
}
typedef int (*tgxe_text_cb)(const char* chunk, int64_t len, void* user);
TGXE_API int tgxe_generate_async_text(tgxe_engine* h, const char* text, tgxe_text_cb cb, void* user, int32_t* out_ids, int64_t cap,
                                      int64_t* out_n, int64_t* out_new, int* out_finish) {
  tgxh::GPTOutput r = h->e->generateAsync(std::string(text), [&](const std::string& c) { return cb ? cb(c.data(), (int64_t)c.size(), user) != 0 : true; });
  if (r.batch == 0) return 1;
  if ((int64_t)r.tokenIds.size() > cap) return 2;
  memcpy(out_ids, r.tokenIds.data(), r.tokenIds.size() * 4);
  *out_n = (int64_t)r.tokenIds.size(); *out_new = r.newTokens; *out_finish = r.finishReason == tgxh::FinishReason::Stop ? 0 : 1;
  return 0;
}
TGXE_API void tgxe_synth_tensor(uint64_t seed, const char* name, int64_t n, double std_dev, uint16_t* out) {
  tgxh::synth_tensor_bf16(seed, name, (size_t)n, std_dev, out);
}

// ---- tokenizer (tokenizer.h) -----------------------------------------------------------------------------------
struct tgxe_tokenizer {
  tgxh::Tokenizer t;
  std::string scratch;
};
TGXE_API tgxe_tokenizer* tgxe_tok_create(const char* tokenizer_json, const char* tokenizer_config_json, char* err, int err_cap) {
  auto* h = new tgxe_tokenizer();
  if (h->t.initWithConfig(tokenizer_json ? tokenizer_json : "", tokenizer_config_json ? tokenizer_config_json : "")) return h;
  if (err && err_cap > 0) { strncpy(err, h->t.lastError().c_str(), (size_t)err_cap - 1); err[err_cap - 1] = 0; }
  delete h;
  return nullptr;
}
TGXE_API void tgxe_tok_destroy(tgxe_tokenizer* h) { delete h; }
// returns the number of ids (written up to cap)
TGXE_API int64_t tgxe_tok_encode(tgxe_tokenizer* h, const char* text, int64_t len, int allow_added, int32_t* out, int64_t cap) {
  const std::vector<int32_t> ids = h->t.encode(std::string(text, (size_t)len), allow_added != 0);
  for (int64_t i = 0; i < (int64_t)ids.size() && i < cap; i++) out[i] = ids[(size_t)i];
  return (int64_t)ids.size();
}
// mode 0: decode(ids); 1: decodeStream(ids); 2: decodeStreamFlush().  Returns the byte length; bytes are copied up to cap.
TGXE_API int64_t tgxe_tok_decode(tgxe_tokenizer* h, const int32_t* ids, int64_t n, int mode, char* out, int64_t cap) {
  const std::vector<int32_t> v(ids, ids + (ids ? n : 0));
  h->scratch = mode == 0 ? h->t.decode(v) : mode == 1 ? h->t.decodeStream(v) : h->t.decodeStreamFlush();
  const int64_t len = (int64_t)h->scratch.size();
  if (out && cap > 0) memcpy(out, h->scratch.data(), (size_t)(len < cap ? len : cap));
  return len;
}
// the bytes produced by the last tgxe_tok_decode call (the stream modes are stateful: read their result from here)
TGXE_API int64_t tgxe_tok_scratch(tgxe_tokenizer* h, char* out, int64_t cap) {
  const int64_t len = (int64_t)h->scratch.size();
  if (out && cap > 0) memcpy(out, h->scratch.data(), (size_t)(len < cap ? len : cap));
  return len;
}
TGXE_API int32_t tgxe_tok_special(tgxe_tokenizer* h, int which) { return which == 0 ? h->t.bosTokenId() : which == 1 ? h->t.eosTokenId() : h->t.padTokenId(); }
TGXE_API int32_t tgxe_tok_token_to_id(tgxe_tokenizer* h, const char* token) { return h->t.token2Id(token ? token : ""); }
// regex matchAll for the pre-tokenizer tests: writes (begin, end) byte offsets, returns the match count or -1 for an invalid pattern
TGXE_API int64_t tgxe_regex_match_all(const char* pattern, const char* text, int64_t len, int64_t* out, int64_t cap_pairs) {
  tgxh::Regex re(pattern);
  if (!re.valid()) return -1;
  std::vector<tgxh::Range> m;
  re.matchAll(std::string(text, (size_t)len), m);
  for (int64_t i = 0; i < (int64_t)m.size() && i < cap_pairs; i++) { out[2 * i] = (int64_t)m[(size_t)i].first; out[2 * i + 1] = (int64_t)m[(size_t)i].second; }
  return (int64_t)m.size();
}

// Unicode normalisation for the normalizer tests: form 0 NFC, 1 NFD, 2 NFKC, 3 NFKD; returns the byte length, copies up to cap
TGXE_API int64_t tgxe_normalize(int form, const char* text, int64_t len, char* out, int64_t cap) {
  const std::string r = tgxh::normalize_unicode(std::string(text, (size_t)len), form == 0 || form == 2, form >= 2);
  if (out && cap > 0) memcpy(out, r.data(), (size_t)((int64_t)r.size() < cap ? (int64_t)r.size() : cap));
  return (int64_t)r.size();
}
