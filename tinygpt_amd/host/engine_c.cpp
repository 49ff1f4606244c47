// engine_c.cpp — a C view of the host engine so that tests (ctypes) and other hosts can drive it.
#include <cstring>

#include "engine.h"

#define TGXE_API extern "C" __attribute__((visibility("default")))

using tgxh::GPTEngine;

struct tgxe_engine {
  GPTEngine* e = nullptr;
  std::string err;
};

TGXE_API tgxe_engine* tgxe_create(const char* model_dir, const char* synthetic, const char* device, const char* backend_lib,
                                  const char* prefix, int device_ordinal, int dtype, int max_batch) {
  tgxh::GPTConfig c;
  c.modelDir = model_dir ? model_dir : "";
  c.synthetic = synthetic ? synthetic : "";
  c.device = device ? device : "mi355x";
  c.backendLib = backend_lib ? backend_lib : "";
  if (prefix && *prefix) c.backendPrefix = prefix;
  c.deviceOrdinal = device_ordinal;
  c.dtype = dtype;
  c.maxBatch = max_batch;
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
TGXE_API void tgxe_synth_tensor(uint64_t seed, const char* name, int64_t n, double std_dev, uint16_t* out) {
  tgxh::synth_tensor_bf16(seed, name, (size_t)n, std_dev, out);
}
