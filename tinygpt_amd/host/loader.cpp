#include "loader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <sstream>

#include "json.h"

namespace tgxh {

namespace {

bool read_file(const std::string& path, std::string& out) {
  std::ifstream ifs(path, std::ios::binary);
  if (!ifs.is_open()) return false;
  std::stringstream ss;
  ss << ifs.rdbuf();
  out = ss.str();
  return true;
}

bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

std::string join_path(const std::string& a, const std::string& b) { return (a.empty() || a.back() == '/') ? a + b : a + "/" + b; }
std::string base_dir(const std::string& p) { size_t k = p.find_last_of('/'); return k == std::string::npos ? "." : p.substr(0, k); }

int family_of(const std::string& t) {
  if (t == "gpt2") return TGX_FAMILY_GPT2;
  if (t == "llama") return TGX_FAMILY_LLAMA;
  if (t == "qwen2") return TGX_FAMILY_QWEN2;
  if (t == "qwen3") return TGX_FAMILY_QWEN3;
  if (t == "mistral") return TGX_FAMILY_MISTRAL;
  return 0;
}

}  // namespace

// every dimension the device shim allocates from must be a positive 32-bit integer (a missing key or a value of the wrong JSON type reads as -1)
static bool check_dims(const tgx_model_desc& d, const std::string& path, std::string& err) {
  struct { const char* name; int64_t v; } f[] = {{"hidden size", d.hidden}, {"layer count", d.layers}, {"attention heads", d.heads}, {"key/value heads", d.kv_heads},
                                                 {"head_dim", d.head_dim}, {"intermediate size", d.inter}, {"vocab_size", d.vocab}, {"context size", d.max_ctx}};
  for (auto& e : f)
    if (e.v <= 0 || e.v > (int64_t)1 << 30) { err = "config.json: " + std::string(e.name) + " is missing, not an integer or out of range in " + path; return false; }
  return true;
}

bool load_model_config(const std::string& path, int compute_dtype, int max_batch, ModelConfig& out, std::string& err) {
  std::string text;
  if (!read_file(path, text) || text.empty()) { err = "Failed to open file: " + path; return false; }
  Json doc;
  if (!JsonParser::parse(text.data(), text.size(), doc) || doc.kind != Json::Obj) { err = "JSON parse error in: " + path; return false; }
  out = ModelConfig();
  out.model_type = doc.get_str("model_type", "");
  if (out.model_type.empty()) { err = "Missing or invalid model_type in config.json"; return false; }
  const int fam = family_of(out.model_type);
  if (!fam) { err = "Unsupported model_type: " + out.model_type; return false; }
  tgx_model_desc& d = out.desc;
  d.family = fam;
  d.compute_dtype = compute_dtype;
  d.max_batch = max_batch < 1 ? 1 : max_batch;
  // 64-bit values are range-checked before narrowing (2^32 + 5 must not read as 5)
  auto dim = [](const Json& j, const char* key, int64_t def) -> int32_t { const int64_t v = j.get_int(key, def); return (v < 0 || v > ((int64_t)1 << 30)) ? -1 : (int32_t)v; };
  out.torch_dtype = doc.get_str("torch_dtype", doc.get_str("dtype", ""));
  out.bos_token_id = doc.get_int("bos_token_id", -1);
  out.eos_token_id = doc.get_int("eos_token_id", -1);
  d.vocab = dim(doc, "vocab_size", -1);
  if (fam == TGX_FAMILY_GPT2) {
    d.hidden = dim(doc, "n_embd", -1);
    d.layers = dim(doc, "n_layer", -1);
    d.heads = d.kv_heads = dim(doc, "n_head", -1);
    d.head_dim = d.heads > 0 ? d.hidden / d.heads : 0;
    d.inter = 4 * d.hidden;
    d.n_positions = dim(doc, "n_positions", -1);
    d.max_ctx = dim(doc, "n_ctx", d.n_positions);      // contextSize = n_ctx (ModelGPT2.h:230)
    d.norm_eps = doc.get_float("layer_norm_epsilon", 1e-5f);
    d.qkv_bias = 1; d.tied = 1;
    return check_dims(d, path, err);
  }
  d.hidden = dim(doc, "hidden_size", -1);
  d.layers = dim(doc, "num_hidden_layers", -1);
  d.heads = dim(doc, "num_attention_heads", -1);
  d.kv_heads = dim(doc, "num_key_value_heads", d.heads);
  d.inter = dim(doc, "intermediate_size", -1);
  d.max_ctx = dim(doc, "max_position_embeddings", -1);
  d.norm_eps = doc.get_float("rms_norm_eps", 1e-5f);
  d.tied = doc.get_bool("tie_word_embeddings", false) ? 1 : 0;
  d.head_dim = d.heads > 0 ? d.hidden / d.heads : 0;               // ModelLlama.h:37 ignores "head_dim"
  if (fam == TGX_FAMILY_QWEN3) d.head_dim = dim(doc, "head_dim", d.head_dim);   // ModelQwen3.h:25
  d.qkv_bias = fam == TGX_FAMILY_QWEN2 ? 1 : 0;                    // ModelQwen2.h:26-31
  d.qk_norm = fam == TGX_FAMILY_QWEN3 ? 1 : 0;                     // AttentionWithQKNorm (ModelQwen3.h:29-33)
  // rope: hub-era flat keys (what the reference parses) or the nested rope_parameters newer transformers write
  const Json* rp = doc.get("rope_parameters");
  const Json* rs = doc.get("rope_scaling");
  if ((!rs || rs->kind != Json::Obj) && rp && rp->kind == Json::Obj && rp->get_str("rope_type", "default") != "default") rs = rp;
  const float def_theta = fam == TGX_FAMILY_LLAMA ? 1.f : 10000.f;  // ModelConfig.cpp:88,91,99
  float theta = doc.get_float("rope_theta", -1.f);
  if (theta < 0.f) theta = (rp && rp->kind == Json::Obj) ? rp->get_float("rope_theta", def_theta) : def_theta;
  d.rope_theta = theta;
  if (fam == TGX_FAMILY_LLAMA && rs && rs->kind == Json::Obj) {     // ModelConfig.cpp:79-87
    d.rope_factor = rs->get_float("factor", 1.f);
    d.rope_high_freq = rs->get_float("high_freq_factor", 1.f);
    d.rope_low_freq = rs->get_float("low_freq_factor", 1.f);
    d.rope_orig_ctx = dim(*rs, "original_max_position_embeddings", -1);
    if (d.rope_orig_ctx > 0) d.max_ctx = d.rope_orig_ctx;           // getContextSize (ModelLlama.h:26-31)
  }
  return check_dims(d, path, err);
}

bool load_generation_config(const std::string& path, GenerationConfig& out, std::string& err) {
  std::string text;
  if (!read_file(path, text) || text.empty()) { err = "Failed to open file: " + path; return false; }
  Json doc;
  if (!JsonParser::parse(text.data(), text.size(), doc) || doc.kind != Json::Obj) { err = "JSON parse error in: " + path; return false; }
  out = GenerationConfig();
  out.bos_token_id = doc.get_int("bos_token_id", -1);
  if (const Json* e = doc.get("eos_token_id")) {
    if (e->kind == Json::Arr) { for (const Json& v : e->arr) if (v.kind == Json::Num && v.is_int) out.eos_token_ids.push_back(v.i); }
    else if (e->kind == Json::Num && e->is_int) out.eos_token_ids.push_back(e->i);
  }
  out.do_sample = doc.get_bool("do_sample", false);
  out.temperature = doc.get_float("temperature", 0.f);
  out.top_k = doc.get_int("top_k", 0);
  out.top_p = doc.get_float("top_p", 1.f);
  return true;
}

namespace {

// one .safetensors file: u64 header size | JSON header | data (SafeTensors.cpp:141-229)
bool load_one(const Backend& be, tgx_ctx* ctx, const std::string& path, const std::set<std::string>* only, int& loaded, std::string& err) {
  int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) { err = "Error mapFileForRead: " + path; return false; }
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < 8) { ::close(fd); err = "Error mapFileForRead: " + path; return false; }
  void* map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (map == MAP_FAILED) { err = "Error mapFileForRead: " + path; return false; }
  bool ok = true;
  const uint64_t header_size = *static_cast<const uint64_t*>(map);
  // st_size >= 8 was checked above; written so that a header_size near 2^64 cannot wrap past the check
  if (header_size > (uint64_t)st.st_size - 8) { err = "corrupt safetensors header: " + path; munmap(map, (size_t)st.st_size); return false; }
  const char* header = static_cast<const char*>(map) + 8;
  const char* data = header + header_size;
  const uint64_t data_size = (uint64_t)st.st_size - 8 - header_size;
  Json doc;
  if (!JsonParser::parse(header, (size_t)header_size, doc) || doc.kind != Json::Obj) { err = "safetensors header is not JSON: " + path; ok = false; }
  for (size_t t = 0; ok && t < doc.obj.size(); t++) {
    const std::string& name = doc.obj[t].first;
    if (name == "__metadata__") continue;
    if (only && !only->count(name)) continue;
    const Json& info = doc.obj[t].second;
    const Json* shape = info.get("shape");
    const Json* off = info.get("data_offsets");
    const std::string dt = info.get_str("dtype", "");
    if (!shape || shape->kind != Json::Arr || !off || off->kind != Json::Arr || off->arr.size() != 2) { err = "bad tensor entry: " + name; ok = false; break; }
    int src = -1;
    size_t esz = 0;
    if (dt == "BF16") { src = TGX_BF16; esz = 2; } else if (dt == "F16") { src = TGX_F16; esz = 2; } else if (dt == "F32") { src = TGX_F32; esz = 4; }
    std::vector<int64_t> dims;
    uint64_t numel = 1;
    bool shape_ok = true;
    for (const Json& v : shape->arr) {      // non-negative integers whose product (times the element size) stays inside 64 bits
      if (v.kind != Json::Num || !v.is_int || v.i < 0) { shape_ok = false; break; }
      if (v.i != 0 && numel > (UINT64_MAX / 8) / (uint64_t)v.i) { shape_ok = false; break; }
      dims.push_back(v.i); numel *= (uint64_t)v.i;
    }
    if (!shape_ok || off->arr[0].kind != Json::Num || !off->arr[0].is_int || off->arr[0].i < 0 || off->arr[1].kind != Json::Num || !off->arr[1].is_int || off->arr[1].i < 0) {
      err = "bad tensor entry: " + name; ok = false; break;
    }
    if (src < 0) {
      // a dtype this path never consumes (I64 position_ids, BOOL / U8 mask buffers of old hub checkpoints): the reference looks the name
      // up first and only warns about keys it does not know (SafeTensors.cpp:176-183) — ask the shim whether it wants the tensor
      static const float probe = 0.f;
      static const int64_t no_dims = 0;
      const int prc = be.upload(ctx, name.c_str(), &probe, dims.empty() ? &no_dims : dims.data(), -1, TGX_F32);     // nd = -1: name lookup only, nothing is copied
      if (prc == TGX_ERR_NAME) { fprintf(stderr, "[tgx] Unexpected key: %s\n", name.c_str()); continue; }
      if (prc == TGX_OK) continue;                                                         // a known non-parameter buffer (GPT-2's attn.bias masks)
      err = "dtype not supported for tensor: " + name + " (" + dt + ")"; ok = false; break;
    }
    const uint64_t b0 = (uint64_t)off->arr[0].i, b1 = (uint64_t)off->arr[1].i;
    if (b1 < b0 || b1 > data_size || b1 - b0 != numel * esz) { err = "size not equal for tensor: " + name; ok = false; break; }
    const int rc = be.upload(ctx, name.c_str(), data + b0, dims.data(), (int)dims.size(), src);
    if (rc == TGX_ERR_NAME) { fprintf(stderr, "[tgx] Unexpected key: %s\n", name.c_str()); continue; }   // non-strict load
    if (rc != TGX_OK) { err = std::string(be.last_error(ctx)); ok = false; break; }
    loaded++;
  }
  munmap(map, (size_t)st.st_size);
  return ok;
}

}  // namespace

bool load_safetensors(const Backend& be, tgx_ctx* ctx, const std::string& path, int& loaded, std::string& err) {
  loaded = 0;
  auto ends_with = [](const std::string& s, const std::string& suf) { return suf.size() <= s.size() && !s.compare(s.size() - suf.size(), suf.size(), suf); };
  if (ends_with(path, ".safetensors")) return load_one(be, ctx, path, nullptr, loaded, err);
  if (!ends_with(path, ".index.json")) { err = "Unknown file type: " + path; return false; }
  std::string text;
  if (!read_file(path, text)) { err = "Error open index file: " + path; return false; }
  Json doc;
  if (!JsonParser::parse(text.data(), text.size(), doc) || doc.kind != Json::Obj) { err = "Invalid index json: " + path; return false; }
  const Json* wm = doc.get("weight_map");
  if (!wm || wm->kind != Json::Obj) { err = "Index json missing weight_map"; return false; }
  std::map<std::string, std::set<std::string>> shard2keys;          // SafeTensors.cpp:258-263
  for (const auto& kv : wm->obj) if (kv.second.kind == Json::Str) shard2keys[kv.second.str].insert(kv.first);
  const std::string dir = base_dir(path);
  for (const auto& sk : shard2keys)
    if (!load_one(be, ctx, join_path(dir, sk.first), &sk.second, loaded, err)) { err = "Failed to load shard: " + sk.first + ": " + err; return false; }
  return true;
}

bool load_model_dir(const Backend& be, const std::string& dir, int device_ordinal, int compute_dtype, int max_batch,
                    LoadedModel& out, std::string& err) {
  if (!load_model_config(join_path(dir, "config.json"), compute_dtype, max_batch, out.config, err)) return false;
  if (!load_generation_config(join_path(dir, "generation_config.json"), out.generation, err)) return false;   // required (ModelLoader.cpp:34-38)
  int rc = be.create(&out.config.desc, device_ordinal, &out.ctx);
  if (rc != TGX_OK) { err = std::string("create failed: ") + be.last_error(out.ctx); if (out.ctx) be.destroy(out.ctx); out.ctx = nullptr; return false; }
  std::string mp = join_path(dir, "model.safetensors");
  if (!file_exists(mp)) mp = join_path(dir, "model.safetensors.index.json");          // ModelLoader.cpp:72-75
  int loaded = 0;
  if (!load_safetensors(be, out.ctx, mp, loaded, err)) { be.destroy(out.ctx); out.ctx = nullptr; return false; }
  rc = be.finalize(out.ctx);
  if (rc != TGX_OK) { err = std::string("Load model failed: ") + be.last_error(out.ctx); be.destroy(out.ctx); out.ctx = nullptr; return false; }
  return true;
}

}  // namespace tgxh
