// engine.h — the host engine above the C ABI: the counterpart of GPTEngine (src/engine/GPTEngine.h:42-73,
// GPTEngine.cpp:37-232).  Text entry points (tokenizer.h) and token-id entry points share one path: encode,
// left-padding without a mask, tail truncation to contextSize, prefill + maxNewTokens-1 decode steps, no EOS stop in
// generateSync, one-step-lookahead streaming with EOS/abort and UTF-8-safe chunks in generateAsync, reconfigure()
// resetting the KV cache — the reference's behaviour line by line.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "backend.h"
#include "loader.h"
#include "tokenizer.h"

namespace tgxh {

struct SamplerConfig {   // src/engine/Sampler.h:13-22
  float temperature = 0.f;
  int64_t topK = 0;
  float topP = 1.f;
  float minP = 0.f;
};

enum class FinishReason { Stop, Length };

struct GPTConfig {       // src/engine/GPTEngine.h:25-32 (+ where to find the device shim)
  std::string modelDir;              // HF directory; empty with `synthetic` set
  std::string synthetic;             // "llama-3.2-1b" ...: deterministic synthetic checkpoint of that public config
  std::string device = "mi355x";     // the reference's CLI knows "cpu" and "cuda" (main.cpp:76-80)
  int dtype = TGX_BF16;
  SamplerConfig samplerConfig;
  int64_t maxNewTokens = 16;
  int deviceOrdinal = 0;
  int maxBatch = 4;
  uint64_t seed = 0;
  std::string tokenizerDir;          // tokenizer.json + tokenizer_config.json; default: modelDir (lets --synthetic runs take text)
#ifdef TGXH_TEST_HOOKS
  // Only in the test build (tests/_build/libtgx_host_test.so, tgx_cli_test: -DTGXH_TEST_HOOKS): bind another library that exports the tgx ABI
  // (the CPU oracle) to check host logic without a GPU.  The shipped library and CLI do not contain these fields or the code that reads them:
  // they can only dlopen libtgx_mi355x.so from their own directory.  LAST in the struct: every field above keeps its offset in both builds.
  std::string backendLib;
  std::string backendPrefix = "tgx_";
#endif
};

struct GPTOutput {       // src/engine/GPTEngine.h:34-40
  int64_t batch = 0;
  int64_t newTokens = 0;
  std::vector<int32_t> tokenIds;     // [batch][padded prompt + new], row-major — prompt tokens included, like the reference
  std::vector<std::string> texts;    // the new tokens of every row, decoded (text entry points only)
  FinishReason finishReason = FinishReason::Stop;
  // not in the reference's struct: the generate call split at the first token (SURVEY.md §8 row H asks the harness for decode-only tok/s)
  double firstTokenMs = 0.0;         // encode-to-first-token: prefill + first sample
  double decodeMs = 0.0;             // the remaining newTokens-1 steps
};

using GenerateCallback = std::function<bool(int32_t tokenId)>;   // return false to abort (GPTEngine.cpp:208-213)
using TextCallback = std::function<bool(const std::string& chunk)>;   // the reference's GenerateCallback: complete UTF-8 only

class GPTEngine {
 public:
  explicit GPTEngine(GPTConfig config);
  ~GPTEngine();
  GPTEngine(const GPTEngine&) = delete;
  GPTEngine& operator=(const GPTEngine&) = delete;

  bool prepare();                                                            // GPTEngine.cpp:41-65
  void reconfigure(const SamplerConfig& samplerConfig, int64_t maxNewTokens,
                   const std::vector<int32_t>& extraStopTokenIds = {});      // GPTEngine.cpp:67-84
  GPTOutput generateSync(const std::vector<std::vector<int32_t>>& prompts, int32_t padToken);   // :154-174
  GPTOutput generateAsync(const std::vector<int32_t>& prompt, const GenerateCallback& callback);   // :180-232
  // text entry points (need a tokenizer: tokenizerDir / modelDir must hold tokenizer.json + tokenizer_config.json)
  GPTOutput generateSync(const std::vector<std::string>& texts);                                   // :154-174 incl. encodeTexts :101-144
  GPTOutput generateAsync(const std::string& text, const TextCallback& callback);                  // :180-232 incl. decodeStream
  bool hasTokenizer() const { return tokenizerOk_; }
  Tokenizer& tokenizer() { return tokenizer_; }
  int32_t padTokenId() const;                                                                       // pad -> eos -> 0 (:108-114)

  bool isEosToken(int32_t id) const;
  const std::vector<int32_t>& eosTokenIds() const { return eosTokenIds_; }
  int64_t contextSize() const;
  const tgx_model_desc& desc() const { return model_.config.desc; }
  const std::string& lastError() const { return err_; }
  tgx_ctx* ctx() { return model_.ctx; }
  const Backend& backend() const { return be_; }

 private:
  // == encodeTexts minus the tokenizer (GPTEngine.cpp:101-144): truncate to contextSize keeping the tail, left-pad
  std::vector<int64_t> alignPrompts(const std::vector<std::vector<int32_t>>& prompts, int32_t padToken, int64_t& maxLen) const;
  bool fail(const std::string& what);

  GPTConfig config_;
  Backend be_;
  LoadedModel model_;
  std::vector<int32_t> baseEosTokenIds_, eosTokenIds_;
  std::string err_;
  bool prepared_ = false;
  Tokenizer tokenizer_;
  bool tokenizerOk_ = false;
};

// Deterministic synthetic checkpoint — bit-identical to tinygpt_amd/synth.py (same integer hash).
void synth_tensor_bf16(uint64_t seed, const std::string& name, size_t n, double std_dev, uint16_t* out);
bool known_config(const std::string& key, int compute_dtype, int max_batch, ModelConfig& out);
bool load_synthetic(const Backend& be, const ModelConfig& cfg, int device_ordinal, uint64_t seed, double std_dev, tgx_ctx** ctx, std::string& err);

}  // namespace tgxh
