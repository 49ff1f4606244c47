// loader.h — what ModelLoader::load does before the first forward (src/huggingface/ModelLoader.cpp:25-89):
// config.json -> model description, generation_config.json -> EOS ids, model.safetensors (or its .index.json) ->
// tensors uploaded by HF name.  The tokenizer half of the reference's loader is out of scope (inputs are ids).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/tgx.h"
#include "backend.h"

namespace tgxh {

struct GenerationConfig {   // src/huggingface/ModelConfig.h:81-89
  int64_t bos_token_id = -1;
  std::vector<int64_t> eos_token_ids;
  bool do_sample = false;
  float temperature = 0.f;
  int64_t top_k = 0;
  float top_p = 1.f;
};

struct ModelConfig {
  tgx_model_desc desc{};
  std::string model_type;
  std::string torch_dtype;
  int64_t bos_token_id = -1, eos_token_id = -1;
};

// == loadModelConfig (ModelConfig.cpp:43-125) + the family factories' derived values (ModelLlama.h:21-53, ...).
// compute_dtype is the CLI's --dtype (TGX_F32/TGX_BF16/TGX_F16).  Returns false and fills err on failure.
bool load_model_config(const std::string& path, int compute_dtype, int max_batch, ModelConfig& out, std::string& err);

// == loadGenerationConfig (ModelConfig.cpp:127-164); eos_token_id may be an int or an array.
bool load_generation_config(const std::string& path, GenerationConfig& out, std::string& err);

// == SafeTensors::load (SafeTensors.cpp:124-280): single file or "*.index.json" with shards.  Every tensor in the
// file is offered to tgx_upload by name; "Unexpected key" is a warning (non-strict load, GPTModel.h:96), shape or
// dtype problems fail the load.  `loaded` counts accepted tensors.
bool load_safetensors(const Backend& be, tgx_ctx* ctx, const std::string& path, int& loaded, std::string& err);

// Directory-level load: config + generation config + weights + finalize.  == ModelLoader::load without the tokenizer.
struct LoadedModel {
  ModelConfig config;
  GenerationConfig generation;
  tgx_ctx* ctx = nullptr;
};
bool load_model_dir(const Backend& be, const std::string& dir, int device_ordinal, int compute_dtype, int max_batch,
                    LoadedModel& out, std::string& err);

}  // namespace tgxh
