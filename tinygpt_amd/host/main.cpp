// tgx_cli — the inference harness of the reference (examples/inference/main.cpp): the same four text prompts when a
// tokenizer is available (--model dir or --tokenizer dir), token-id prompts otherwise;
// same flags and defaults (--model --device --dtype --max-tokens --temperature --top-p), the same timing window
// (generate only; load excluded, main.cpp:97-102) and the same "speed" convention (ALL ids incl. prompt / wall time,
// main.cpp:112-114) — plus the new-token rate, the time to first token and the decode-only rate.  `--device mi355x` is the only device this binary executes on.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>

#include "engine.h"

// INPUT_STRS (main.cpp:12-17), and their gpt2 ids for runs without a tokenizer (usable with any vocabulary >= 50257)
static const std::vector<std::string> kInputStrs = {"Hello, my name is", "The president of the United States is", "The capital of France is", "The future of AI is"};
static const std::vector<std::vector<int32_t>> kDefaultPrompts = {
    {15496, 11, 616, 1438, 318}, {464, 1893, 286, 262, 1578, 1829, 318}, {464, 3139, 286, 4881, 318}, {464, 2003, 286, 9552, 318}};

static void usage(const char* prog) {
  fprintf(stderr,
          "Usage: %s [options]\n"
          "  --model <path>            HuggingFace model directory (config.json, generation_config.json, model.safetensors[.index.json])\n"
          "  --synthetic <name>        instead of --model: llama-3.2-1b | llama-3.2-3b | qwen2.5-0.5b | mistral-7b-v0.3 (deterministic weights)\n"
          "  --device <mi355x>         device type (default: mi355x)\n"
          "  --dtype <bf16>            data type (default: bf16)\n"
          "  --max-tokens <n>          max new tokens (default: 32)\n"
          "  --temperature <f>         sampling temperature (default: 0.8)\n"
          "  --top-p <f>               top-p sampling (default: 0.9)\n"
          "  --top-k <n> --min-p <f>   (default: off)\n"
          "  --tokenizer <dir>         tokenizer.json + tokenizer_config.json (default: the --model directory)\n"
          "  --prompt <text>           a text prompt (repeatable; default with a tokenizer: the reference's 4 prompts)\n"
          "  --stream                  batch-1 generateAsync: print UTF-8-safe chunks as they are produced\n"
          "  --prompt-ids <a,b,c;d,e>  prompts as token ids, ';' between batch rows (default without a tokenizer: the 4 prompts as gpt2 ids)\n"
          "  --pad-id <n>              left-pad id (default: eos_token_id of the model, else 0)\n"
          "  --seed <n>                sampler seed (default: 0)\n",
          prog);
}

int main(int argc, char** argv) {
  tgxh::GPTConfig cfg;
  cfg.maxNewTokens = 32;
  cfg.samplerConfig.temperature = 0.8f;
  cfg.samplerConfig.topP = 0.9f;
  std::string dtype = "bf16", prompt_ids;
  long pad_id = -1;
  std::vector<std::string> text_prompts;
  bool stream = false;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--help" || a == "-h") { usage(argv[0]); return 0; }
    else if (a == "--model") cfg.modelDir = next();
    else if (a == "--synthetic") cfg.synthetic = next();
    else if (a == "--device") cfg.device = next();
    else if (a == "--dtype") dtype = next();
    else if (a == "--max-tokens") cfg.maxNewTokens = atoi(next());
    else if (a == "--temperature") cfg.samplerConfig.temperature = strtof(next(), nullptr);
    else if (a == "--top-p") cfg.samplerConfig.topP = strtof(next(), nullptr);
    else if (a == "--top-k") cfg.samplerConfig.topK = atoll(next());
    else if (a == "--min-p") cfg.samplerConfig.minP = strtof(next(), nullptr);
    else if (a == "--prompt-ids") prompt_ids = next();
    else if (a == "--prompt") text_prompts.push_back(next());
    else if (a == "--tokenizer") cfg.tokenizerDir = next();
    else if (a == "--stream") stream = true;
    else if (a == "--pad-id") pad_id = atol(next());
    else if (a == "--seed") cfg.seed = strtoull(next(), nullptr, 10);
#ifdef TGXH_TEST_HOOKS
    // tgx_cli_test only (tests/_build, -DTGXH_TEST_HOOKS): bind the host engine to a library of the test's choice that exports the tgx ABI
    // (the CPU oracle), to check host logic on a machine without a GPU.  The shipped tgx_cli has neither flag.
    else if (a == "--backend-lib") cfg.backendLib = next();
    else if (a == "--backend-prefix") cfg.backendPrefix = next();
#endif
    else { fprintf(stderr, "Unknown argument: %s\n", a.c_str()); usage(argv[0]); return 1; }
  }
  if (cfg.modelDir.empty() && cfg.synthetic.empty()) { fprintf(stderr, "Error: --model (or --synthetic) is required\n"); usage(argv[0]); return 1; }
  cfg.dtype = dtype == "fp32" ? TGX_F32 : dtype == "fp16" ? TGX_F16 : TGX_BF16;

  std::vector<std::vector<int32_t>> prompts = kDefaultPrompts;
  if (!prompt_ids.empty()) {
    prompts.clear();
    std::stringstream rows(prompt_ids);
    std::string row;
    while (std::getline(rows, row, ';')) {
      std::vector<int32_t> ids;
      std::stringstream toks(row);
      std::string t;
      while (std::getline(toks, t, ',')) if (!t.empty()) ids.push_back((int32_t)atol(t.c_str()));
      if (!ids.empty()) prompts.push_back(ids);
    }
  }
  cfg.maxBatch = (int)std::max(std::max(prompts.size(), text_prompts.size()), kInputStrs.size());

  tgxh::GPTEngine engine(cfg);
  if (!engine.prepare()) { fprintf(stderr, "Prepare engine failed\n"); return 1; }

  if (engine.hasTokenizer() && prompt_ids.empty()) {       // the reference's flow: texts in, texts out (main.cpp:97-114)
    if (text_prompts.empty()) text_prompts = kInputStrs;
    const auto t0 = std::chrono::steady_clock::now();
    tgxh::GPTOutput out;
    if (stream) {
      printf("%s", text_prompts[0].c_str());
      out = engine.generateAsync(text_prompts[0], [](const std::string& chunk) { fputs(chunk.c_str(), stdout); fflush(stdout); return true; });
      printf("\n");
    } else out = engine.generateSync(text_prompts);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (out.batch == 0) { fprintf(stderr, "generate failed: %s\n", engine.lastError().c_str()); return 1; }
    if (!stream) {
      printf("Generated Outputs:\n------------------------------------------------------------\n");
      for (int64_t b = 0; b < out.batch; b++)
        printf("Prompt:    '%s'\nOutput:    '%s'\n------------------------------------------------------------\n", text_prompts[(size_t)b].c_str(), out.texts[(size_t)b].c_str());
    }
    printf("Time cost: %lld ms, speed: %.2f token/s\n", (long long)ms, out.tokenIds.size() * 1000.0 / ms);
    printf("new tokens: %lld, new-token rate: %.2f token/s\n", (long long)(out.batch * out.newTokens), out.batch * out.newTokens * 1000.0 / ms);
    if (out.newTokens > 1) printf("time to first token: %.1f ms, decode-only rate: %.2f token/s\n", out.firstTokenMs, out.batch * (out.newTokens - 1) * 1000.0 / out.decodeMs);
    return 0;
  }
  int32_t pad = pad_id >= 0 ? (int32_t)pad_id : (!engine.eosTokenIds().empty() ? engine.eosTokenIds()[0] : 0);
  for (auto& p : prompts) for (auto& t : p) if (t >= engine.desc().vocab) t = t % engine.desc().vocab;

  const auto t0 = std::chrono::steady_clock::now();
  tgxh::GPTOutput out = engine.generateSync(prompts, pad);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (out.batch == 0) { fprintf(stderr, "generate failed: %s\n", engine.lastError().c_str()); return 1; }

  printf("Generated Outputs:\n------------------------------------------------------------\n");
  const int64_t row = (int64_t)out.tokenIds.size() / out.batch;
  for (int64_t b = 0; b < out.batch; b++) {
    printf("Prompt ids: ");
    for (int64_t i = 0; i < row - out.newTokens; i++) printf("%d ", out.tokenIds[(size_t)(b * row + i)]);
    printf("\nOutput ids: ");
    for (int64_t i = row - out.newTokens; i < row; i++) printf("%d ", out.tokenIds[(size_t)(b * row + i)]);
    printf("\n------------------------------------------------------------\n");
  }
  printf("Time cost: %lld ms, speed: %.2f token/s\n", (long long)ms, out.tokenIds.size() * 1000.0 / ms);
  printf("new tokens: %lld, new-token rate: %.2f token/s\n", (long long)(out.batch * out.newTokens), out.batch * out.newTokens * 1000.0 / ms);
  if (out.newTokens > 1) printf("time to first token: %.1f ms, decode-only rate: %.2f token/s\n", out.firstTokenMs, out.batch * (out.newTokens - 1) * 1000.0 / out.decodeMs);
  return 0;
}
