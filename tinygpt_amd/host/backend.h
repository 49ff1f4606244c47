// backend.h — the device shim as a table of C function pointers resolved with dlopen/dlsym.
// `--device mi355x` binds tinygpt_amd/lib/libtgx_mi355x.so (symbols tgx_*, include/tgx.h).  The table is the C++
// counterpart of the reference's virtual GPTModel + Sampler + AsyncTokenPipeline (src/model/GPTModel.h:80-106,
// src/engine/Sampler.h:30, src/engine/GPTEngine.cpp:17-35); the host engine sees nothing else of the device.
// Any library exporting the same entry points under another prefix can be bound (the tests bind the CPU oracle).
#pragma once
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/tgx.h"

namespace tgxh {

struct Backend {
  void* handle = nullptr;
  std::string error;

  int (*device_count)(int*) = nullptr;
  int (*create)(const tgx_model_desc*, int, tgx_ctx**) = nullptr;
  int (*upload)(tgx_ctx*, const char*, const void*, const int64_t*, int, int) = nullptr;
  int (*finalize)(tgx_ctx*) = nullptr;
  void (*destroy)(tgx_ctx*) = nullptr;
  int (*forward)(tgx_ctx*, const int64_t*, int, int) = nullptr;
  int (*read_logits)(tgx_ctx*, float*, int) = nullptr;
  int (*sample)(tgx_ctx*, const tgx_sampler_cfg*, uint64_t, int64_t*) = nullptr;
  int (*decode)(tgx_ctx*, const tgx_sampler_cfg*, uint64_t, int, int64_t*) = nullptr;
  int (*step_async)(tgx_ctx*, const tgx_sampler_cfg*, uint64_t, int64_t*) = nullptr;   // optional
  int (*fetch_token)(tgx_ctx*, int64_t, int32_t*) = nullptr;                            // optional
  int (*reset_cache)(tgx_ctx*) = nullptr;
  int64_t (*past_length)(const tgx_ctx*) = nullptr;
  int64_t (*context_size)(const tgx_ctx*) = nullptr;
  const char* (*last_error)(const tgx_ctx*) = nullptr;
  // per-row sequence lifecycle (include/tgx.h ABI 3; optional: a backend without them serves whole-batch generation only)
  int (*reset_row)(tgx_ctx*, int) = nullptr;
  int (*forward_row)(tgx_ctx*, int, const int64_t*, int) = nullptr;
  int (*sample_row)(tgx_ctx*, int, const tgx_sampler_cfg*, uint64_t, int64_t*) = nullptr;
  int64_t (*past_length_row)(const tgx_ctx*, int) = nullptr;

  bool open(const std::string& path, const std::string& prefix) {
    // RTLD_NODELETE: the shim's runtime owns threads (HIP's signal/event workers; libgomp's team under the CPU oracle) that
    // outlive the last context, so dlclose() must drop the handle without unmapping their code.
    handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NODELETE);
    if (!handle) { error = std::string("dlopen failed: ") + dlerror(); return false; }
    bool ok = true;
    auto sym = [&](const char* name, bool required) -> void* {
      void* p = dlsym(handle, (prefix + name).c_str());
      if (!p && required) { error += "missing symbol " + prefix + name + "; "; ok = false; }
      return p;
    };
#define TGXH_BIND(field, required) field = reinterpret_cast<decltype(field)>(sym(#field, required))
    TGXH_BIND(device_count, false);
    TGXH_BIND(create, true); TGXH_BIND(upload, true); TGXH_BIND(finalize, true); TGXH_BIND(destroy, true);
    TGXH_BIND(forward, true); TGXH_BIND(read_logits, true); TGXH_BIND(sample, true); TGXH_BIND(decode, true);
    TGXH_BIND(step_async, false); TGXH_BIND(fetch_token, false);
    TGXH_BIND(reset_cache, true); TGXH_BIND(past_length, true); TGXH_BIND(context_size, true); TGXH_BIND(last_error, true);
    TGXH_BIND(reset_row, false); TGXH_BIND(forward_row, false); TGXH_BIND(sample_row, false); TGXH_BIND(past_length_row, false);
#undef TGXH_BIND
    return ok;
  }
  void close() { if (handle) dlclose(handle); handle = nullptr; }
};

}  // namespace tgxh
