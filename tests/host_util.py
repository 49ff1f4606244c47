"""ctypes view of tinygpt_amd/lib/libtgx_host.so (the C++ host engine) + helpers to write HF-style model dirs."""
import ctypes
import json
import os
from ctypes import CFUNCTYPE, POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint16, c_uint64, c_void_p

import numpy as np

from tinygpt_amd import build, synth
from tinygpt_amd.desc import desc_from_hf_config

TOKEN_CB = CFUNCTYPE(c_int, c_int32, c_void_p)


def host_lib(test_hooks=False):
    """The shipped libtgx_host.so (binds libtgx_mi355x.so only) or, for CPU host-logic tests that bind the oracle, the
    -DTGXH_TEST_HOOKS build under tests/_build."""
    lib_path, _ = build.build_host(test_hooks=test_hooks)
    lib = ctypes.CDLL(lib_path)
    lib.tgxe_create.restype = c_void_p
    lib.tgxe_create.argtypes = [c_char_p, c_char_p, c_char_p, c_char_p, c_char_p, c_int, c_int, c_int]
    lib.tgxe_destroy.argtypes = [c_void_p]
    lib.tgxe_prepare.argtypes = [c_void_p]
    lib.tgxe_last_error.restype = c_char_p
    lib.tgxe_last_error.argtypes = [c_void_p]
    lib.tgxe_context_size.restype = c_int64
    lib.tgxe_context_size.argtypes = [c_void_p]
    lib.tgxe_eos_ids.argtypes = [c_void_p, POINTER(c_int32), c_int]
    lib.tgxe_reconfigure.argtypes = [c_void_p, c_float, c_int64, c_float, c_float, c_int64, POINTER(c_int32), c_int]
    lib.tgxe_generate_sync.argtypes = [c_void_p, POINTER(c_int32), POINTER(c_int32), c_int, c_int32, POINTER(c_int32), c_int64,
                                       POINTER(c_int64), POINTER(c_int64), POINTER(c_int)]
    lib.tgxe_generate_async.argtypes = [c_void_p, POINTER(c_int32), c_int, TOKEN_CB, c_void_p, POINTER(c_int32), c_int64,
                                        POINTER(c_int64), POINTER(c_int64), POINTER(c_int)]
    lib.tgxe_synth_tensor.argtypes = [c_uint64, c_char_p, c_int64, c_double, POINTER(c_uint16)]
    return lib


class HostEngine:
    def __init__(self, lib, model_dir=None, synthetic=None, device="mi355x", backend_lib=None, prefix="tgx_", dtype=1, max_batch=4, tokenizer_dir=None):
        self.lib = lib
        lib.tgxe_create2.restype = c_void_p
        lib.tgxe_create2.argtypes = [c_char_p, c_char_p, c_char_p, c_char_p, c_char_p, c_int, c_int, c_int, c_char_p]
        self.h = lib.tgxe_create2((model_dir or "").encode(), (synthetic or "").encode(), device.encode(),
                                  (backend_lib or "").encode(), prefix.encode(), 0, dtype, max_batch,
                                  tokenizer_dir.encode() if tokenizer_dir else None)

    def prepare(self):
        return self.lib.tgxe_prepare(self.h) == 0

    def error(self):
        return self.lib.tgxe_last_error(self.h).decode()

    def eos_ids(self):
        buf = (c_int32 * 16)()
        n = self.lib.tgxe_eos_ids(self.h, buf, 16)
        return list(buf[:n])

    def reconfigure(self, temperature=0.0, top_k=0, top_p=1.0, min_p=0.0, max_new=16, extra_stop=()):
        ex = (c_int32 * max(1, len(extra_stop)))(*extra_stop)
        self.lib.tgxe_reconfigure(self.h, temperature, top_k, top_p, min_p, max_new, ex, len(extra_stop))

    def generate_sync(self, prompts, pad=0, cap=1 << 16):
        flat = np.concatenate([np.asarray(p, np.int32) for p in prompts])
        lens = np.asarray([len(p) for p in prompts], np.int32)
        out = np.zeros(cap, np.int32)
        n, new, fin = c_int64(), c_int64(), c_int()
        rc = self.lib.tgxe_generate_sync(self.h, flat.ctypes.data_as(POINTER(c_int32)), lens.ctypes.data_as(POINTER(c_int32)), len(prompts), pad,
                                         out.ctypes.data_as(POINTER(c_int32)), cap, ctypes.byref(n), ctypes.byref(new), ctypes.byref(fin))
        assert rc == 0, self.error()
        return out[:n.value].reshape(len(prompts), -1), new.value, ("stop", "length")[fin.value]

    def generate_async(self, prompt, on_token=None, cap=1 << 16):
        seen = []

        def cb(tok, _):
            seen.append(int(tok))
            return 1 if (on_token is None or on_token(int(tok))) else 0

        p = np.asarray(prompt, np.int32)
        out = np.zeros(cap, np.int32)
        n, new, fin = c_int64(), c_int64(), c_int()
        rc = self.lib.tgxe_generate_async(self.h, p.ctypes.data_as(POINTER(c_int32)), len(p), TOKEN_CB(cb), None,
                                          out.ctypes.data_as(POINTER(c_int32)), cap, ctypes.byref(n), ctypes.byref(new), ctypes.byref(fin))
        assert rc == 0, self.error()
        return out[:n.value], new.value, ("stop", "length")[fin.value], seen

    def generate_sync_text(self, texts, cap=1 << 16):
        arr = (c_char_p * len(texts))(*[t.encode("utf-8") for t in texts])
        out = np.zeros(cap, np.int32)
        buf = ctypes.create_string_buffer(1 << 16)
        n, new, tl = c_int64(), c_int64(), c_int64()
        self.lib.tgxe_generate_sync_text.argtypes = [c_void_p, POINTER(c_char_p), c_int, POINTER(c_int32), c_int64, POINTER(c_int64),
                                                     POINTER(c_int64), c_char_p, c_int64, POINTER(c_int64)]
        rc = self.lib.tgxe_generate_sync_text(self.h, arr, len(texts), out.ctypes.data_as(POINTER(c_int32)), cap, ctypes.byref(n),
                                              ctypes.byref(new), buf, 1 << 16, ctypes.byref(tl))
        assert rc == 0, self.error()
        return out[:n.value].reshape(len(texts), -1), new.value, buf.raw[:tl.value].split(b"\x1e")

    def generate_async_text(self, text, on_chunk=None, cap=1 << 16):
        chunks = []
        TEXT_CB = CFUNCTYPE(c_int, POINTER(ctypes.c_char), c_int64, c_void_p)

        def cb(ptr, length, _):
            c = ctypes.string_at(ptr, length)
            chunks.append(c)
            return 1 if (on_chunk is None or on_chunk(c)) else 0

        out = np.zeros(cap, np.int32)
        n, new, fin = c_int64(), c_int64(), c_int()
        self.lib.tgxe_generate_async_text.argtypes = [c_void_p, c_char_p, TEXT_CB, c_void_p, POINTER(c_int32), c_int64, POINTER(c_int64),
                                                      POINTER(c_int64), POINTER(c_int)]
        rc = self.lib.tgxe_generate_async_text(self.h, text.encode("utf-8"), TEXT_CB(cb), None, out.ctypes.data_as(POINTER(c_int32)), cap,
                                               ctypes.byref(n), ctypes.byref(new), ctypes.byref(fin))
        assert rc == 0, self.error()
        return out[:n.value], new.value, ("stop", "length")[fin.value], chunks

    def close(self):
        if self.h:
            self.lib.tgxe_destroy(self.h)
            self.h = None


def write_model_dir(path, cfg, seed, std, shards=1, dtype="bf16", eos=None, peaked=False):
    """HF-layout directory with deterministic weights: config.json, generation_config.json, model.safetensors
    (or 2 shards + model.safetensors.index.json).  Written with the `safetensors` package (an independent writer)."""
    import torch
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    d = desc_from_hf_config(cfg, "bf16")
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    with open(os.path.join(path, "generation_config.json"), "w") as f:
        json.dump({"bos_token_id": 1, "eos_token_id": eos if eos is not None else cfg.get("eos_token_id", 2)}, f)
    tensors = {}
    for name, bits in synth.synth_checkpoint(d, seed, std, peaked):
        if dtype == "bf16":      # reinterpret the bit patterns: no float round trip (matters at full model size)
            tensors[name] = torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16)
        elif dtype == "fp16":
            tensors[name] = torch.from_numpy(synth.bf16_bits_to_f32(bits).copy()).to(torch.float16)
        else:
            tensors[name] = torch.from_numpy(synth.bf16_bits_to_f32(bits).copy())
    if shards == 1:
        save_file(tensors, os.path.join(path, "model.safetensors"))
    else:
        names = list(tensors)
        parts = [names[i::shards] for i in range(shards)]
        wm = {}
        for i, part in enumerate(parts):
            fn = f"model-{i + 1:05d}-of-{shards:05d}.safetensors"
            save_file({n: tensors[n] for n in part}, os.path.join(path, fn))
            wm.update({n: fn for n in part})
        with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {}, "weight_map": wm}, f)
    return d


class HostTokenizer:
    """ctypes view of tgxh::Tokenizer (tinygpt_amd/host/tokenizer.h) through the tgxe_tok_* entry points."""

    def __init__(self, lib, directory):
        self.lib = lib
        lib.tgxe_tok_create.restype = c_void_p
        lib.tgxe_tok_create.argtypes = [c_char_p, c_char_p, c_char_p, c_int]
        lib.tgxe_tok_destroy.argtypes = [c_void_p]
        lib.tgxe_tok_encode.restype = c_int64
        lib.tgxe_tok_encode.argtypes = [c_void_p, c_char_p, c_int64, c_int, POINTER(c_int32), c_int64]
        lib.tgxe_tok_decode.restype = c_int64
        lib.tgxe_tok_decode.argtypes = [c_void_p, POINTER(c_int32), c_int64, c_int, c_char_p, c_int64]
        lib.tgxe_tok_special.restype = c_int32
        lib.tgxe_tok_special.argtypes = [c_void_p, c_int]
        lib.tgxe_tok_token_to_id.restype = c_int32
        lib.tgxe_tok_token_to_id.argtypes = [c_void_p, c_char_p]
        err = ctypes.create_string_buffer(512)
        self.h = lib.tgxe_tok_create(os.path.join(directory, "tokenizer.json").encode(), os.path.join(directory, "tokenizer_config.json").encode(), err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def close(self):
        if self.h:
            self.lib.tgxe_tok_destroy(self.h)
            self.h = None

    def encode(self, text, allow_added=True):
        b = text.encode("utf-8")
        cap = 2 * len(b) + 16
        buf = (c_int32 * cap)()
        n = self.lib.tgxe_tok_encode(self.h, b, len(b), int(allow_added), buf, cap)
        return list(buf[:n])

    def _decode(self, ids, mode):
        arr = (c_int32 * max(1, len(ids)))(*ids)
        n = self.lib.tgxe_tok_decode(self.h, arr, len(ids), mode, None, 0)
        out = ctypes.create_string_buffer(max(1, n))
        if mode == 0:
            self.lib.tgxe_tok_decode(self.h, arr, len(ids), 0, out, n)
            return out.raw[:n]
        # stream calls are stateful: the first call consumed the ids; the bytes sit in the handle's scratch
        return self._scratch(n)

    def _scratch(self, n):
        self.lib.tgxe_tok_scratch.restype = c_int64
        self.lib.tgxe_tok_scratch.argtypes = [c_void_p, c_char_p, c_int64]
        out = ctypes.create_string_buffer(max(1, n))
        self.lib.tgxe_tok_scratch(self.h, out, n)
        return out.raw[:n]

    def decode(self, ids):
        return self._decode(list(ids), 0).decode("utf-8", errors="replace")

    def decode_stream(self, ids):
        return self._decode(list(ids), 1)

    def decode_stream_flush(self):
        return self._decode([], 2)

    @property
    def bos(self): return self.lib.tgxe_tok_special(self.h, 0)
    @property
    def eos(self): return self.lib.tgxe_tok_special(self.h, 1)
    @property
    def pad(self): return self.lib.tgxe_tok_special(self.h, 2)

    def token_to_id(self, token):
        return self.lib.tgxe_tok_token_to_id(self.h, token.encode("utf-8"))


def regex_match_all(lib, pattern, text):
    lib.tgxe_regex_match_all.restype = c_int64
    lib.tgxe_regex_match_all.argtypes = [c_char_p, c_char_p, c_int64, POINTER(c_int64), c_int64]
    b = text.encode("utf-8")
    cap = len(b) + 1
    buf = (c_int64 * (2 * cap))()
    n = lib.tgxe_regex_match_all(pattern.encode("utf-8"), b, len(b), buf, cap)
    if n < 0:
        return None
    return [[buf[2 * i], buf[2 * i + 1]] for i in range(n)]
