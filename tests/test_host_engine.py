"""The C++ host engine (tinygpt_amd/host: GPTEngine mirror, config + safetensors readers) bound to the CPU oracle
through the same function table that binds the HIP library — host logic is checked without a GPU."""
import os

import numpy as np
import pytest

from conftest import load_golden
from host_util import HostEngine, host_lib, write_model_dir
from tinygpt_amd import synth


@pytest.fixture(scope="module")
def lib():
    return host_lib(test_hooks=True)      # binds the CPU oracle: only the -DTGXH_TEST_HOOKS build can


@pytest.fixture(scope="module")
def oracle_path(oracle_lib):
    return oracle_lib.path


def make_engine(lib, oracle_path, tmp_path, fam, dtype, shards=1, max_batch=4, eos=None, file_dtype="bf16"):
    cfg, g = load_golden(fam)
    write_model_dir(str(tmp_path), cfg, int(g["seed"]), float(g["std"]), shards=shards, eos=eos, dtype=file_dtype)
    e = HostEngine(lib, model_dir=str(tmp_path), backend_lib=oracle_path, prefix="tgxo_", dtype=dtype, max_batch=max_batch)
    assert e.prepare(), e.error()
    return e, g


def test_synth_cpp_equals_python(lib):
    import ctypes
    for name, n, std in [("model.layers.3.mlp.up_proj.weight", 70001, 0.02), ("model.norm.weight", 777, 0.02),
                         ("model.layers.0.input_layernorm.weight", 64, 0.08), ("lm_head.weight", 1 << 21, 0.08)]:
        out = np.zeros(n, np.uint16)
        lib.tgxe_synth_tensor(1234, name.encode(), n, std, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)))
        np.testing.assert_array_equal(out, synth.synth_tensor_bf16(1234, name, (n,), std, force_numpy=True))


@pytest.mark.parametrize("fam,shards", [("llama_tiny", 1), ("qwen2_tiny", 2), ("mistral_tiny", 1), ("qwen3_tiny", 1)])
def test_generate_sync_matches_hf_golden(lib, oracle_path, tmp_path, fam, shards):
    """config.json + (sharded) safetensors -> engine -> greedy ids == HF's, prompt ids echoed, finishReason Length."""
    e, g = make_engine(lib, oracle_path, tmp_path, fam, dtype=0, shards=shards)
    n_new = g["ids_fp32"].shape[1]
    e.reconfigure(max_new=n_new)
    ids, new, fin = e.generate_sync([g["prompt"][0]])
    assert new == n_new and fin == "length"                          # generateSync never stops on EOS (GPTEngine.cpp:170-172)
    np.testing.assert_array_equal(ids[0, :g["prompt"].shape[1]], g["prompt"][0])
    np.testing.assert_array_equal(ids[0, g["prompt"].shape[1]:], g["ids_fp32"][0])
    e.close()


def test_left_padding_without_mask_gpt2_batch4(lib, oracle_path, tmp_path):
    """E4: prompts of lengths 5/7/5/5 are left-padded with id 0 to 7 and run WITHOUT a mask (GPTEngine.cpp:95,130-138).
    The golden ids come from HF run on the padded batch with no attention mask."""
    e, g = make_engine(lib, oracle_path, tmp_path, "gpt2_tiny", dtype=0, file_dtype="fp32")
    prompts = [row[np.argmax(row != 0):] for row in g["prompt"]]      # strip the pads the generator added
    assert [len(p) for p in prompts] == [5, 7, 5, 5]
    e.reconfigure(max_new=g["ids_fp32"].shape[1])
    ids, new, fin = e.generate_sync(prompts, pad=0)
    np.testing.assert_array_equal(ids[:, :7], g["prompt"])
    np.testing.assert_array_equal(ids[:, 7:], g["ids_fp32"])
    e.close()


def test_truncation_keeps_tail(lib, oracle_path, tmp_path):
    """Prompts longer than contextSize keep their LAST contextSize tokens (GPTEngine.cpp:127-129); llama3-scaled
    models use original_max_position_embeddings as contextSize (ModelLlama.h:26-31) = 64 for the fixture."""
    e, g = make_engine(lib, oracle_path, tmp_path, "llama_tiny", dtype=0)
    assert e.lib.tgxe_context_size(e.h) == 64
    long = (np.arange(100) * 7 + 3) % 256
    e.reconfigure(max_new=1)
    ids, new, _ = e.generate_sync([long])
    assert ids.shape[1] == 64 + 1
    np.testing.assert_array_equal(ids[0, :64], long[-64:])
    e.close()


def test_empty_and_overlong_requests_fail_loudly(lib, oracle_path, tmp_path):
    """Edge cases of the engine boundary: a batch of empty prompts has nothing to prefill; a prompt that fills the context
    leaves no room for a second token (the reference would walk off its RoPE table, Attention.h:81-83) — both are errors
    with a message, never a crash or a silent truncation of the output."""
    e, g = make_engine(lib, oracle_path, tmp_path, "llama_tiny", dtype=0)
    e.reconfigure(max_new=4)
    with pytest.raises(Exception, match="out of range|empty"):
        e.generate_sync([np.zeros(0, np.int64)])
    full = (np.arange(64) * 5 + 1) % 256                       # contextSize of the fixture (llama3-scaled) is 64
    with pytest.raises(Exception, match="context size"):
        e.generate_sync([full])
    e.reconfigure(max_new=1)                                    # exactly one new token still fits: logits of position 63
    ids, new, _ = e.generate_sync([full])
    assert new == 1 and ids.shape == (1, 65)
    e.close()


def test_generate_async_stream_eos_abort(lib, oracle_path, tmp_path):
    e, g = make_engine(lib, oracle_path, tmp_path, "llama_tiny", dtype=0, eos=[2, 999])
    gold = g["ids_fp32"][0]
    prompt = g["prompt"][0]
    assert e.eos_ids() == [2, 999]                                   # generation_config eos array (ModelConfig.cpp:142-156)
    n = len(gold)
    # by length: callback sees T1..T(n-1); the token list also holds the never-reported last one (appendix A.9)
    e.reconfigure(max_new=n)
    ids, new, fin, seen = e.generate_async(prompt)
    assert fin == "length" and seen == list(gold[:n - 1]) and new == n
    np.testing.assert_array_equal(ids[len(prompt):], gold)
    # extra stop id -> Stop at that token, which is in the list but not reported
    stop = int(gold[5])
    k = list(gold).index(stop)
    e.reconfigure(max_new=n, extra_stop=[stop])
    ids, new, fin, seen = e.generate_async(prompt)
    assert fin == "stop" and seen == list(gold[:k])
    np.testing.assert_array_equal(ids[len(prompt):], gold[:k + 1])
    # abort from the callback after 3 tokens
    e.reconfigure(max_new=n)
    count = []
    ids, new, fin, seen = e.generate_async(prompt, on_token=lambda t: (count.append(t), len(count) < 3)[1])
    assert fin == "stop" and seen == list(gold[:3])                  # the aborting token was delivered, nothing after it
    np.testing.assert_array_equal(ids[len(prompt):], gold[:3])
    e.close()


def test_sampler_defaults_and_reconfigure_resets_cache(lib, oracle_path, tmp_path):
    """reconfigure() resets the KV cache (GPTEngine.cpp:83): two identical requests give identical ids."""
    e, g = make_engine(lib, oracle_path, tmp_path, "qwen2_tiny", dtype=1)
    outs = []
    for _ in range(2):
        e.reconfigure(temperature=0.8, top_p=0.9, max_new=8)          # CLI defaults (main.cpp:36-37)
        outs.append(e.generate_sync([g["prompt"][0]])[0])
    np.testing.assert_array_equal(outs[0], outs[1])
    e.close()


def test_bad_inputs_fail_loudly(lib, oracle_path, tmp_path):
    import json, os
    cfg, g = load_golden("llama_tiny")
    write_model_dir(str(tmp_path), cfg, int(g["seed"]), float(g["std"]))
    os.remove(os.path.join(str(tmp_path), "generation_config.json"))     # required file (ModelLoader.cpp:34-38)
    e = HostEngine(lib, model_dir=str(tmp_path), backend_lib=oracle_path, prefix="tgxo_", dtype=0)
    assert not e.prepare() and "generation_config" in e.error()
    e.close()
    e = HostEngine(lib, model_dir=str(tmp_path), device="cpu")          # no CPU execution path in the product engine
    assert not e.prepare() and "mi355x" in e.error()
    e.close()
    bad = dict(cfg, model_type="falcon")
    d2 = tmp_path / "bad"
    write_model_dir(str(d2), cfg, 1, 0.05)
    json.dump(bad, open(d2 / "config.json", "w"))
    e = HostEngine(lib, model_dir=str(d2), backend_lib=oracle_path, prefix="tgxo_", dtype=0)
    assert not e.prepare() and "Unsupported model_type" in e.error()
    e.close()


# ---- text entry points: tokenizer -> engine -> tokenizer (GPTEngine.cpp:101-174,180-232) -----------------------------------
TOK_DIR = __import__("os").path.join(__import__("conftest").GOLDEN, "tokenizer", "llama3_style")


def text_engine(lib, oracle_path, tmp_path, max_batch=4):
    """llama_tiny widened to the 1200-token vocabulary of the llama3_style fixture tokenizer."""
    cfg, g = load_golden("llama_tiny")
    cfg = dict(cfg, vocab_size=1280)
    write_model_dir(str(tmp_path), cfg, 77, 0.08)
    e = HostEngine(lib, model_dir=str(tmp_path), backend_lib=oracle_path, prefix="tgxo_", dtype=1, max_batch=max_batch, tokenizer_dir=TOK_DIR)
    assert e.prepare(), e.error()
    return e


def test_generate_sync_from_text_equals_id_path(lib, oracle_path, tmp_path):
    """encodeTexts == tokenizer.encodeBatch + left pad with pad -> eos -> 0 (:101-144); output texts == decodeBatch of the
    new tokens only (decodeTokens, :146-152)."""
    from host_util import HostTokenizer
    e = text_engine(lib, oracle_path, tmp_path)
    tok = HostTokenizer(lib, TOK_DIR)
    texts = ["Hello, my name is", "The president of the United States is", "The capital of France is", "The future of AI is"]
    e.reconfigure(max_new=6)
    ids_t, new_t, out_texts = e.generate_sync_text(texts)
    prompts = [tok.encode(t) for t in texts]
    assert all(p[0] == tok.bos for p in prompts)
    pad = tok.pad if tok.pad >= 0 else (tok.eos if tok.eos >= 0 else 0)
    e.reconfigure(max_new=6)                                   # resets the KV cache (:83)
    ids_i, new_i, _ = e.generate_sync(prompts, pad=pad)
    np.testing.assert_array_equal(ids_t, ids_i)
    assert new_t == new_i == 6
    S = ids_t.shape[1] - 6
    assert S == max(len(p) for p in prompts)
    assert (ids_t[2, : S - len(prompts[2])] == pad).all()      # left padding
    for b in range(4):
        assert out_texts[b].decode("utf-8", errors="replace") == tok.decode(list(ids_t[b, S:]))
    e.close(); tok.close()


def test_generate_async_text_streams_utf8_safe_chunks(lib, oracle_path, tmp_path):
    from host_util import HostTokenizer
    e = text_engine(lib, oracle_path, tmp_path, max_batch=1)
    tok = HostTokenizer(lib, TOK_DIR)
    e.reconfigure(max_new=24)
    ids, new, fin, chunks = e.generate_async_text("你好, hello")
    prompt = tok.encode("你好, hello")
    np.testing.assert_array_equal(ids[:len(prompt)], prompt)
    def truncated_tail(b):                                      # a lead byte at the end that announces more bytes than follow
        for back in range(1, min(4, len(b)) + 1):
            c = b[-back]
            if c & 0xC0 == 0x80:
                continue
            need = 4 if c >= 0xF0 else 3 if c >= 0xE0 else 2 if c >= 0xC0 else 1
            return need > back
        return False
    assert chunks and not any(truncated_tail(c) for c in chunks[:-1])   # never a split character (a random model may emit invalid bytes)
    # the callback saw tokens T1..T(n-1) (the last future token is not reported when the loop ends by length, :196-217)
    streamed = b"".join(chunks)
    reported = list(ids[len(prompt):len(prompt) + new - (1 if fin == "length" else 0)])
    assert streamed.decode("utf-8", errors="replace") == tok.decode(reported)
    assert len(streamed) > 0
    # abort from the callback: finish reason Stop, nothing flushed afterwards
    e.reconfigure(max_new=24)
    ids2, new2, fin2, chunks2 = e.generate_async_text("你好, hello", on_chunk=lambda c: False)
    assert fin2 == "stop" and len(chunks2) == 1
    e.close(); tok.close()


def test_text_entry_points_need_a_tokenizer(lib, oracle_path, tmp_path):
    e, g = make_engine(lib, oracle_path, tmp_path, "llama_tiny", 1)
    arr_err = None
    try:
        e.generate_sync_text(["hi"])
    except AssertionError as ex:
        arr_err = str(ex)
    assert arr_err and "no tokenizer" in arr_err
    e.close()


def test_corrupt_model_directories_fail_loudly(lib, oracle_path, tmp_path):
    """Truncated / malformed config.json, generation_config.json and safetensors files (header length, offsets, shapes, short data):
    prepare() returns false with a message for every one of them — no crash, no partially loaded model (ModelLoader.cpp:25-89,
    SafeTensors.cpp:141-229 fail the same way: bool + log)."""
    import json, os, shutil, struct
    cfg, g = load_golden("llama_tiny")
    base = str(tmp_path / "base")
    write_model_dir(base, cfg, int(g["seed"]), float(g["std"]))

    def attempt(mutate):
        d = str(tmp_path / "case")
        shutil.rmtree(d, ignore_errors=True); shutil.copytree(base, d)
        mutate(d)
        e = HostEngine(lib, model_dir=d, backend_lib=oracle_path, prefix="tgxo_", dtype=0)
        ok, err = e.prepare(), e.error()
        e.close()
        return ok, err

    def write(path, data, mode="w"):
        with open(path, mode) as f:
            f.write(data)

    def edit_header(d, fn):
        p = os.path.join(d, "model.safetensors"); b = open(p, "rb").read(); n = struct.unpack("<Q", b[:8])[0]
        h = json.loads(b[8:8 + n]); fn(h, [k for k in h if k != "__metadata__"][0])
        hb = json.dumps(h).encode(); write(p, struct.pack("<Q", len(hb)) + hb + b[8 + n:], "wb")

    def set_header_len(d, n):
        p = os.path.join(d, "model.safetensors"); b = bytearray(open(p, "rb").read()); b[:8] = struct.pack("<Q", n); write(p, bytes(b), "wb")

    def truncate(d, keep):
        p = os.path.join(d, "model.safetensors"); b = open(p, "rb").read(); write(p, b[:keep if keep > 0 else len(b) + keep], "wb")

    assert attempt(lambda d: None) == (True, "")
    cases = {
        "config truncated": lambda d: write(os.path.join(d, "config.json"), json.dumps(cfg)[:50]),
        "config empty": lambda d: write(os.path.join(d, "config.json"), ""),
        "config is a list": lambda d: write(os.path.join(d, "config.json"), "[1,2,3]"),
        "config wrong types": lambda d: write(os.path.join(d, "config.json"), json.dumps(dict(cfg, hidden_size="big", num_hidden_layers=None))),
        "config negative heads": lambda d: write(os.path.join(d, "config.json"), json.dumps(dict(cfg, num_attention_heads=-4))),
        "config unknown model_type": lambda d: write(os.path.join(d, "config.json"), json.dumps(dict(cfg, model_type="bert"))),
        "generation_config missing": lambda d: os.remove(os.path.join(d, "generation_config.json")),
        "safetensors 4 bytes": lambda d: truncate(d, 4),
        "safetensors header cut": lambda d: truncate(d, 100),
        "safetensors data cut": lambda d: truncate(d, -1000),
        "safetensors header length 2^60": lambda d: set_header_len(d, 2 ** 60),
        "safetensors header length 0": lambda d: set_header_len(d, 0),
        "safetensors header length 2^64-4 (8 + n wraps)": lambda d: set_header_len(d, 2 ** 64 - 4),
        "safetensors header length 2^64-8": lambda d: set_header_len(d, 2 ** 64 - 8),
        "safetensors shape product overflows": lambda d: edit_header(d, lambda h, k: h[k].__setitem__("shape", [2 ** 40, 2 ** 40])),
        "safetensors fractional shape": lambda d: edit_header(d, lambda h, k: h[k].__setitem__("shape", [1.5, 2])),
        "safetensors needed tensor in an unsupported dtype": lambda d: edit_header(d, lambda h, k: h[k].__setitem__("dtype", "I64")),
        "config nested 100000 deep": lambda d: write(os.path.join(d, "config.json"), "[" * 100000 + "]" * 100000),
        "safetensors offsets beyond file": lambda d: edit_header(d, lambda h, k: h[k].__setitem__("data_offsets", [0, 2 ** 50])),
        "safetensors negative shape": lambda d: edit_header(d, lambda h, k: h[k].__setitem__("shape", [-1, 7])),
        "model dir missing": lambda d: shutil.rmtree(d),
    }
    for label, mutate in cases.items():
        ok, err = attempt(mutate)
        assert not ok and err, label


def test_extra_non_float_buffers_are_unexpected_keys_not_errors(lib, oracle_path, tmp_path):
    """A checkpoint that also carries I64 / BOOL / U8 buffers the model has no parameter for (position_ids, mask buffers of old hub
    checkpoints) loads: the reference looks the name up first and only warns "Unexpected key" (SafeTensors.cpp:176-183); the dtype of a
    tensor matters only when the model consumes it."""
    import torch
    from safetensors.torch import load_file, save_file
    cfg, g = load_golden("llama_tiny")
    write_model_dir(str(tmp_path), cfg, int(g["seed"]), float(g["std"]))
    p = os.path.join(str(tmp_path), "model.safetensors")
    t = load_file(p)
    t["model.layers.0.self_attn.rotary_emb.position_ids"] = torch.arange(16, dtype=torch.int64)
    t["model.causal_mask"] = torch.ones(4, 4, dtype=torch.bool)
    t["model.byte_buffer"] = torch.zeros(3, dtype=torch.uint8)
    t["model.scalar"] = torch.tensor(3, dtype=torch.int32)
    save_file(t, p)
    e = HostEngine(lib, model_dir=str(tmp_path), backend_lib=oracle_path, prefix="tgxo_", dtype=0)
    assert e.prepare(), e.error()
    e.reconfigure(max_new=4)
    ids, new, fin = e.generate_sync([np.asarray(g["prompt"][0], np.int32)])
    assert new == 4
    e.close()
