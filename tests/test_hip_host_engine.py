"""The C++ host engine bound to the HIP library (`--device mi355x`): HF-layout directory in, greedy ids out."""
import os
import subprocess

import numpy as np
import pytest

from conftest import load_golden
from host_util import HostEngine, host_lib, write_model_dir
from tinygpt_amd import build

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fam,shards", [("llama_tiny", 2), ("mistral_tiny", 1)])
def test_engine_on_mi355x_matches_hf_ids(tmp_path, fam, shards):
    cfg, g = load_golden(fam)
    write_model_dir(str(tmp_path), cfg, int(g["seed"]), float(g["std"]), shards=shards)
    e = HostEngine(host_lib(), model_dir=str(tmp_path), device="mi355x", dtype=1)      # default shim: libtgx_mi355x.so next to it
    assert e.prepare(), e.error()
    n = g["ids_bf16"].shape[1]
    e.reconfigure(max_new=n)
    ids, new, fin = e.generate_sync([g["prompt"][0]])
    np.testing.assert_array_equal(ids[0, g["prompt"].shape[1]:], g["ids_bf16"][0])
    # streaming path with the async ticket pipeline
    e.reconfigure(max_new=n)
    ids2, new2, fin2, seen = e.generate_async(g["prompt"][0])
    assert fin2 == "length" and seen == list(g["ids_bf16"][0][:n - 1])
    np.testing.assert_array_equal(ids2[g["prompt"].shape[1]:], g["ids_bf16"][0])
    e.close()


@pytest.mark.parametrize("dtype,key", [(0, "ids_fp32"), (1, "ids_bf16")])
def test_gpt2_left_padded_batch_of_four_on_mi355x(tmp_path, dtype, key):
    """BASELINE.json configs[0] semantics on the GPU: GPT-2 checkpoint in the hub layout (Conv1D [in][out] weights, fp32 file),
    the CLI's 4 prompts of lengths 5/7/5/5 left-padded with id 0 and run WITHOUT a mask (GPTEngine.cpp:95,130-138);
    ids == HF's on the padded batch, for --dtype fp32 (the reference's CPU case) and bf16 (its CLI default)."""
    cfg, g = load_golden("gpt2_hd64")
    write_model_dir(str(tmp_path), cfg, int(g["seed"]), float(g["std"]), dtype="fp32")
    e = HostEngine(host_lib(), model_dir=str(tmp_path), device="mi355x", dtype=dtype, max_batch=4)
    assert e.prepare(), e.error()
    prompts = [row[np.argmax(row != 0):] for row in g["prompt"]]
    assert [len(p) for p in prompts] == [5, 7, 5, 5]
    e.reconfigure(max_new=g[key].shape[1])
    ids, new, fin = e.generate_sync(prompts, pad=0)
    np.testing.assert_array_equal(ids[:, :7], g["prompt"])
    np.testing.assert_array_equal(ids[:, 7:], g[key])
    e.close()


def test_cli_runs_batch_of_four(tmp_path):
    """tgx_cli with the reference's flags on a 4-row batch (left-padded, no mask), greedy: rows equal the engine's."""
    cfg, g = load_golden("qwen2_tiny")
    write_model_dir(str(tmp_path), cfg, int(g["seed"]), float(g["std"]))
    _, cli = build.build_host()
    out = subprocess.run([cli, "--model", str(tmp_path), "--device", "mi355x", "--dtype", "bf16", "--max-tokens", "8",
                          "--temperature", "0", "--top-p", "1", "--prompt-ids", "5,6,7,8,9;1,2,3;4,4,4,4,4,4,4;9"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.count("Output ids:") == 4 and "speed:" in out.stdout and "token/s" in out.stdout


def test_cli_baseline_config0_gpt2_on_gpu():
    """BASELINE.json configs[0] through the harness: GPT-2 124M geometry, fp32, the CLI's four prompts as gpt2 ids left-padded to 7,
    greedy, 32 new tokens (main.cpp:12-17,33-37) — on the GPU instead of the reference's CPU device."""
    _, cli = build.build_host()
    out = subprocess.run([cli, "--synthetic", "gpt2", "--device", "mi355x", "--dtype", "fp32", "--max-tokens", "32", "--temperature", "0", "--top-p", "1"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.count("Output ids:") == 4 and "Prompt ids: 0 0 464 3139 286 4881 318" in out.stdout and "token/s" in out.stdout
    rows = [l.split(":")[1].split() for l in out.stdout.splitlines() if l.startswith("Output ids:")]
    assert all(len(r) == 32 for r in rows)


def test_cli_text_prompts_on_gpu(tmp_path, oracle_lib):
    """The reference harness end to end (main.cpp:12-17,97-114): the four text prompts -> tokenizer -> HIP decode path ->
    detokenised outputs, batch 4 sharing each weight pass; the ids equal the CPU oracle's through the same host engine."""
    import os
    import numpy as np
    from conftest import GOLDEN
    from host_util import HostEngine, HostTokenizer, host_lib
    tok_dir = os.path.join(GOLDEN, "tokenizer", "llama3_style")
    cfg, g = load_golden("llama_tiny")
    cfg = dict(cfg, vocab_size=1280)
    write_model_dir(str(tmp_path), cfg, 77, 0.08)
    _, cli = build.build_host()
    out = subprocess.run([cli, "--model", str(tmp_path), "--tokenizer", tok_dir, "--max-tokens", "8", "--temperature", "0", "--top-p", "1"],
                         capture_output=True, timeout=300)          # bytes: a random model may emit invalid UTF-8
    assert out.returncode == 0, out.stderr
    assert out.stdout.count(b"Prompt:") == 4 and b"'Hello, my name is'" in out.stdout and b"token/s" in out.stdout
    lib = host_lib()
    texts = ["Hello, my name is", "The president of the United States is", "The capital of France is", "The future of AI is"]
    gpu = HostEngine(lib, model_dir=str(tmp_path), tokenizer_dir=tok_dir, max_batch=4)
    ref = HostEngine(host_lib(test_hooks=True), model_dir=str(tmp_path), tokenizer_dir=tok_dir, max_batch=4, backend_lib=oracle_lib.path, prefix="tgxo_")
    assert gpu.prepare(), gpu.error()
    assert ref.prepare(), ref.error()
    gpu.reconfigure(max_new=8); ref.reconfigure(max_new=8)
    ids_g, _, txt_g = gpu.generate_sync_text(texts)
    ids_r, _, txt_r = ref.generate_sync_text(texts)
    np.testing.assert_array_equal(ids_g, ids_r)
    assert txt_g == txt_r
    for t in txt_g:
        assert b"'" + t + b"'" in out.stdout                    # the CLI printed the same continuations
    gpu.close(); ref.close()
