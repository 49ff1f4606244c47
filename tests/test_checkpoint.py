"""tinygpt_amd/checkpoint.py (the Python side's safetensors reader: bench.py --model-dir) against the directories the C++ loader's tests write:
single-file and sharded checkpoints, BF16 / F16 / F32, every tensor name and bit pattern equal to the synthetic checkpoint it was written from
(ModelLoader.cpp:18-89, SafeTensors.cpp:141-229); through the oracle the directory's logits equal the synthetic model's."""
import numpy as np
import pytest

from conftest import load_golden
from host_util import write_model_dir
from tinygpt_amd import synth
from tinygpt_amd.checkpoint import iter_checkpoint
from tinygpt_amd.desc import desc_from_hf_config, load_desc


@pytest.mark.parametrize("fam,shards,dtype", [("llama_tiny", 1, "bf16"), ("qwen2_tiny", 3, "bf16"), ("mistral_tiny", 2, "fp32"), ("gpt2_hd64", 1, "fp16")])
def test_reader_returns_what_was_written(fam, shards, dtype, tmp_path):
    cfg, g = load_golden(fam)
    write_model_dir(str(tmp_path), cfg, int(g["seed"]), float(g["std"]), shards=shards, dtype=dtype)
    d = load_desc(str(tmp_path), "bf16")
    want = {n: b for n, b in synth.synth_checkpoint(desc_from_hf_config(cfg, "bf16"), int(g["seed"]), float(g["std"]))}
    got = dict(iter_checkpoint(str(tmp_path)))
    assert set(want) <= set(got)
    for name, bits in want.items():
        a = got[name]
        assert tuple(a.shape) == tuple(bits.shape), name
        ref = synth.bf16_bits_to_f32(bits)
        if dtype == "bf16":
            assert a.dtype == np.uint16 and np.array_equal(a, bits), name
        elif dtype == "fp32":
            assert a.dtype == np.float32 and np.array_equal(a, ref), name
        else:
            assert a.dtype == np.float16 and np.array_equal(a, ref.astype(np.float16)), name
    assert d.hidden == cfg.get("hidden_size", cfg.get("n_embd"))


def test_directory_through_the_oracle_equals_the_synthetic_model(tmp_path, oracle_lib):
    from oracle.oracle_ffi import OracleModel
    cfg, g = load_golden("llama_tiny")
    write_model_dir(str(tmp_path), cfg, int(g["seed"]), float(g["std"]), shards=2)
    d = load_desc(str(tmp_path), "bf16")
    a = OracleModel(d)
    for name, arr in iter_checkpoint(str(tmp_path)):
        a.upload(name, arr, strict=False)
    a.finalize()
    b = OracleModel(d).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
    a.forward(g["prompt"]); b.forward(g["prompt"])
    assert np.array_equal(a.logits(rounded=False), b.logits(rounded=False))


def test_unloadable_dtype_is_skipped_unless_the_model_consumes_it(tmp_path, oracle_lib):
    """A directory that also carries I64 / BOOL buffers (position_ids, attn.bias ...) loads: the reference drops a key no module owns before it looks
    at the dtype (SafeTensors.cpp:176-182 vs :196).  A parameter the model needs in such a dtype is an error that names the tensor."""
    import torch
    from safetensors.torch import load_file, save_file
    from oracle.oracle_ffi import OracleModel
    from tinygpt_amd.checkpoint import UnsupportedTensor
    cfg, g = load_golden("llama_tiny")
    write_model_dir(str(tmp_path), cfg, int(g["seed"]), float(g["std"]))
    d = load_desc(str(tmp_path), "bf16")
    f = str(tmp_path / "model.safetensors")
    t = load_file(f)
    t["model.layers.0.self_attn.rotary_emb.position_ids"] = torch.arange(8, dtype=torch.int64)
    t["model.layers.0.self_attn.masked_bias"] = torch.zeros(4, dtype=torch.bool)
    save_file(t, f)
    got = dict(iter_checkpoint(str(tmp_path)))
    assert isinstance(got["model.layers.0.self_attn.masked_bias"], UnsupportedTensor)
    a = OracleModel(d)
    skipped = [n for n, arr in got.items() if not a.upload(n, arr, strict=False)]
    assert sorted(skipped) == ["model.layers.0.self_attn.masked_bias", "model.layers.0.self_attn.rotary_emb.position_ids"]
    a.finalize()
    b = OracleModel(d).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
    a.forward(g["prompt"]); b.forward(g["prompt"])
    assert np.array_equal(a.logits(rounded=False), b.logits(rounded=False))
    t["model.norm.weight"] = torch.ones(d.hidden, dtype=torch.int64)          # a consumed parameter in a dtype the path does not load
    save_file(t, f)
    c = OracleModel(d)
    with pytest.raises(ValueError, match="model.norm.weight has dtype I64"):
        for n, arr in iter_checkpoint(str(tmp_path)):
            c.upload(n, arr, strict=False)


def test_missing_checkpoint_is_an_error(tmp_path):
    with pytest.raises(FileNotFoundError):
        list(iter_checkpoint(str(tmp_path)))


@pytest.mark.gpu
def test_bench_runs_a_model_directory_on_the_gpu(tmp_path):
    """bench.py --model-dir: hyper-parameters and weights from a HF directory (sharded here) instead of the synthetic ones; the line says so"""
    import json, os, subprocess, sys
    cfg, g = load_golden("llama_tiny")
    write_model_dir(str(tmp_path), cfg, int(g["seed"]), float(g["std"]), shards=2)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--model-dir", str(tmp_path), "--prompt", "9", "--steps", "8", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["data"] == "checkpoint" and line["value"] > 0 and line["roofline"]["traffic"] is None
