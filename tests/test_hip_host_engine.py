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
