"""The oracle at the REAL Llama-3.2-1B geometry (1.24 B parameters, vocabulary 128 256, llama3 RoPE scaling 32 / 8192) against
HF transformers fp32 (tools/gen_fullsize_fixture.py): top-64 logits and 64 probe logits of a 12-token prompt and 3 forced
steps.  ~25 s on 8 cores (2.5 GB of synthetic bf16 weights widened to fp32 inside the oracle)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from tinygpt_amd import known_desc


@pytest.mark.parametrize("key,fixture", [("llama-3.2-1b", "llama_3_2_1b_full"), ("qwen2.5-0.5b", "qwen2_5_0_5b_full"), ("qwen3-0.6b", "qwen3_0_6b_full"), ("gpt2", "gpt2_124m_full")])
def test_oracle_matches_hf_at_full_geometry(key, fixture, oracle_lib):
    """Also Qwen2.5-0.5B (QKV bias, 14 / 2 heads, theta 1e6, V = 151 936, tied head) at its real size."""
    from oracle.oracle_ffi import OracleModel
    g = np.load(os.path.join(GOLDEN, fixture, "golden.npz"))
    d = known_desc(key, "fp32")
    d.max_ctx = 64                                     # KV capacity only
    m = OracleModel(d).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
    m.forward(g["prompt"])
    for step in range(4):
        l = m.logits(rounded=False)[0]
        scale = float(np.abs(g["top_v"][step]).max())
        assert np.abs(l[g["top_i"][step]] - g["top_v"][step]).max() < 1e-4 * scale, step
        assert np.abs(l[g["probe"]] - g["probe_v"][step]).max() < 1e-4 * scale, step
        assert int(np.argmax(l)) == int(g["top_i"][step][0])
        if step == 3:
            break
        m.forward(np.array([[int(g["forced"][step])]], dtype=np.int64))
    m.close()
