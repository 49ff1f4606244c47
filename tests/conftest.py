import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The oracle is OpenMP code checking tiny fixtures: on a 256-thread host libgomp's default team makes every parallel region
# a 256-way barrier (minutes per test on the GPU box).  Must be set before liboracle.so (libgomp) is loaded.
os.environ.setdefault("OMP_NUM_THREADS", "8")

GOLDEN = os.path.join(ROOT, "tests", "golden")
FAMILIES = ["llama_tiny", "qwen2_tiny", "mistral_tiny", "qwen3_tiny", "gpt2_tiny", "gpt2_hd64"]
# gpt2_tiny (head_dim 32) stays an oracle-only fixture; gpt2_hd64 has the head_dim of every released GPT-2 size
GPU_FAMILIES = ["llama_tiny", "qwen2_tiny", "mistral_tiny", "qwen3_tiny", "gpt2_hd64"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(fam):
    with open(os.path.join(GOLDEN, fam, "config.json")) as f:
        cfg = json.load(f)
    return cfg, np.load(os.path.join(GOLDEN, fam, "golden.npz"))


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle.oracle_ffi import build_oracle, oracle_backend
    build_oracle()
    return oracle_backend()


def rel_err(a, b):
    """max |a-b| / max |b| — the 'relative fp32' measure used for every logits tolerance."""
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))
