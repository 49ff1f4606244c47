"""tinygpt_amd — MI355X-native decode path for TinyGPT behind the C ABI in include/tgx.h.

Holds only what the hot path needs: csrc/ (HIP kernels + the extern "C" shim), host/ (the C++ engine
that mirrors GPTEngine/Sampler/ModelLoader above the shim), and thin Python glue (ctypes binding,
model description, deterministic synthetic checkpoints) for bench.py and the parity tests.
"""
from .desc import ModelDesc, desc_from_hf_config, known_desc, load_desc  # noqa: F401

__version__ = "0.1.0"
