"""Deterministic synthetic checkpoints, keyed by HF tensor name.

No real checkpoint is available offline, so every benchmark and parity model is
filled from a counter-based integer hash: the same (seed, tensor name, element index) gives the
same bf16 bit pattern on every machine, with no dependence on libm or on numpy's distribution
code.  Values are uniform in (-a, a) with a = std*sqrt(3); norm weights are 1 + U(-0.1, 0.1) so a
wrong norm weight is visible in parity tests.  All tensors are produced as bf16 bit patterns
(uint16) — the storage dtype of the BASELINE configs — and widened exactly when fp32 is wanted.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .desc import ModelDesc

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_WORKERS = max(1, min(32, (os.cpu_count() or 1)))


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def f32_to_bf16_bits(f: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bit pattern (uint16); NaN kept quiet."""
    u = np.ascontiguousarray(f, dtype=np.float32).view(np.uint32)
    rounded = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    nan = (u & np.uint32(0x7FFFFFFF)) > np.uint32(0x7F800000)
    return np.where(nan, (u >> np.uint32(16)) | np.uint32(0x40), rounded).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


_native = None


def _native_fill():
    """tgxe_synth_tensor of the C++ host library (tinygpt_amd/host/engine.cpp): the same integer hash, ~25x faster than the numpy form
    below on large tensors (7B parameters: seconds instead of minutes).  Bit-identity of the two is a test
    (tests/test_host_engine.py::test_synth_cpp_equals_python); TGX_SYNTH_NUMPY=1 forces the numpy form."""
    global _native
    if _native is None:
        _native = False
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libtgx_host.so")
        if os.environ.get("TGX_SYNTH_NUMPY") != "1" and os.path.exists(path):
            import ctypes
            try:
                fn = ctypes.CDLL(path).tgxe_synth_tensor
                fn.argtypes = [ctypes.c_uint64, ctypes.c_char_p, ctypes.c_int64, ctypes.c_double, ctypes.POINTER(ctypes.c_uint16)]
                fn.restype = None
                _native = fn
            except (OSError, AttributeError):
                _native = False
    return _native


def synth_tensor_bf16(seed: int, name: str, shape, std: float, force_numpy: bool = False) -> np.ndarray:
    """uint16 (bf16 bits) tensor of `shape` for checkpoint entry `name`."""
    n = int(np.prod(shape))
    fn = None if force_numpy else _native_fill()
    if fn and n >= (1 << 16):
        import ctypes
        out = np.empty(n, dtype=np.uint16)
        fn(seed & 0xFFFFFFFFFFFFFFFF, name.encode(), n, float(std), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)))
        return out.reshape(shape)
    base = np.uint64(_fnv1a64(name) ^ ((seed * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF))
    out = np.empty(n, dtype=np.uint16)
    leaf = name.rsplit(".", 2)[-2:]
    is_norm = leaf[-1] == "weight" and (leaf[0].endswith("norm") or leaf[0].endswith("layernorm")
                                        or leaf[0] in ("ln_1", "ln_2", "ln_f"))
    if is_norm:
        lo, a = np.float32(1.0), np.float32(0.1)
    else:
        lo, a = np.float32(0.0), np.float32(std * 1.7320508)
    CH = 1 << 21

    def fill(s):
        e = min(n, s + CH)
        with np.errstate(over="ignore"):
            idx = np.arange(s, e, dtype=np.uint64)
            z = _splitmix64((base + idx * np.uint64(0x9E3779B97F4A7C15)) & _M64)
        u24 = (z >> np.uint64(40)).astype(np.float32)                  # 24 random bits, exact in fp32
        f = (u24 * np.float32(2.0 ** -24) - np.float32(0.5)) * (np.float32(2.0) * a) + lo
        out[s:e] = f32_to_bf16_bits(f)

    starts = range(0, n, CH)
    if n <= CH:
        fill(0)
    else:   # numpy releases the GIL inside ufuncs: chunks scale across host cores
        with ThreadPoolExecutor(max_workers=_WORKERS) as ex:
            list(ex.map(fill, starts))
    return out.reshape(shape)


PEAK_ROWS, PEAK_SCALE = 4, 256.0


def peaked_rows(seed: int, vocab: int):
    """(loud, mirror): PEAK_ROWS + PEAK_ROWS distinct vocabulary rows of a *peaked* synthetic checkpoint, hashed from the seed"""
    base = np.uint64((seed * 0xC2B2AE3D27D4EB4F + 0x165667B19E3779F9) & 0xFFFFFFFFFFFFFFFF)
    picked, k = [], 0
    while len(picked) < 2 * PEAK_ROWS:
        with np.errstate(over="ignore"):
            z = _splitmix64(np.array([(int(base) + k * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
        r = int(z % np.uint64(vocab)); k += 1
        if r not in picked:
            picked.append(r)
    return picked[:PEAK_ROWS], picked[PEAK_ROWS:]


def synth_checkpoint(desc: ModelDesc, seed: int = 1234, std: float = 0.02, peaked: bool = False):
    """Yield (hf_name, uint16 bf16-bits ndarray) for every tensor of `desc`, in checkpoint order.

    peaked=True (tests/test_hip_parity_bar.py, tools/verify_checkpoint.py --synthetic-peaked; never the bench; untied heads only): PEAK_ROWS rows of
    lm_head are scaled by PEAK_SCALE (a power of two: exact in bf16) and PEAK_ROWS other rows hold their NEGATION.  Every step's winner is then the loud
    row with the largest |row . h| (or its mirror, by sign): the logits of 128 256 random rows have a top-2 gap under 2e-3 of the maximum at one step in
    twenty, these 2 x 4 at one step in four hundred — which is what lets "greedy tokens identical" be tested free-running over a whole decode
    (a prompt seed whose 257 gaps all clear the bar is found in a few tries; the test asserts the bar on the oracle's own logits)."""
    shapes = desc.tensor_shapes()
    if peaked and "lm_head.weight" not in shapes:
        raise ValueError("peaked synthetic checkpoints need an untied lm_head (a loud tied embedding row dominates its own residual stream)")
    for name, shape in shapes.items():
        bits = synth_tensor_bf16(seed, name, shape, std)
        if peaked and name == "lm_head.weight":
            loud, mirror = peaked_rows(seed, shape[0])
            for lo, mi in zip(loud, mirror):
                bits[lo] = f32_to_bf16_bits(bf16_bits_to_f32(bits[lo]) * np.float32(PEAK_SCALE))
                bits[mi] = bits[lo] ^ np.uint16(0x8000)
        yield name, bits


def synth_prompt(vocab: int, length: int, seed: int = 1234) -> np.ndarray:
    """Prompt ids uniform in [0, vocab), int64."""
    base = np.uint64((seed * 0xA24BAED4963EE407) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        z = _splitmix64((base + np.arange(length, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) & _M64)
    return ((z >> np.uint64(33)) % np.uint64(vocab)).astype(np.int64)
