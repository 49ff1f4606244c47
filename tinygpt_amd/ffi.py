"""ctypes binding of the C ABI in include/tgx.h and the host-side model wrapper used by
bench.py and the parity tests.

`Model` mirrors the engine-facing interface of the reference's GPTModel
(src/model/GPTModel.h:80-106: forward / resetCache / load / numLayers / contextSize) plus
Sampler::sample (src/engine/Sampler.cpp:23-79) on top of one opaque `tgx_ctx`.

The product library is tinygpt_amd/lib/libtgx_mi355x.so (built by tinygpt_amd.build); loading
fails loudly if it is missing — there is no CPU fallback in this package.  `Backend` binds any
library exporting the same entry points under a symbol prefix (the test infrastructure reuses it).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

import numpy as np

from .desc import CDesc, ModelDesc, DTYPE_BF16, DTYPE_F32, DTYPE_F16
from . import synth

HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(HERE, "lib", "libtgx_mi355x.so")

STATUS_NAMES = {0: "TGX_OK", 1: "TGX_ERR_INVALID", 2: "TGX_ERR_UNSUPPORTED", 3: "TGX_ERR_DEVICE", 4: "TGX_ERR_STATE",
                5: "TGX_ERR_NOMEM", 6: "TGX_ERR_NAME", 7: "TGX_ERR_SHAPE", 8: "TGX_ERR_CONTEXT"}

KERNEL_CLASSES = ["qkv", "attn", "oproj", "gateup", "down", "lmhead"]


class SamplerCfg(ctypes.Structure):
    """struct tgx_sampler_cfg == SamplerConfig (src/engine/Sampler.h:13-22)."""
    _fields_ = [("temperature", c_float), ("top_k", c_int64), ("top_p", c_float), ("min_p", c_float)]

    def __init__(self, temperature=0.0, top_k=0, top_p=1.0, min_p=0.0):
        super().__init__(temperature, top_k, top_p, min_p)

    @property
    def greedy(self):   # Sampler.cpp:15-21
        return not (self.temperature > 0 or self.top_k > 0 or self.top_p < 1 or self.min_p > 0)


GREEDY = SamplerCfg()


class TgxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {msg}")
        self.status = status


# name -> (restype, argtypes); every symbol include/tgx.h declares
ABI = {
    "device_count": (c_int, [POINTER(c_int)]),
    "create": (c_int, [POINTER(CDesc), c_int, POINTER(c_void_p)]),
    "upload": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int, c_int]),
    "finalize": (c_int, [c_void_p]),
    "destroy": (None, [c_void_p]),
    "forward": (c_int, [c_void_p, POINTER(c_int64), c_int, c_int]),
    "read_logits": (c_int, [c_void_p, POINTER(c_float), c_int]),
    "sample": (c_int, [c_void_p, POINTER(SamplerCfg), c_uint64, POINTER(c_int64)]),
    "decode": (c_int, [c_void_p, POINTER(SamplerCfg), c_uint64, c_int, POINTER(c_int64)]),
    "step_async": (c_int, [c_void_p, POINTER(SamplerCfg), c_uint64, POINTER(c_int64)]),
    "fetch_token": (c_int, [c_void_p, c_int64, POINTER(c_int32)]),
    "reset_cache": (c_int, [c_void_p]),
    "past_length": (c_int64, [c_void_p]),
    "reset_row": (c_int, [c_void_p, c_int]),
    "forward_row": (c_int, [c_void_p, c_int, POINTER(c_int64), c_int]),
    "sample_row": (c_int, [c_void_p, c_int, POINTER(SamplerCfg), c_uint64, POINTER(c_int64)]),
    "past_length_row": (c_int64, [c_void_p, c_int]),
    "context_size": (c_int64, [c_void_p]),
    "num_layers": (c_int32, [c_void_p]),
    "last_error": (c_char_p, [c_void_p]),
    "synchronize": (c_int, [c_void_p]),
    "read_kv": (c_int, [c_void_p, c_int, c_int, POINTER(c_float), POINTER(c_float)]),
    "write_kv": (c_int, [c_void_p, c_int, c_int, POINTER(c_float), POINTER(c_float), c_int64]),
    "profile_decode": (c_int, [c_void_p, c_int, POINTER(c_int64), POINTER(c_double)]),
    "set_option": (c_int, [c_void_p, c_char_p, c_int]),
    "get_option": (c_int, [c_void_p, c_char_p, POINTER(c_int)]),
    "read_probs": (c_int, [c_void_p, POINTER(c_float)]),
    "set_logits": (c_int, [c_void_p, POINTER(c_float), c_int]),
    "bytes_per_token": (c_int64, [c_void_p, c_int64]),
    "abi_version": (c_int, []),
}


class Backend:
    """A loaded shared library exporting `<prefix><name>` for (a subset of) ABI."""

    def __init__(self, path: str, prefix: str = "tgx_", required=None, extra=None):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found — build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                "tinygpt_amd has no CPU fallback.")
        self.path = path
        self.prefix = prefix
        self.lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        table = dict(ABI)
        if extra:
            table.update(extra)
        names = required if required is not None else list(ABI)
        if extra:
            names = list(names) + list(extra)
        for name in names:
            res, args = table[name]
            fn = getattr(self.lib, prefix + name)   # AttributeError if the symbol is missing: loud by design
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def has(self, name):
        return hasattr(self, name)


_product = None


def product_backend() -> Backend:
    global _product
    if _product is None:
        _product = Backend(PRODUCT_LIB, "tgx_")
    return _product


def _as_bits(arr: np.ndarray):
    """ndarray -> (contiguous array, tgx_dtype).  uint16 is taken as bf16 bit patterns."""
    a = np.ascontiguousarray(arr)
    if a.dtype == np.uint16:
        return a, DTYPE_BF16
    if a.dtype == np.float32:
        return a, DTYPE_F32
    if a.dtype == np.float16:
        return a.view(np.uint16), DTYPE_F16
    raise TypeError(f"unsupported tensor dtype {a.dtype}")


class Model:
    """One model instance on one device behind the C ABI."""

    def __init__(self, desc: ModelDesc, backend: Backend | None = None, device: int = 0):
        self.be = backend or product_backend()
        self.desc = desc
        self._ctx = c_void_p()
        cd = desc.to_c()
        st = self.be.create(ctypes.byref(cd), device, ctypes.byref(self._ctx))
        if st != 0:
            msg = self.be.last_error(self._ctx if self._ctx else None)
            raise TgxError(st, (msg or b"").decode())
        self.batch = 0

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, st):
        if st != 0:
            raise TgxError(st, (self.be.last_error(self._ctx) or b"").decode())

    def close(self):
        if self._ctx:
            self.be.destroy(self._ctx)
            self._ctx = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- load (== GPTModel::load -> SafeTensors::load) ---------------------------------------
    def upload(self, name: str, arr: np.ndarray, strict: bool = True):
        if hasattr(arr, "error") and hasattr(arr, "dtype") and not isinstance(arr, np.ndarray):
            # checkpoint.UnsupportedTensor: a file dtype this path cannot convert (I64 / BOOL / U8 ...).  The ABI's name probe (ndim < 0, include/tgx.h)
            # tells a parameter the model needs (TGX_ERR_SHAPE: raise) from a key it ignores or does not know (skipped like the reference, SafeTensors.cpp:176-182)
            one = (c_int64 * 1)(0)
            st = self.be.upload(self._ctx, name.encode(), ctypes.cast(one, c_void_p), one, -1, 0)   # 0 = TGX_F32
            if st == 7 or (st == 6 and strict):
                raise arr.error()
            return False
        a, dt = _as_bits(arr)
        shape = (c_int64 * a.ndim)(*a.shape)
        st = self.be.upload(self._ctx, name.encode(), a.ctypes.data_as(c_void_p), shape, a.ndim, dt)
        if st == 6 and not strict:   # "Unexpected key" is a warning in the reference (non-strict load, GPTModel.h:96)
            return False
        self._check(st)
        return True

    def load_synthetic(self, seed: int = 1234, std: float = 0.02, peaked: bool = False):
        for name, bits in synth.synth_checkpoint(self.desc, seed, std, peaked):
            self.upload(name, bits)
        return self

    def load_state(self, tensors: "dict[str, np.ndarray]", strict: bool = False):
        for name, arr in tensors.items():
            self.upload(name, arr, strict=strict)
        return self

    def finalize(self):
        self._check(self.be.finalize(self._ctx))
        return self

    # -- hot path ---------------------------------------------------------------------------
    def forward(self, ids):
        ids = np.ascontiguousarray(np.atleast_2d(np.asarray(ids, dtype=np.int64)))
        B, S = ids.shape
        self._check(self.be.forward(self._ctx, ids.ctypes.data_as(POINTER(c_int64)), B, S))
        self.batch = B
        return self

    def logits(self, rounded: bool = True) -> np.ndarray:
        out = np.empty((self.batch, self.desc.vocab), dtype=np.float32)
        self._check(self.be.read_logits(self._ctx, out.ctypes.data_as(POINTER(c_float)), int(rounded)))
        return out

    def sample(self, cfg: SamplerCfg = GREEDY, seed: int = 0) -> np.ndarray:
        out = np.empty(self.batch, dtype=np.int64)
        self._check(self.be.sample(self._ctx, ctypes.byref(cfg), seed, out.ctypes.data_as(POINTER(c_int64))))
        return out

    def decode(self, n_steps: int, cfg: SamplerCfg = GREEDY, seed: int = 0, fetch: bool = True):
        out = np.empty((n_steps, self.batch), dtype=np.int64) if fetch else None
        ptr = out.ctypes.data_as(POINTER(c_int64)) if fetch else None
        self._check(self.be.decode(self._ctx, ctypes.byref(cfg), seed, n_steps, ptr))
        return out

    def step_async(self, cfg: SamplerCfg = GREEDY, seed: int = 0) -> int:
        t = c_int64()
        self._check(self.be.step_async(self._ctx, ctypes.byref(cfg), seed, ctypes.byref(t)))
        return t.value

    def fetch_token(self, ticket: int) -> int:
        v = c_int32()
        self._check(self.be.fetch_token(self._ctx, ticket, ctypes.byref(v)))
        return v.value

    def reset_cache(self):
        self._check(self.be.reset_cache(self._ctx))

    # -- per-row sequence lifecycle (include/tgx.h, ABI 3) ------------------------------------
    def reset_row(self, row: int):
        self._check(self.be.reset_row(self._ctx, row))
        return self

    def forward_row(self, row: int, ids):
        """prefill one prompt into `row` of the live batch (a retired row, or the next free one); the other rows keep their state"""
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64).reshape(-1))
        self._check(self.be.forward_row(self._ctx, row, ids.ctypes.data_as(POINTER(c_int64)), len(ids)))
        self.batch = max(self.batch, row + 1)
        return self

    def sample_row(self, row: int, cfg: SamplerCfg = GREEDY, seed: int = 0) -> int:
        out = c_int64()
        self._check(self.be.sample_row(self._ctx, row, ctypes.byref(cfg), seed, ctypes.byref(out)))
        return out.value

    def past_length_row(self, row: int) -> int:
        return self.be.past_length_row(self._ctx, row)

    @property
    def past_length(self) -> int:
        return self.be.past_length(self._ctx)

    @property
    def context_size(self) -> int:
        return self.be.context_size(self._ctx)

    def synchronize(self):
        self._check(self.be.synchronize(self._ctx))

    def read_kv(self, row: int, layer: int):
        T = self.be.past_length_row(self._ctx, row) if self.be.has("past_length_row") else self.past_length
        shape = (T, self.desc.kv_heads, self.desc.head_dim)
        k = np.empty(shape, dtype=np.float32)
        v = np.empty(shape, dtype=np.float32)
        self._check(self.be.read_kv(self._ctx, row, layer, k.ctypes.data_as(POINTER(c_float)), v.ctypes.data_as(POINTER(c_float))))
        return k, v

    def write_kv(self, row: int, layer: int, k: np.ndarray, v: np.ndarray):
        """overwrite cache rows [0, len(k)) of (row, layer) from fp32 [T][kv_heads][head_dim] arrays (the layout read_kv returns)"""
        k = np.ascontiguousarray(k, dtype=np.float32); v = np.ascontiguousarray(v, dtype=np.float32)
        assert k.shape == v.shape and k.shape[1:] == (self.desc.kv_heads, self.desc.head_dim)
        self._check(self.be.write_kv(self._ctx, row, layer, k.ctypes.data_as(POINTER(c_float)), v.ctypes.data_as(POINTER(c_float)), k.shape[0]))

    def profile_decode(self, n_steps: int):
        n = len(KERNEL_CLASSES)
        launches = (c_int64 * n)()
        ms = (c_double * n)()
        self._check(self.be.profile_decode(self._ctx, n_steps, launches, ms))
        return {k: (launches[i], ms[i]) for i, k in enumerate(KERNEL_CLASSES)}

    def probs(self) -> np.ndarray:
        out = np.empty((self.batch, self.desc.vocab), dtype=np.float32)
        self._check(self.be.read_probs(self._ctx, out.ctypes.data_as(POINTER(c_float))))
        return out

    def set_logits(self, logits):
        l = np.ascontiguousarray(np.atleast_2d(np.asarray(logits, dtype=np.float32)))
        self._check(self.be.set_logits(self._ctx, l.ctypes.data_as(POINTER(c_float)), l.shape[0]))
        self.batch = l.shape[0]
        return self

    def set_option(self, key: str, value: int):
        self._check(self.be.set_option(self._ctx, key.encode(), int(value)))
        return self

    def get_option(self, key: str) -> int:
        out = c_int(0)
        self._check(self.be.get_option(self._ctx, key.encode(), ctypes.byref(out)))
        return out.value

    def bytes_per_token(self, T: int) -> int:
        return self.be.bytes_per_token(self._ctx, T)
