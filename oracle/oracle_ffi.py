"""ctypes loader for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY — may be
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by tinygpt_amd/."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_float, c_int, c_int64, c_void_p

import numpy as np

from tinygpt_amd.ffi import Backend, Model, SamplerCfg

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(HERE, "liboracle.so")

ORACLE_SYMBOLS = ["create", "upload", "finalize", "destroy", "forward", "read_logits", "sample", "decode",
                  "reset_cache", "past_length", "context_size", "last_error", "read_kv"]
ORACLE_EXTRA = {
    "filter_logits": (c_int, [POINTER(SamplerCfg), POINTER(c_float), POINTER(c_float), c_int]),
    "set_next_token": (c_int, [c_void_p, POINTER(c_int64), c_int]),
    "read_rope": (c_int, [c_void_p, c_int, POINTER(c_float)]),
    "set_logits": (c_int, [c_void_p, POINTER(c_float), c_int]),
    "set_threads": (c_int, [c_int]),
    "set_reorder": (c_int, [c_void_p, c_int]),
    "set_act16": (c_int, [c_void_p, c_int]),
    "set_one_row_dots": (c_int, [c_int]),
    "set_torch_rounding": (c_int, [c_void_p, c_int]),
}


def build_oracle(force: bool = False):
    src = os.path.join(HERE, "tgx_oracle.c")
    if force or not os.path.exists(ORACLE_LIB) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return ORACLE_LIB


_oracle = None


def oracle_backend() -> Backend:
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_LIB):
            build_oracle()
        _oracle = Backend(ORACLE_LIB, "tgxo_", required=ORACLE_SYMBOLS, extra=ORACLE_EXTRA)
    return _oracle


class OracleModel(Model):
    def __init__(self, desc, device: int = 0):
        super().__init__(desc, backend=oracle_backend(), device=device)

    def set_next_token(self, ids):
        ids = np.ascontiguousarray(np.asarray(ids, dtype=np.int64).reshape(-1))
        self._check(self.be.set_next_token(self._ctx, ids.ctypes.data_as(POINTER(c_int64)), len(ids)))
        self.batch = len(ids)

    def set_reorder(self, on: bool = True):
        """every reduction from the last element to the first: a second fp32 schedule of the same arithmetic (test hook)"""
        self._check(self.be.set_reorder(self._ctx, 1 if on else 0))
        return self

    def set_act16(self, on: bool = True):
        """the input of every Linear rounded to the storage dtype first: the counterpart of the library's option act.round16"""
        self._check(self.be.set_act16(self._ctx, 1 if on else 0))
        return self

    def set_torch_rounding(self, on: bool = True):
        """every op output rounded to bf16 (a torch-bf16 module's contract); before finalize()"""
        self._check(self.be.set_torch_rounding(self._ctx, 1 if on else 0))
        return self

    def rope_tables(self, n_pos: int):
        half = self.desc.head_dim // 2
        out = np.empty((2, n_pos, half), dtype=np.float32)
        self._check(self.be.read_rope(self._ctx, n_pos, out.ctypes.data_as(POINTER(c_float))))
        return out[0], out[1]


def filter_logits(cfg: SamplerCfg, logits: np.ndarray):
    """Sampler.cpp:34-77 up to (not including) the multinomial draw: returns (masked logits, probs)."""
    l = np.ascontiguousarray(logits, dtype=np.float32).copy()
    p = np.empty_like(l)
    rc = oracle_backend().filter_logits(ctypes.byref(cfg), l.ctypes.data_as(POINTER(c_float)),
                                        p.ctypes.data_as(POINTER(c_float)), l.size)
    assert rc == 0
    return l, p
