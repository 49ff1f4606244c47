"""Reads a HF model directory the way the reference's loader does (src/huggingface/ModelLoader.cpp:18-89, SafeTensors.cpp:141-229):
`model.safetensors`, else the shards of `model.safetensors.index.json`; BF16 / F16 / F32 tensors, yielded by checkpoint name as bit arrays that
Model.upload takes (uint16 = bf16 bits, float16, float32).  The C++ engine has its own loader (host/loader.cpp); this one serves bench.py and
the Python tools.  No torch, no safetensors package: the format is an 8-byte header length, a JSON header, raw little-endian data."""
from __future__ import annotations

import json
import mmap
import os
import struct

import numpy as np

class UnsupportedTensor:
    """a checkpoint entry whose dtype the path does not load; an error only if the model actually consumes it"""
    def __init__(self, path, name, dtype):
        self.path, self.name, self.dtype = path, name, dtype

    def error(self):
        return ValueError(f"{self.path}: tensor {self.name} has dtype {self.dtype} (BF16 / F16 / F32 are loaded)")


_DT = {"BF16": (np.uint16, 2), "F16": (np.float16, 2), "F32": (np.float32, 4)}


def _read_file(path: str):
    with open(path, "rb") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        header = json.loads(f.read(n))
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    base = 8 + n
    for name, info in header.items():
        if name == "__metadata__":
            continue
        if info["dtype"] not in _DT:
            # the reference drops a key no module owns before it looks at the dtype (SafeTensors.cpp:176-182 vs :196): I64 / BOOL / U8 buffers
            # (position_ids, attn.bias, masked_bias ...) must not abort the load.  Yielded as an UnsupportedTensor marker: Model.upload(strict=False)
            # skips it like any other unknown key, and raises only if the model consumes that name.
            yield name, UnsupportedTensor(path, name, info["dtype"])
            continue
        dt, esz = _DT[info["dtype"]]
        b0, b1 = info["data_offsets"]
        shape = tuple(info["shape"])
        if (b1 - b0) != int(np.prod(shape, dtype=np.int64)) * esz:
            raise ValueError(f"{path}: tensor {name} data size does not match its shape")
        yield name, np.frombuffer(mm, dtype=dt, count=(b1 - b0) // esz, offset=base + b0).reshape(shape)


def iter_checkpoint(model_dir: str):
    """(name, array) for every tensor of the directory's checkpoint, single-file or sharded"""
    single = os.path.join(model_dir, "model.safetensors")
    if os.path.exists(single):
        yield from _read_file(single)
        return
    index = os.path.join(model_dir, "model.safetensors.index.json")
    if not os.path.exists(index):
        raise FileNotFoundError(f"{model_dir}: neither model.safetensors nor model.safetensors.index.json")
    with open(index) as f:
        shards = sorted(set(json.load(f)["weight_map"].values()))
    for s in shards:
        yield from _read_file(os.path.join(model_dir, s))
