"""Builds tinygpt_amd/lib/libtgx_mi355x.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU build container; the .so travels to the
GPU box with the gpurun snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtgx_mi355x.so")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-DNDEBUG"]


def sources():
    out = [os.path.join(CSRC, "tgx_mi355x.hip")]
    deps = list(out) + [os.path.join(HERE, "..", "include", "tgx.h")]
    kd = os.path.join(CSRC, "kernels")
    deps += [os.path.join(kd, f) for f in sorted(os.listdir(kd))]
    return out, deps


def build_lib(force: bool = False, verbose: bool = False, extra_flags=()):
    srcs, deps = sources()
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    if os.environ.get("TGX_DISSECT") == "1":      # experiment build: the debug.gemv / debug.attn switches become live
        extra_flags = list(extra_flags) + ["-DTGX_DISSECT=1"]
    cmd = [HIPCC] + FLAGS + list(extra_flags) + srcs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


HOST = os.path.join(HERE, "host")
HOST_LIB = os.path.join(LIBDIR, "libtgx_host.so")
HOST_CLI = os.path.join(LIBDIR, "tgx_cli")
CXX = shutil.which("g++") or "g++"
CXXFLAGS = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-exceptions", "-Wall", "-Wextra", "-pthread"]


def build_host(force: bool = False, verbose: bool = False):
    """The C++ host engine (no HIP needed: it dlopen()s the device shim): libtgx_host.so + the tgx_cli binary."""
    srcs = [os.path.join(HOST, f) for f in ("loader.cpp", "engine.cpp", "regex.cpp", "tokenizer.cpp")]
    deps = [os.path.join(HOST, f) for f in os.listdir(HOST)] + [os.path.join(HERE, "..", "include", "tgx.h")]
    os.makedirs(LIBDIR, exist_ok=True)
    for target, extra in ((HOST_LIB, [os.path.join(HOST, "engine_c.cpp"), "-shared"]), (HOST_CLI, [os.path.join(HOST, "main.cpp")])):
        if not force and os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps):
            continue
        cmd = [CXX] + CXXFLAGS + srcs + extra + ["-o", target, "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return HOST_LIB, HOST_CLI


if __name__ == "__main__":
    print(build_host(force="-f" in sys.argv, verbose=True))
    print(build_lib(force="-f" in sys.argv, verbose=True))
