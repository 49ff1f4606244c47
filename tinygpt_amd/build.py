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
         "-Wall", "-Wno-unused-function", "-DNDEBUG",
         # MFMA accumulators in the (unified) VGPR file instead of AGPRs: the softmax / epilogue arithmetic reads them without
         # v_accvgpr_read/write copies (attention prefill: 127 of ~700 loop instructions) and most MFMA kernels need fewer registers
         ] + (["-mllvm", "-amdgpu-mfma-vgpr-form"] if os.environ.get("TGX_VGPR_FORM", "1") == "1" else [])


TUS = ["abi", "decode", "attn", "sampler", "prefill", "prefill_f32", "skinny"]     # csrc/<name>.hip, compiled in parallel (csrc/ctx.h lists what each holds)
OBJDIR = os.path.join(LIBDIR, "obj")


def _deps():
    deps = [os.path.join(HERE, "..", "include", "tgx.h"), os.path.join(CSRC, "ctx.h")]
    kd = os.path.join(CSRC, "kernels")
    return deps + [os.path.join(kd, f) for f in sorted(os.listdir(kd))]


def build_lib(force: bool = False, verbose: bool = False, extra_flags=()):
    """hipcc -c every translation unit (one process each, all at once: the longest is the GEMV instantiations of decode.hip, ~80 s), then link."""
    if os.environ.get("TGX_DISSECT") == "1":      # experiment build: the debug.gemv / debug.attn switches become live
        extra_flags = list(extra_flags) + ["-DTGX_DISSECT=1"]
    if os.environ.get("TGX_EXTRA_FLAGS"):         # experiment builds (e.g. -DTGX_GEMM_TERMS=1, tools/probes/README.md); never the shipped library
        extra_flags = list(extra_flags) + os.environ["TGX_EXTRA_FLAGS"].split()
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra_flags)
    # the flag set is part of the staleness test (a stamp next to the library): toggling TGX_VGPR_FORM / TGX_DISSECT / extra_flags rebuilds
    stamp, want = LIB + ".flags", " ".join([HIPCC] + cflags)
    same_flags = os.path.exists(stamp) and open(stamp).read() == want
    os.makedirs(OBJDIR, exist_ok=True)
    deps = _deps()
    jobs = []
    for tu in TUS:
        src, obj = os.path.join(CSRC, tu + ".hip"), os.path.join(OBJDIR, tu + ".o")
        fresh = same_flags and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in deps + [src])
        if force or not fresh:
            cmd = [HIPCC] + cflags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((tu, subprocess.Popen(cmd)))
    failed = [tu for tu, p in jobs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c " + ", ".join(failed))
    objs = [os.path.join(OBJDIR, tu + ".o") for tu in TUS]
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(LIB) < os.path.getmtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(want)
    return LIB


HOST = os.path.join(HERE, "host")
HOST_LIB = os.path.join(LIBDIR, "libtgx_host.so")
HOST_CLI = os.path.join(LIBDIR, "tgx_cli")
TEST_BUILD = os.path.join(HERE, "..", "tests", "_build")     # test-hook variants live with the tests, not in the product lib dir
HOST_TEST_LIB = os.path.join(TEST_BUILD, "libtgx_host_test.so")
HOST_TEST_CLI = os.path.join(TEST_BUILD, "tgx_cli_test")
CXX = shutil.which("g++") or "g++"
CXXFLAGS = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-exceptions", "-Wall", "-Wextra", "-pthread"]


def build_host(force: bool = False, verbose: bool = False, test_hooks: bool = False):
    """The C++ host engine (no HIP needed: it dlopen()s the device shim): libtgx_host.so + the tgx_cli binary.

    test_hooks=True builds the TEST variants instead (tests/_build/libtgx_host_test.so, tgx_cli_test, -DTGXH_TEST_HOOKS): the only
    builds that can bind a library other than libtgx_mi355x.so (the CPU oracle, for host-logic tests without a GPU)."""
    srcs = [os.path.join(HOST, f) for f in ("loader.cpp", "engine.cpp", "regex.cpp", "tokenizer.cpp")]
    deps = [os.path.join(HOST, f) for f in os.listdir(HOST)] + [os.path.join(HERE, "..", "include", "tgx.h")]
    lib, cli, flags = HOST_LIB, HOST_CLI, []
    if test_hooks:
        lib, cli, flags = HOST_TEST_LIB, HOST_TEST_CLI, ["-DTGXH_TEST_HOOKS"]
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    for target, extra in ((lib, [os.path.join(HOST, "engine_c.cpp"), "-shared"]), (cli, [os.path.join(HOST, "main.cpp")])):
        if not force and os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps):
            continue
        cmd = [CXX] + CXXFLAGS + flags + srcs + extra + ["-o", target, "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return lib, cli


if __name__ == "__main__":
    print(build_host(force="-f" in sys.argv, verbose=True))
    print(build_host(force="-f" in sys.argv, verbose=True, test_hooks=True))
    print(build_lib(force="-f" in sys.argv, verbose=True))
