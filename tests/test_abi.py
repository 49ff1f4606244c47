"""The C-ABI library loads and exports every symbol include/tgx.h declares (no compute: runs without a GPU)."""
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "tgx.h")).read()
    return sorted(set(re.findall(r"TGX_API\s+[\w\s\*]+?\b(tgx_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_all_exported():
    from tinygpt_amd import build
    from tinygpt_amd.ffi import ABI, Backend
    lib = build.build_lib()
    names = declared_symbols()
    assert len(names) >= 20
    assert sorted("tgx_" + n for n in ABI) == names, "ffi.ABI and include/tgx.h disagree"
    be = Backend(lib, "tgx_")              # resolves every symbol or raises AttributeError
    assert be.abi_version() == 3


def test_no_gpu_is_a_loud_error():
    """Without a GPU tgx_create must fail with a device error — never fall back to a CPU path."""
    import ctypes
    from tinygpt_amd.desc import known_desc
    from tinygpt_amd.ffi import Model, TgxError, product_backend
    be = product_backend()
    n = ctypes.c_int(0)
    be.device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(TgxError) as ei:
        Model(known_desc("llama-3.2-1b"), be)
    assert ei.value.status == 3


def test_product_never_touches_the_oracle():
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "tinygpt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", ".c")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|liboracle|tgxo_", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, f"product files reference the oracle: {bad}"


def test_shipped_host_binaries_have_no_backend_hook(tmp_path):
    """The shipped CLI and host library can bind libtgx_mi355x.so only: the --backend-lib / --backend-prefix test hooks exist in the
    -DTGXH_TEST_HOOKS builds under tests/_build alone."""
    import subprocess
    from tinygpt_amd import build
    lib, cli = build.build_host()
    for path in (lib, cli):
        data = open(path, "rb").read()
        assert b"backend-lib" not in data and b"backend-prefix" not in data, path
    out = subprocess.run([cli, "--synthetic", "gpt2", "--backend-lib", "/nonexistent.so"], capture_output=True, text=True)
    assert out.returncode == 1 and "Unknown argument: --backend-lib" in out.stderr
    # the C view refuses a backend override in the shipped build
    import ctypes
    h = ctypes.CDLL(lib)
    h.tgxe_create.restype = ctypes.c_void_p
    h.tgxe_create.argtypes = [ctypes.c_char_p] * 5 + [ctypes.c_int] * 3
    assert h.tgxe_create(b"", b"gpt2", b"mi355x", b"/tmp/other.so", b"tgxo_", 0, 1, 1) is None
    tlib, tcli = build.build_host(test_hooks=True)
    assert b"backend-lib" in open(tcli, "rb").read()
