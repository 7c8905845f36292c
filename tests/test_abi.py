"""The C-ABI library loads (no GPU needed) and exports exactly what include/pdmp_mi355.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="pdmp_mi355.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pdmp_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(pkg):
    assert declared_functions() == sorted(pkg._lib.EXPORTED_SYMBOLS)
    assert declared_functions("pdmp_debug.h") == sorted(pkg._lib.DEBUG_SYMBOLS)  # diagnostics live in their own header
    assert not [f for f in declared_functions() if f.startswith("pdmp_debug")]


def test_library_reads_no_environment(pkg):
    """The ABI promises "no globals": no getenv in the library sources (diagnostics are per-ensemble calls, include/pdmp_debug.h)."""
    csrc = os.path.join(ROOT, "zigzagboomerang.jl_amd", "csrc")
    for f in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, f), errors="replace").read(), f


def test_library_loads_and_exports_every_symbol(pkg):
    pkg.build.build()
    L = ctypes.CDLL(pkg._lib.lib_path())
    for name in declared_functions() + declared_functions("pdmp_debug.h"):
        assert hasattr(L, name), name
    assert pkg._lib.load().pdmp_abi_version() == pkg._lib.ABI_VERSION == 3


def test_struct_sizes(pkg):
    assert ctypes.sizeof(pkg._lib.PdmpConfig) == 48
    assert pkg._lib.EVENT_DTYPE.itemsize == 32 and pkg._lib.COUNTERS_DTYPE.itemsize == 72
    assert ctypes.sizeof(pkg._lib.Config1d) == 88 and pkg._lib.EVENT1D_DTYPE.itemsize == 24 and pkg._lib.STATE1D_DTYPE.itemsize == 96  # pdmp_1d_*


def test_no_cpu_fallback_without_device(pkg):
    """On a box without a GPU, creating an ensemble must fail loudly (PDMP_ERR_NO_DEVICE), never emulate."""
    if pkg._lib.device_count() > 0:
        pytest.skip("a gfx950 device is present")
    with pytest.raises(pkg._lib.PdmpError) as ei:
        pkg.Ensemble(1, 4)
    assert ei.value.code == 2


def test_product_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "zigzagboomerang.jl_amd")
    for dp, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle_lib" not in txt and "liboracle" not in txt, (dp, f)
                assert not re.search(r'#include\s*[<"][^>"]*oracle', txt), (dp, f)
                assert not re.search(r'^\s*(from|import)\s+\S*oracle', txt, flags=re.M), (dp, f)


def test_headers_are_plain_c99(tmp_path):
    """include/pdmp_mi355.h and include/pdmp_detmath.h must compile as C99 (the boundary is a C ABI: cgo / ccall / ctypes bind it),
    with the struct sizes the bindings assume."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    src = tmp_path / "t.c"
    src.write_text('#include "pdmp_mi355.h"\n#include "pdmp_debug.h"\n#include "pdmp_detmath.h"\n'
                   "int main(void){ return (int)(sizeof(pdmp_event) != 32) + (int)(sizeof(pdmp_chain_counters) != 72) + "
                   "(int)(sizeof(pdmp_config) != 48) + (int)(pdmp_log(1.0) != 0.0); }\n")
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-ffp-contract=off", "-I", inc, str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0
