"""The C-ABI library loads on a CPU-only box and exports every symbol include/badger_hip.h declares."""
import ctypes
import os
import re

from badger_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "badger_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bdg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    lib = _native.load()
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_native.EXPORTS)
    assert b"gfx950" in lib.bdg_version()


def test_record_layouts_match_header_and_oracle():
    from oracle import pyoracle as orc
    assert _native.REC_DTYPE == orc.REC_DTYPE and _native.REC_DTYPE.itemsize == 32
    assert _native.EDGE_DTYPE == orc.EDGE_DTYPE and _native.EDGE_DTYPE.itemsize == 12
    assert ctypes.sizeof(_native.KernelTime) == 64


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    lib = _native.load()
    h = ctypes.c_void_p()
    assert lib.bdg_init(0, ctypes.byref(h)) < 0
    try:
        _native.Context(0)
    except _native.BadgerHipError:
        pass
    else:
        raise AssertionError("Context() must raise without a GPU")


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "badger_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                # no import, include, link or dlopen of anything under oracle/ (comments may cite it)
                assert not re.search(r"^\s*(from|import)\s+oracle|pyoracle|#include.*oracle|libbadger_oracle", src, re.M), f


def test_no_sort_or_scan_library_in_the_product():
    """Every kernel of the library is the repo's own: no source under csrc/ includes hipCUB / rocPRIM / thrust, and the built
    library holds no rocprim symbol (round 3's radix sorts, run-length encode and scans are the bucket partition of
    csrc/bdg_partition.hpp now)."""
    import subprocess
    csrc = os.path.join(ROOT, "badger_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".hpp")):
            src = open(os.path.join(csrc, f)).read()
            assert not re.search(r"#include\s*<(hipcub|rocprim|thrust)/", src), f
    so = os.path.join(ROOT, "badger_amd", "libbadger_hip.so")
    names = subprocess.run(["nm", "-C", so], capture_output=True, text=True, check=True).stdout
    assert "rocprim" not in names and "hipcub" not in names
