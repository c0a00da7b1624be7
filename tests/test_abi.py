"""The drop-in boundary: libkmdiff_hip.so must load without a GPU and export every symbol
include/kmdiff_hip.h declares (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from kmdiff_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "kmdiff_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(kmd_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_what_python_binds():
    assert header_functions() == sorted(N.SIGNATURES)


def test_library_exports_every_declared_symbol():
    L = C.CDLL(N.LIB_PATH)
    for name in header_functions():
        assert hasattr(L, name), name


def test_library_loads_and_reports_version_without_gpu():
    L = N.lib()
    assert L.kmd_abi_version() == 1
    assert L.kmd_status_string(0) == b"ok"
    assert L.kmd_status_string(-4) == b"survivor capacity exceeded"


def test_no_cpu_fallback_without_device():
    """On a box without a GPU every compute entry point must fail loudly."""
    import kmdiff_amd as K
    if K.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(K.KmdError):
        K.PoissonLikelihood(2, 2, [1, 1], [1, 1], 10)
    with pytest.raises(K.KmdError):
        K.DeviceBuffer(16)


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "kmdiff_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "kmd_oracle" not in txt and "oracle/" not in txt.replace(
                    "CPU oracle", ""), os.path.join(dirpath, f)
