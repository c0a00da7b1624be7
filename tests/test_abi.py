"""The drop-in boundary: libkmdiff_hip.so must load without a GPU and export every symbol
include/kmdiff_hip.h declares (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from kmdiff_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(header="kmdiff_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(kmd_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_what_python_binds():
    assert header_functions() == sorted(N.SIGNATURES)
    # the test hooks have a header of their own: none of them in the interface a host binds
    assert header_functions("kmdiff_hip_test.h") == sorted(N.TEST_SIGNATURES)
    assert not [f for f in header_functions() if f.startswith("kmd_test_")]


def test_library_exports_every_declared_symbol():
    L = C.CDLL(N.LIB_PATH)
    for name in header_functions() + header_functions("kmdiff_hip_test.h"):
        assert hasattr(L, name), name


def test_library_loads_and_reports_version_without_gpu():
    L = N.lib()
    assert L.kmd_abi_version() == 3 == N.ABI_VERSION
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


def test_command_line_fails_loudly_without_a_device(tmp_path):
    """`kmdiff-hip diff` has no CPU path either: without a GPU it says so and exits 1 (and `--help` works)."""
    import subprocess
    import kmdiff_amd as K
    cli = os.path.join(ROOT, "kmdiff_amd", "bin", "kmdiff-hip")
    if not os.path.exists(cli):
        subprocess.run(["make", "-C", os.path.join(ROOT, "kmdiff_amd", "host")], check=True, capture_output=True)
    h = subprocess.run([cli, "diff", "--help"], capture_output=True, text=True, timeout=60)
    assert h.returncode == 0 and "--nb-controls" in h.stdout
    u = subprocess.run([cli, "diff", "-d", "x", "-1", "1", "-2", "1", "--no-such-flag"], capture_output=True, text=True, timeout=60)
    assert u.returncode == 1 and "unknown option" in u.stderr
    # --covariates (src/cli.cpp:310): parsed like the reference does (the value must be a file) and refused with the reason
    c = subprocess.run([cli, "diff", "-d", "x", "-1", "1", "-2", "1", "--pop-correction", "--covariates", __file__], capture_output=True, text=True, timeout=60)
    assert c.returncode == 1 and "unknown option" not in c.stderr and "popstrat.cpp:207" in c.stderr
    c = subprocess.run([cli, "diff", "-d", "x", "-1", "1", "-2", "1", "--covariates", "/no/such/file"], capture_output=True, text=True, timeout=60)
    assert c.returncode == 1 and "is not a file" in c.stderr
    if K.device_count() > 0:
        pytest.skip("GPU present")
    r = subprocess.run([cli, "diff", "-d", os.path.join(ROOT, "tests", "golden", "km_out_dir"), "-1", "1", "-2", "1", "-o", str(tmp_path / "o")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr and "no CPU path" in r.stderr
    assert not (tmp_path / "o" / "control_kmers.fasta").exists()


def test_rccl_transport_library_exports_its_header():
    """libkmdiff_hip_rccl.so (the RCCL transport of kmd_correct_sharded) loads without a GPU and exports what
    include/kmdiff_hip_rccl.h declares; libkmdiff_hip.so itself does not depend on librccl."""
    import subprocess
    assert header_functions("kmdiff_hip_rccl.h") == sorted(N.RCCL_SIGNATURES)
    L = N.rccl_lib()
    for name in N.RCCL_SIGNATURES:
        assert hasattr(L, name), name
    needed = subprocess.run(["readelf", "-d", N.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed
    needed = subprocess.run(["readelf", "-d", N.RCCL_LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" in needed


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: the three headers compile as C99 (no C++ in the signatures, no torch types)."""
    import subprocess
    src = tmp_path / "c_abi.c"
    src.write_text('#include "kmdiff_hip.h"\n#include "kmdiff_hip_rccl.h"\n#include "kmdiff_hip_test.h"\n'
                   'int main(void) { kmd_transport t; kmd_survivors s; kmd_tile x; (void)t; (void)s; (void)x; return KMD_PACK_BLOCK == 256 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_abort_trace_reaches_the_descriptor_it_is_given(tmp_path):
    """KMD_ABORT_TRACE=fd:<n> (tests/conftest.py gives the library pytest's saved real stderr): a process that aborts with
    the library loaded leaves the native backtrace of the aborting thread there -- here a file -- and still dies of SIGABRT."""
    import subprocess
    import sys
    out = tmp_path / "trace.txt"
    code = "\n".join(["import os, sys",
                      "fd = os.open(%r, os.O_WRONLY | os.O_CREAT, 0o644)" % str(out),
                      "os.environ['KMD_ABORT_TRACE'] = 'fd:%d' % fd",
                      "sys.path.insert(0, %r)" % ROOT,
                      "from kmdiff_amd import _native",
                      "_native.lib()",
                      "os.abort()"])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == -6, (r.returncode, r.stderr[-500:])
    text = out.read_text()
    assert "[kmdiff_hip] SIGABRT: native backtrace" in text and "abort" in text


def test_conftest_names_a_descriptor_for_the_abort_trace():
    v = os.environ.get("KMD_ABORT_TRACE", "")
    assert v.startswith("fd:") or v.startswith("/") or v == "1", v
    if v.startswith("fd:"):
        os.fstat(int(v[3:]))                      # open
