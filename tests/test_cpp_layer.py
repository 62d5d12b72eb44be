"""The header-only C++ layer (cusp-autotuned_amd/include/cusp) -- the host side of the drop-in
boundary.  tests/cpp/*.cpp are written like the reference's own tests (same matrices and protocol,
see tests/cpp/spmv_tests.h); this file builds them with g++ and runs them:
  * test_host   (host_memory containers, conversions, CG, MatrixMarket)      -- CPU, -m "not gpu"
  * test_device (device_memory containers -> C-ABI -> gfx950 kernels)        -- -m gpu
"""
import os
import subprocess

import pytest

from conftest import ROOT

CPP = os.path.join(ROOT, "tests", "cpp")


def _build():
    r = subprocess.run(["make", "-C", CPP, "-j2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def _run(name, timeout=600):
    exe = os.path.join(CPP, "bin", name)
    if not os.path.exists(exe):
        _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=timeout)
    print(r.stdout[-4000:])
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert ", 0 failed" in r.stdout
    return r.stdout


def test_cpp_layer_builds_and_host_tests_pass(cmi):
    _build()
    out = _run("test_host")
    assert int(out.strip().splitlines()[-1].split()[0]) >= 50


def test_headers_compile_standalone(tmp_path):
    """Every public header compiles on its own (reference testing/trivial_tests)."""
    inc = os.path.join(ROOT, "cusp-autotuned_amd", "include")
    headers = []
    for d, _, files in os.walk(os.path.join(inc, "cusp")):
        headers += [os.path.relpath(os.path.join(d, f), inc) for f in files if f.endswith(".h")]
    assert len(headers) >= 20
    src = tmp_path / "one.cpp"
    for h in sorted(headers):
        src.write_text(f"#include <{h}>\nint main() {{ return 0; }}\n")
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", f"-I{inc}", str(src)], capture_output=True, text=True)
        assert r.returncode == 0, h + "\n" + r.stderr[-2000:]


@pytest.mark.gpu
def test_cpp_layer_device_tests_pass(cmi):
    out = _run("test_device")
    assert int(out.strip().splitlines()[-1].split()[0]) >= 50


@pytest.mark.gpu
def test_reference_style_benchmark_driver(cmi, tmp_path):
    """tools/spmv_bench (the reference's performance/spmv CLI on this engine): every format converts,
    multiplies on the device and matches the host multiply on the reference's default input."""
    exe = os.path.join(ROOT, "tools", "bin", "spmv_bench")
    if not os.path.exists(exe):
        _build()
    r = subprocess.run([exe, "--grid=300"], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "MISMATCH" not in r.stdout
    for fmt in ("coo", "csr", "dia", "ell", "hyb"):
        assert f"\t{fmt} :" in r.stdout, r.stdout
    # a MatrixMarket file goes through the reader (the SuiteSparse path of BASELINE.json configs[3])
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "5pt_10x10.mtx")], capture_output=True, text=True,
                       timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0 and "MISMATCH" not in r.stdout and "460 entries" in r.stdout, r.stdout + r.stderr
