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


# ---- the reference's own example programs, compiled unchanged against this layer (oracle/_ref/examples) ----
EXAMPLES = os.path.join(ROOT, "oracle", "_ref", "examples")
HOST_EXAMPLES = ["Algorithms_multiply", "MatrixFormats_coo", "MatrixFormats_csr", "MatrixFormats_dia", "MatrixFormats_ell", "MatrixFormats_hyb",
                 "Algorithms_transpose", "Algorithms_blas"]
DEVICE_EXAMPLES = ["Solvers_cg", "Gallery_poisson", "Monitors_monitor", "Monitors_verbose_monitor", "InputOutput_matrix_market",
                   "Solvers_bicgstab", "Solvers_cr", "Preconditioners_diagonal"]


def _examples_ready():
    if os.path.isdir("/root/reference/examples"):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "examples"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return all(os.path.exists(os.path.join(EXAMPLES, e)) for e in HOST_EXAMPLES + DEVICE_EXAMPLES)


def test_reference_examples_build_unchanged_and_host_ones_run(cmi):
    """examples/{Solvers/cg, MatrixFormats/*, Gallery/poisson, Monitors/*, InputOutput/matrix_market}.cu of the
    reference compile as they are (g++ -x c++) against cusp-autotuned_amd/include; the host_memory ones run here
    and print what the reference's cusp::print prints."""
    if not _examples_ready():
        pytest.skip("reference tree not present and no prebuilt oracle/_ref/examples")
    for e in HOST_EXAMPLES:
        r = subprocess.run([os.path.join(EXAMPLES, e)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, e + r.stderr[-500:]
        first = r.stdout.splitlines()[0]
        if e == "Algorithms_multiply":  # dense 2x2 times [10, 20]: y = [50, 140] (examples/Algorithms/multiply.cu)
            assert r.stdout.split() == ["array1d", "<2>", "(50)", "(140)"], r.stdout
            continue
        if e == "Algorithms_transpose":  # a 2 x 3 array2d and its transpose (examples/Algorithms/transpose.cu)
            assert r.stdout.split() == ["array2d", "<2,", "3>", "(10)", "(20)", "(30)", "(40)", "(50)", "(60)",
                                        "array2d", "<3,", "2>", "(10)", "(40)", "(20)", "(50)", "(30)", "(60)"], r.stdout
            continue
        if e == "Algorithms_blas":  # examples/Algorithms/blas.cu: axpy, xmy results and four norms of z
            assert "|z|_1 = 15" in r.stdout and "max(|z_i|) = 12" in r.stdout and r.stdout.count("array1d <2>") == 2, r.stdout
            continue
        assert first == ("sparse matrix <3, 4> with 8 entries" if e.endswith("hyb") else "sparse matrix <4, 3> with 6 entries"), (e, first)
    coo = subprocess.run([os.path.join(EXAMPLES, "MatrixFormats_coo")], capture_output=True, text=True).stdout.splitlines()
    assert coo[1].split() == ["0", "0", "(10)"] and coo[6].split() == ["3", "2", "(60)"]


@pytest.mark.gpu
def test_reference_examples_run_on_device(cmi, tmp_path):
    """The device_memory examples of the reference, unchanged, on the MI355X through the C-ABI: its CG example
    converges and prints the monitor's report."""
    if not all(os.path.exists(os.path.join(EXAMPLES, e)) for e in DEVICE_EXAMPLES):
        pytest.skip("no prebuilt oracle/_ref/examples (they are built where /root/reference exists)")
    outs = {}
    import shutil
    for e in DEVICE_EXAMPLES:
        if e == "Preconditioners_diagonal":  # reads ./A.mtx (InputOutput_matrix_market WRITES a file of that name: copy the fixture right before)
            shutil.copy(os.path.join(ROOT, "tests", "golden", "ref_data", "examples", "Preconditioners_A.mtx"), os.path.join(str(tmp_path), "A.mtx"))
        r = subprocess.run([os.path.join(EXAMPLES, e)], capture_output=True, text=True, timeout=120, cwd=tmp_path)
        assert r.returncode == 0, e + r.stdout[-500:] + r.stderr[-500:]
        outs[e] = r.stdout
    assert "Successfully converged after" in outs["Solvers_cg"]
    assert "Successfully converged after" in outs["Solvers_bicgstab"] and "Successfully converged after" in outs["Solvers_cr"]
    # examples/Preconditioners/diagonal.cu: the same 25 x 25 system without and with M = D^-1; both converge
    assert outs["Preconditioners_diagonal"].count("Successfully converged after") == 2, outs["Preconditioners_diagonal"][-600:]
    assert "sparse matrix <" in outs["Gallery_poisson"] and "sparse matrix <" in outs["InputOutput_matrix_market"]
    assert "onverged" in outs["Monitors_monitor"] or "residual" in outs["Monitors_monitor"].lower()


# ---- the one-process-per-GPU layer (cusp/distributed/*.h) through C++ only ---------------------------------------------------------
def _launch(ranks, mode, port, timeout=300, extra_env=None):
    for exe in ("tests/cpp/bin/test_distributed", "tools/bin/cmi_launch"):
        if not os.path.exists(os.path.join(ROOT, exe)):
            _build()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    r = subprocess.run([os.path.join(ROOT, "tools", "bin", "cmi_launch"), "-n", str(ranks), "--port", str(port), "--",
                        os.path.join(CPP, "bin", "test_distributed"), mode], capture_output=True, text=True, timeout=timeout, env=env)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("ranks", [1, 2, 3])
def test_sharded_operator_and_cg_host_transport(cmi, ranks):
    """world 1 / 2 / 3 on CPUs: row partitions (equal rows, balanced by entries), exchange plans (all-gather, unequal pieces, halo),
    cusp::multiply and cusp::krylov::cg on a sharded host_memory operator over the TCP star -- the logic RCCL carries on the GPUs.
    Every rank's rows have the single-process multiply's bits; CG takes the single-process iteration count."""
    out = _launch(ranks, "host", 29620 + ranks)
    lines = [l for l in out.splitlines() if l.startswith("ok ")]
    assert len(lines) == 12 and "sharded bicgstab" in out, out
    for fmt in ("ell", "coo", "dia", "hyb"):  # round 4: the sharded operator in the other formats (cusp/distributed/matrix.h)
        assert any(l.startswith(f"ok  sharded {fmt}") and "multiply bit-identical" in l for l in lines), (fmt, out)
    if ranks > 1:
        assert "banded/equal-rows/auto" in out and "mode halo" in out and "mode allgather" in out


@pytest.mark.gpu
def test_sharded_operator_and_cg_through_rccl_one_rank(cmi):
    """The same program on device_memory with a ONE-rank RCCL communicator (one GPU per box here): cmi_comm_create,
    cmi_allgather / cmi_allgatherv / cmi_halo_exchange / cmi_allreduce are the calls an 8-GPU node makes; the local SpMV and the
    fused CG steps are the single-GPU hot path."""
    out = _launch(1, "device", 29631)
    assert "RCCL version code" in out
    lines = [l for l in out.splitlines() if l.startswith("ok ")]
    assert len(lines) == 12 and "sharded bicgstab" in out, out
    for fmt in ("ell", "coo", "dia", "hyb"):
        assert any(l.startswith(f"ok  sharded {fmt}") for l in lines), (fmt, out)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_sharded_operator_ranks_share_one_gpu_one_sided_exchange(cmi, ranks):
    """2 / 3 ranks on the one GPU of the box (CMI_COMM_STAGED=1: RCCL refuses ranks sharing a device, so the collectives are staged
    through the host) with the REAL local kernels and the REAL one-sided exchange: IPC mappings between processes, cmi_copy_ranges
    pulls, and the fused CG whose pulls are ordered by its own reductions.  Every variant of the program incl. exchange_mode::peer."""
    out = _launch(ranks, "device", 29640 + ranks, extra_env={"CMI_COMM_STAGED": "1"})
    lines = [l for l in out.splitlines() if l.startswith("ok ")]
    assert len(lines) == 13 and "sharded bicgstab" in out, out
    for fmt in ("ell", "coo", "dia", "hyb"):  # (through the CSR operator's one-sided exchange)
        assert any(l.startswith(f"ok  sharded {fmt}") and "mode peer" in l for l in lines), (fmt, out)
    assert "mode peer" in out and "banded/by-entries/peer" in out


def test_host_layers_under_address_and_ub_sanitizers(cmi):
    """`make -C tests/cpp asan`: test_host and the sharded layer (3 ranks over the TCP star, host_memory) built with
    -fsanitize=address,undefined -- CPU build only, GPU sanitizers are not available on this pool."""
    _build()
    r = subprocess.run(["make", "-C", CPP, "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "runtime error" not in r.stdout + r.stderr and "AddressSanitizer" not in r.stdout + r.stderr
    assert r.stdout.count("ok  ") >= 7 and ", 0 failed" in r.stdout
