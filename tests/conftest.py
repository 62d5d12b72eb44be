import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The plain-C CPU oracle (test infrastructure; never the thing under test on the GPU side)."""
    import oracle
    return oracle.Oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference's own sequential kernels, when oracle/_ref was built (build container only)."""
    import oracle
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/libcusp_ref.so not built (needs /root/reference)")
    return oracle.Reference()


@pytest.fixture(scope="session")
def known():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_poisson():
    return np.load(os.path.join(GOLDEN, "poisson_100x100.npz"))


@pytest.fixture(scope="session")
def golden_irregular():
    return np.load(os.path.join(GOLDEN, "irregular_1500x1237.npz"))


@pytest.fixture(scope="session")
def golden_banded():
    return np.load(os.path.join(GOLDEN, "banded_700x900_dia.npz"))


@pytest.fixture(scope="session")
def cmi():
    """The product's Python plumbing over the C-ABI (loads lib/libcusp_mi355x.so or raises).  A checkout that has not
    been built yet is built first (hipcc cross-compiles gfx950 without a GPU; the oracle and the C++ test programs
    build themselves on demand): `python -m pytest tests -m "not gpu"` works on a fresh clone."""
    import cusp_autotuned_amd
    if not os.path.exists(cusp_autotuned_amd.lib_path()):
        cusp_autotuned_amd.build()
    cusp_autotuned_amd.lib()
    return cusp_autotuned_amd


def read_mtx(path):
    """Small MatrixMarket reader for the fixture files (test infrastructure): coordinate real / integer /
    pattern, general / symmetric, and dense `array` storage.  Returns rows, cols, I, J, V (0-based,
    symmetric entries mirrored, dense zeros dropped) in file order."""
    with open(path) as f:
        banner = f.readline().split()
        lines = [l for l in f if l.strip() and not l.startswith("%")]
    storage, kind, sym = banner[2], banner[3], banner[4]
    if storage == "array":
        rows, cols = map(int, lines[0].split())
        D = np.array([float(l) for l in lines[1:1 + rows * cols]]).reshape(cols, rows).T
        I, J = np.nonzero(D)
        return rows, cols, I.astype(np.int64), J.astype(np.int64), D[I, J]
    rows, cols, nnz = map(int, lines[0].split())
    if nnz == 0:
        return rows, cols, np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0)
    ent = np.array([l.split() for l in lines[1:1 + nnz]], dtype=np.float64)
    I, J = ent[:, 0].astype(np.int64) - 1, ent[:, 1].astype(np.int64) - 1
    V = ent[:, 2] if kind != "pattern" else np.ones(nnz)
    if sym == "symmetric":
        off = I != J
        I, J, V = np.concatenate([I, J[off]]), np.concatenate([J, I[off]]), np.concatenate([V, V[off]])
    return rows, cols, I, J, V


def coo_to_csr(rows, I, J, V, dtype=np.float64):
    """Sort by (row, column) and build CSR (test infrastructure)."""
    order = np.lexsort((J, I))
    I, J, V = I[order], J[order], V[order]
    Ap = np.zeros(rows + 1, np.int32)
    np.add.at(Ap, I + 1, 1)
    return np.cumsum(Ap).astype(np.int32), J.astype(np.int32), V.astype(dtype)


def reference_data_files():
    """The data files the reference's own tests hold (testing/data/{test,laplacian,random_10x10}/*.mtx)."""
    import glob
    return sorted(glob.glob(os.path.join(GOLDEN, "ref_data", "*", "*.mtx")) + [os.path.join(GOLDEN, "5pt_10x10.mtx")])


def dense_to_csr(D, dtype=np.float64):
    D = np.asarray(D, dtype=np.float64)
    Ap, Aj, Ax = [0], [], []
    for i in range(D.shape[0]):
        for j in range(D.shape[1]):
            if D[i, j] != 0:
                Aj.append(j)
                Ax.append(D[i, j])
        Ap.append(len(Aj))
    return np.array(Ap, np.int32), np.array(Aj, np.int32), np.array(Ax, dtype)
