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
    """The product's Python plumbing over the C-ABI (loads lib/libcusp_mi355x.so or raises)."""
    import cusp_autotuned_amd
    cusp_autotuned_amd.lib()
    return cusp_autotuned_amd


def read_mtx(path):
    """Tiny MatrixMarket coordinate reader for the fixture file (general, integer/real)."""
    with open(path) as f:
        lines = [l for l in f if not l.startswith("%")]
    rows, cols, nnz = map(int, lines[0].split())
    ent = np.array([l.split() for l in lines[1:1 + nnz]], dtype=np.float64)
    return rows, cols, ent[:, 0].astype(np.int64) - 1, ent[:, 1].astype(np.int64) - 1, ent[:, 2]


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
