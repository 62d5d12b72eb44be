"""bench.py's exchange-free checker (the 5-point stencil's closed form) against the oracle: it must
be bit-identical to the reference host loop on the gallery matrix, for whole matrices and row blocks."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("m,n", [(1, 1), (1, 7), (7, 1), (2, 3), (10, 10), (37, 23), (100, 100)])
def test_stencil_closed_form_is_the_oracle(orc, m, n):
    import torch
    import bench
    import cusp_autotuned_amd as cmi
    N = m * n
    Ap, Aj, Ax = orc.poisson5pt_csr(m, n)
    x = cmi.fill_x(N).numpy()
    want = orc.spmv_csr(Ap, Aj, Ax, x)
    got = bench.stencil_expected(torch, cmi, m, n, 0, N, "cpu").numpy()
    assert np.array_equal(got, want)
    # row blocks, as the ranks of a sharded run see them
    for world in (2, 3):
        cuts = [min(r * -(-N // world), N) for r in range(world + 1)]
        parts = [bench.stencil_expected(torch, cmi, m, n, a, b, "cpu").numpy() for a, b in zip(cuts, cuts[1:]) if b > a]
        assert np.array_equal(np.concatenate(parts) if parts else np.zeros(0), want)
    # scale (used by the max-size test to run a second, different x)
    got3 = bench.stencil_expected(torch, cmi, m, n, 0, N, "cpu", scale=3.0).numpy()
    assert np.array_equal(got3, orc.spmv_csr(Ap, Aj, Ax, x * 3.0))
