#!/usr/bin/env python3
"""Generates the golden fixtures in tests/golden/.  Run in the BUILD container only (it needs
oracle/_ref/libcusp_ref.so, i.e. /root/reference):

    python tests/golden/make_golden.py

Expected outputs come from the REFERENCE's own sequential kernels
(cusp/system/detail/sequential/multiply/*_spmv.h compiled by oracle/Makefile into oracle/_ref/),
never from this repo's code.  Inputs are either literal data of the reference's tests (cited below)
or seeded synthetic matrices stored in full, so the fixtures are self-contained data: inputs and
expected outputs.  The fixtures then pin the C restatement (oracle/spmv_oracle.c) in the CPU tests
and the HIP kernels in the GPU tests.
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

orc = oracle.Oracle()
ref = oracle.Reference()


def dense_to_csr(D, dtype):
    D = np.asarray(D, dtype=np.float64)
    Ap, Aj, Ax = [0], [], []
    for i in range(D.shape[0]):
        for j in range(D.shape[1]):
            if D[i, j] != 0:
                Aj.append(j)
                Ax.append(D[i, j])
        Ap.append(len(Aj))
    return np.array(Ap, np.int32), np.array(Aj, np.int32), np.array(Ax, dtype)


# ---- literal matrices of the reference's tests -------------------------------------------------
# testing/multiply.cu:441-505 (TestSparseMatrixVectorMultiply): A(5x4) B(2x4) C(2x2) D(2x1) E(2x2) F(2x3)
TEST_MATRICES = {
    "A": [[13, 80, 0, 0], [0, 27, 0, 0], [55, 0, 24, 42], [0, 69, 0, 83], [0, 0, 27, 0]],
    "B": [[0, 2, 3, 4], [5, 0, 0, 8]],
    "C": [[0, 0], [3, 5]],
    "D": [[2], [3]],
    "E": [[0, 0], [0, 0]],
    "F": [[0, 1.5, 3.0], [0.5, 0, 0]],
}
# testing/poisson.cu:6-25 (TestPoisson5pt): poisson5pt(2,3) as a dense 6x6
POISSON_2x3 = [[4, -1, -1, 0, 0, 0], [-1, 4, 0, -1, 0, 0], [-1, 0, 4, -1, -1, 0],
               [0, -1, -1, 4, 0, -1], [0, 0, -1, 0, 4, -1], [0, 0, 0, -1, -1, 4]]
# testing/convert.cu:65-175 (initialize_conversion_example): the 4x4 matrix in every format
CONVERSION_EXAMPLE = {
    "csr": {"row_offsets": [0, 2, 3, 6, 7], "column_indices": [0, 1, 2, 0, 2, 3, 1],
            "values": [10.25, 11.00, 12.50, 13.75, 14.00, 15.25, 16.50]},
    "coo": {"row_indices": [0, 0, 1, 2, 2, 2, 3], "column_indices": [0, 1, 2, 0, 2, 3, 1],
            "values": [10.25, 11.00, 12.50, 13.75, 14.00, 15.25, 16.50]},
    "dia": {"alignment": 1, "diagonal_offsets": [-2, 0, 1],
            "values": [0, 0, 13.75, 16.50, 10.25, 0, 14.00, 0, 11.00, 12.50, 15.25, 0]},
    "ell": {"alignment": 1, "num_entries_per_row": 3,
            "column_indices": [0, 2, 0, 1, 1, -1, 2, -1, -1, -1, 3, -1],
            "values": [10.25, 12.50, 13.75, 16.50, 11.00, 0, 14.00, 0, 0, 0, 15.25, 0]},
    # testing/convert.cu:177-215: hyb.resize(4,4,4,3,1,1): ELL width 1 (+ 3 COO entries)
    "hyb": {"alignment": 1, "num_entries_per_row": 1,
            "ell_column_indices": [0, 2, 0, 1], "ell_values": [10.25, 12.50, 13.75, 16.50],
            "coo_row_indices": [0, 2, 2], "coo_column_indices": [1, 2, 3], "coo_values": [11.00, 14.00, 15.25]},
}


def known_answers():
    out = {"source": "reference testing/multiply.cu:383-512, testing/generalized_spmv.cu:20-70, "
                     "testing/poisson.cu:6-25, testing/convert.cu:65-215, testing/ell_matrix.cu:5-22",
           "spmv": [], "poisson_2x3_dense": POISSON_2x3, "conversion_example": CONVERSION_EXAMPLE,
           # testing/ell_matrix.cu:5-22: ell_matrix(3,2,6,2,alignment=4) has pitch 4
           "ell_pitch": {"num_rows": 3, "alignment": 4, "pitch": 4}}
    # protocol of CompareSparseMatrixVectorMultiply: x[i] = i % 10, y pre-filled with 10
    for name, D in TEST_MATRICES.items():
        D = np.array(D, np.float64)
        Ap, Aj, Ax = dense_to_csr(D, np.float64)
        x = (np.arange(D.shape[1]) % 10).astype(np.float64)
        y = ref.spmv_csr(D.shape[1], Ap, Aj, Ax, x)
        assert np.array_equal(y, D @ x), name  # small integers / halves: exact
        ys = ref.spmv_csr(D.shape[1], Ap, Aj, Ax, x, y0=np.full(D.shape[0], 10.0))  # scaled test :569-645
        out["spmv"].append({"name": name, "dense": D.tolist(), "x": x.tolist(), "y": y.tolist(),
                            "y_accumulate_from_10": ys.tolist()})
    # testing/generalized_spmv.cu:20-70: z = y + A x with x=[1,2,3,4], y=[10..50] -> [183,74,325,510,131]
    D = np.array(TEST_MATRICES["A"], np.float64)
    Ap, Aj, Ax = dense_to_csr(D, np.float64)
    z = ref.spmv_csr(4, Ap, Aj, Ax, np.array([1., 2., 3., 4.]), y0=np.array([10., 20., 30., 40., 50.]))
    assert z.tolist() == [183.0, 74.0, 325.0, 510.0, 131.0], z
    out["generalized_spmv"] = {"x": [1, 2, 3, 4], "y": [10, 20, 30, 40, 50], "z": z.tolist()}
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(out, f, indent=1)


def all_formats(rows, cols, Ap, Aj, Ax, x, hyb_width, tag, store):
    """Run the five reference kernels on one matrix; return y per format."""
    dtype = Ax.dtype
    ys = {}
    ys["csr"] = ref.spmv_csr(cols, Ap, Aj, Ax, x)
    Ai = orc.csr_row_indices(Ap)
    ys["coo"] = ref.spmv_coo(rows, cols, Ai, Aj, Ax, x)
    width = int(np.diff(Ap).max()) if rows else 0
    pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width)
    ys["ell"] = ref.spmv_ell(rows, cols, width, pitch, eAj, eAx, x)
    pitch_h, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, hyb_width)
    ys["hyb"] = ref.spmv_hyb(rows, cols, hyb_width, pitch_h, hAj, hAx, cAi, cAj, cAx, x)
    store[f"{tag}_hyb_width"] = np.int64(hyb_width)
    for k, v in ys.items():
        store[f"{tag}_y_{k}"] = v.astype(dtype)
    # y <- y + A x (initialize = identity) from a non-trivial y0
    y0 = (oracle.fill_x(rows + 7)[7:] * 3.0).astype(dtype)
    store[f"{tag}_y0"] = y0
    store[f"{tag}_yacc_csr"] = ref.spmv_csr(cols, Ap, Aj, Ax, x, y0=y0)
    store[f"{tag}_yacc_coo"] = ref.spmv_coo(rows, cols, Ai, Aj, Ax, x, y0=y0)
    store[f"{tag}_yacc_ell"] = ref.spmv_ell(rows, cols, width, pitch, eAj, eAx, x, y0=y0)
    store[f"{tag}_yacc_hyb"] = ref.spmv_hyb(rows, cols, hyb_width, pitch_h, hAj, hAx, cAi, cAj, cAx, x, y0=y0)
    return ys


def poisson_fixture():
    """config 1 of BASELINE.json: poisson5pt 100x100 CSR fp64 on host_memory; plus the other formats."""
    store = {}
    for dtype, tag in ((np.float64, "f64"), (np.float32, "f32")):
        m = n = 100
        N = m * n
        off, vals, nnz = orc.poisson5pt_dia(m, n, dtype)
        Ap, Aj, Ax = orc.dia_to_csr(N, N, off, vals, nnz)
        assert nnz == 5 * m * n - 2 * m - 2 * n == 49600
        x = oracle.fill_x(N, dtype)
        ys = all_formats(N, N, Ap, Aj, Ax, x, 3, tag, store)
        ys["dia"] = ref.spmv_dia(N, N, N, off, vals, x)
        store[f"{tag}_y_dia"] = ys["dia"]
        store[f"{tag}_yacc_dia"] = ref.spmv_dia(N, N, N, off, vals, x, y0=store[f"{tag}_y0"])
        # all five formats give bit-identical y on this matrix (every row is summed in
        # ascending-column order in each of them)
        for k in ("coo", "ell", "dia"):
            assert np.array_equal(ys["csr"], ys[k]), k
        store[f"{tag}_x"] = x
    assert store["f64_y_csr"][0] == -1.8074222668004014 and store["f64_y_csr"][5050] == -1.59679037111334
    store["m"], store["n"] = np.int64(100), np.int64(100)
    np.savez_compressed(os.path.join(HERE, "poisson_100x100.npz"), **store)


def irregular_fixture():
    """Seeded irregular matrices, stored in full: empty rows, rows longer than a wave (64), one row
    longer than an LDS tile (> 4096 entries), duplicate columns, rectangular shape."""
    rng = np.random.default_rng(20250215)
    store = {}
    for dtype, tag in ((np.float64, "f64"), (np.float32, "f32")):
        rows, cols = 1500, 1237
        lens = rng.integers(0, 12, size=rows)
        lens[rng.integers(0, rows, size=60)] = 0
        lens[rng.integers(0, rows, size=10)] = rng.integers(65, 400, size=10)
        lens[777] = 5000
        lens[0] = 0
        lens[rows - 1] = 0
        Ap = np.zeros(rows + 1, np.int32)
        Ap[1:] = np.cumsum(lens)
        nnz = int(Ap[-1])
        Aj = np.empty(nnz, np.int32)
        for i in range(rows):
            Aj[Ap[i]:Ap[i + 1]] = np.sort(rng.integers(0, cols, size=lens[i]))
        Ax = rng.standard_normal(nnz).astype(dtype)
        x = rng.standard_normal(cols).astype(dtype)
        store[f"{tag}_Ap"], store[f"{tag}_Aj"], store[f"{tag}_Ax"], store[f"{tag}_x"] = Ap, Aj, Ax, x
        hyb_width = orc.optimal_entries_per_row(Ap, 3.0, 4096)
        # with < 4096 long rows the heuristic picks a tiny width; also pin a mid split
        store[f"{tag}_heuristic_width"] = np.int64(hyb_width)
        all_formats(rows, cols, Ap, Aj, Ax, x, 6, tag, store)
    store["rows"], store["cols"] = np.int64(1500), np.int64(1237)
    np.savez_compressed(os.path.join(HERE, "irregular_1500x1237.npz"), **store)


def banded_dia_fixture():
    """A rectangular banded matrix for DIA incl. diagonals that start outside the square part."""
    rng = np.random.default_rng(7)
    store = {}
    rows, cols = 700, 900
    offsets = np.array([-699, -64, -3, -1, 0, 1, 2, 17, 255, 899], np.int32)
    pitch = 704
    for dtype, tag in ((np.float64, "f64"), (np.float32, "f32")):
        vals = rng.standard_normal(len(offsets) * pitch).astype(dtype)
        x = rng.standard_normal(cols).astype(dtype)
        y0 = rng.standard_normal(rows).astype(dtype)
        store[f"{tag}_vals"], store[f"{tag}_x"], store[f"{tag}_y0"] = vals, x, y0
        store[f"{tag}_y"] = ref.spmv_dia(rows, cols, pitch, offsets, vals, x)
        store[f"{tag}_yacc"] = ref.spmv_dia(rows, cols, pitch, offsets, vals, x, y0=y0)
    store["offsets"], store["rows"], store["cols"], store["pitch"] = offsets, np.int64(rows), np.int64(cols), np.int64(pitch)
    np.savez_compressed(os.path.join(HERE, "banded_700x900_dia.npz"), **store)


def data_files():
    """A data file the reference's own tests hold: the 5-point Laplacian on a 10x10 grid
    (testing/data/laplacian/5pt_10x10.mtx, 100x100, 460 entries) -- pins the generator."""
    src = "/root/reference/testing/data/laplacian/5pt_10x10.mtx"
    shutil.copyfile(src, os.path.join(HERE, "5pt_10x10.mtx"))


if __name__ == "__main__":
    known_answers()
    poisson_fixture()
    irregular_fixture()
    banded_dia_fixture()
    data_files()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
