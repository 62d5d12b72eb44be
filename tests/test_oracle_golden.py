"""CPU tests (-m "not gpu"): the oracle (plain-C restatement) against every golden vector and known
answer the reference's own tests hold for the SpMV path, and against the reference's own kernels
(oracle/_ref) where that library exists."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, coo_to_csr, dense_to_csr, read_mtx, reference_data_files

DT = {"f64": np.float64, "f32": np.float32}


# ---- reference testing/multiply.cu:383-512 + :569-645 ------------------------------------------
def test_known_answer_matrices_all_formats(orc, known):
    for case in known["spmv"]:
        D = np.array(case["dense"])
        rows, cols = D.shape
        x = np.array(case["x"])
        want = np.array(case["y"])
        want_acc = np.array(case["y_accumulate_from_10"])
        for dtype in (np.float64, np.float32):
            Ap, Aj, Ax = dense_to_csr(D, dtype)
            xd = x.astype(dtype)
            y10 = np.full(rows, 10.0, dtype)
            assert np.array_equal(orc.spmv_csr(Ap, Aj, Ax, xd), want), case["name"]
            assert np.array_equal(orc.spmv_csr(Ap, Aj, Ax, xd, omp=True), want)
            assert np.array_equal(orc.spmv_csr(Ap, Aj, Ax, xd, y0=y10), want_acc)
            Ai = orc.csr_row_indices(Ap)
            assert np.array_equal(orc.spmv_coo(rows, Ai, Aj, Ax, xd), want)
            width = int(np.diff(Ap).max())
            pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width)
            assert np.array_equal(orc.spmv_ell(rows, width, pitch, eAj, eAx, xd), want)
            if len(Ax):
                pitch_d, off, vals = orc.csr_to_dia(rows, cols, Ap, Aj, Ax)
                assert np.array_equal(orc.spmv_dia(rows, cols, pitch_d, off, vals, xd), want)
            for w in range(0, width + 1):
                p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, w)
                assert np.array_equal(orc.spmv_hyb(rows, w, p, hAj, hAx, cAi, cAj, cAx, xd), want)
                assert np.array_equal(orc.spmv_hyb(rows, w, p, hAj, hAx, cAi, cAj, cAx, xd, y0=y10), want_acc)


# ---- reference testing/generalized_spmv.cu:20-70 ------------------------------------------------
def test_generalized_spmv_known_answer(orc, known):
    g = known["generalized_spmv"]
    D = np.array([c for c in known["spmv"] if c["name"] == "A"][0]["dense"])
    Ap, Aj, Ax = dense_to_csr(D)
    z = orc.spmv_csr(Ap, Aj, Ax, np.array(g["x"], np.float64), y0=np.array(g["y"], np.float64))
    assert z.tolist() == [183.0, 74.0, 325.0, 510.0, 131.0] == g["z"]


# ---- reference testing/poisson.cu:6-25 and testing/data/laplacian/5pt_10x10.mtx -----------------
def test_poisson_generator_matches_reference_dense_2x3(orc, known):
    E = np.array(known["poisson_2x3_dense"], np.float64)
    off, vals, nnz = orc.poisson5pt_dia(2, 3)
    assert off.tolist() == [-2, -1, 0, 1, 2]
    Ap, Aj, Ax = orc.dia_to_csr(6, 6, off, vals, nnz)
    D = np.zeros((6, 6))
    for i in range(6):
        D[i, Aj[Ap[i]:Ap[i + 1]]] = Ax[Ap[i]:Ap[i + 1]]
    assert np.array_equal(D, E)
    assert nnz == int((E != 0).sum()) == 5 * 6 - 2 * 2 - 2 * 3


def test_poisson_generator_matches_reference_mtx_10x10(orc):
    rows, cols, I, J, V = read_mtx(os.path.join(GOLDEN, "5pt_10x10.mtx"))
    Ap, Aj, Ax = orc.poisson5pt_csr(10, 10)
    assert rows == cols == 100 and len(V) == 460 == Ap[-1]
    order = np.lexsort((J, I))
    assert np.array_equal(orc.csr_row_indices(Ap), I[order])
    assert np.array_equal(Aj, J[order])
    assert np.array_equal(Ax, V[order])
    # columns ascending within each row
    for i in range(100):
        assert np.all(np.diff(Aj[Ap[i]:Ap[i + 1]]) > 0)


# ---- reference testing/convert.cu:65-215,402-497 and testing/ell_matrix.cu:5-22 -----------------
def test_conversions_match_reference_4x4_example(orc, known):
    ex = known["conversion_example"]
    Ap = np.array(ex["csr"]["row_offsets"], np.int32)
    Aj = np.array(ex["csr"]["column_indices"], np.int32)
    Ax = np.array(ex["csr"]["values"], np.float32)  # the reference test is float
    assert orc.csr_row_indices(Ap).tolist() == ex["coo"]["row_indices"]
    pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, 3, alignment=1)
    assert pitch == 4 and eAj.tolist() == ex["ell"]["column_indices"] and eAx.tolist() == ex["ell"]["values"]
    pitch, off, vals = orc.csr_to_dia(4, 4, Ap, Aj, Ax, alignment=1)
    assert off.tolist() == ex["dia"]["diagonal_offsets"] and vals.tolist() == ex["dia"]["values"]
    p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, 1, alignment=1)
    assert hAj.tolist() == ex["hyb"]["ell_column_indices"] and hAx.tolist() == ex["hyb"]["ell_values"]
    assert cAi.tolist() == ex["hyb"]["coo_row_indices"] and cAj.tolist() == ex["hyb"]["coo_column_indices"]
    assert cAx.tolist() == ex["hyb"]["coo_values"]
    # DIA -> CSR brings the example back
    Ap2, Aj2, Ax2 = orc.dia_to_csr(4, 4, off, vals, 7)
    assert np.array_equal(Ap2, Ap) and np.array_equal(Aj2, Aj) and np.array_equal(Ax2, Ax)


def test_ell_pitch_alignment(orc, known):
    e = known["ell_pitch"]
    Ap = np.array([0, 2, 4, 6], np.int32)
    Aj = np.array([0, 1, 0, 1, 0, 1], np.int32)
    pitch, _, _ = orc.csr_to_ell(Ap, Aj, np.ones(6), 2, alignment=e["alignment"])
    assert pitch == e["pitch"]
    # default alignment 32 (cusp/detail/ell_matrix.inl:30-37): 9 998 244 rows -> 9 998 272
    assert 32 * ((9998244 + 31) // 32) == 9998272


def test_hyb_heuristic(orc):
    # for 5-pt Poisson K = 5 (HYB == ELL(5) + empty COO)
    Ap, _, _ = orc.poisson5pt_csr(100, 100)
    assert orc.optimal_entries_per_row(Ap) == 5
    # 3*(#rows longer than K) < num_rows picks the smallest such K once > 4096 rows are long
    lens = np.r_[np.full(20000, 2), np.full(9000, 10)]
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    assert orc.optimal_entries_per_row(Ap) == 2   # 3*9000 = 27000 < 29000
    lens = np.r_[np.full(10000, 2), np.full(9000, 10)]
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    assert orc.optimal_entries_per_row(Ap) == 10  # 27000 > 19000 and 9000 > 4096: never satisfied -> max
    lens = np.r_[np.full(10000, 2), np.full(4000, 10)]
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    assert orc.optimal_entries_per_row(Ap) == 2   # only 4000 (< 4096 breakeven) rows are longer than 2


# ---- golden vectors produced by the reference's own kernels (tests/golden/make_golden.py) --------
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_oracle_bit_exact_on_poisson_golden(orc, golden_poisson, tag):
    g, dtype = golden_poisson, DT[tag]
    m, n = int(g["m"]), int(g["n"])
    N = m * n
    x = g[f"{tag}_x"]
    assert np.array_equal(x, orc.fill_x(N, dtype)) and np.array_equal(x, oracle.fill_x(N, dtype))
    off, vals, nnz = orc.poisson5pt_dia(m, n, dtype)
    Ap, Aj, Ax = orc.dia_to_csr(N, N, off, vals, nnz)
    y0 = g[f"{tag}_y0"]
    assert np.array_equal(orc.spmv_csr(Ap, Aj, Ax, x), g[f"{tag}_y_csr"])
    assert np.array_equal(orc.spmv_csr(Ap, Aj, Ax, x, omp=True), g[f"{tag}_y_csr"])
    assert np.array_equal(orc.spmv_csr(Ap, Aj, Ax, x, y0=y0), g[f"{tag}_yacc_csr"])
    Ai = orc.csr_row_indices(Ap)
    assert np.array_equal(orc.spmv_coo(N, Ai, Aj, Ax, x), g[f"{tag}_y_coo"])
    assert np.array_equal(orc.spmv_coo(N, Ai, Aj, Ax, x, y0=y0), g[f"{tag}_yacc_coo"])
    pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, 5)
    assert pitch == 10016
    assert np.array_equal(orc.spmv_ell(N, 5, pitch, eAj, eAx, x), g[f"{tag}_y_ell"])
    assert np.array_equal(orc.spmv_ell(N, 5, pitch, eAj, eAx, x, y0=y0), g[f"{tag}_yacc_ell"])
    assert np.array_equal(orc.spmv_dia(N, N, N, off, vals, x), g[f"{tag}_y_dia"])
    assert np.array_equal(orc.spmv_dia(N, N, N, off, vals, x, y0=y0), g[f"{tag}_yacc_dia"])
    w = int(g[f"{tag}_hyb_width"])
    p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, w)
    assert np.array_equal(orc.spmv_hyb(N, w, p, hAj, hAx, cAi, cAj, cAx, x), g[f"{tag}_y_hyb"])
    assert np.array_equal(orc.spmv_hyb(N, w, p, hAj, hAx, cAi, cAj, cAx, x, y0=y0), g[f"{tag}_yacc_hyb"])
    if tag == "f64":  # the two sampled values recorded in SURVEY.md 8(c)
        assert g["f64_y_csr"][0] == -1.8074222668004014 and g["f64_y_csr"][5050] == -1.59679037111334


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_oracle_bit_exact_on_irregular_golden(orc, golden_irregular, tag):
    g = golden_irregular
    rows, cols = int(g["rows"]), int(g["cols"])
    Ap, Aj, Ax, x, y0 = (g[f"{tag}_{k}"] for k in ("Ap", "Aj", "Ax", "x", "y0"))
    assert np.array_equal(orc.spmv_csr(Ap, Aj, Ax, x), g[f"{tag}_y_csr"])
    assert np.array_equal(orc.spmv_csr(Ap, Aj, Ax, x, omp=True), g[f"{tag}_y_csr"])
    assert np.array_equal(orc.spmv_csr(Ap, Aj, Ax, x, y0=y0), g[f"{tag}_yacc_csr"])
    Ai = orc.csr_row_indices(Ap)
    assert np.array_equal(orc.spmv_coo(rows, Ai, Aj, Ax, x), g[f"{tag}_y_coo"])
    assert np.array_equal(orc.spmv_coo(rows, Ai, Aj, Ax, x, y0=y0), g[f"{tag}_yacc_coo"])
    width = int(np.diff(Ap).max())
    pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width)
    assert np.array_equal(orc.spmv_ell(rows, width, pitch, eAj, eAx, x), g[f"{tag}_y_ell"])
    assert np.array_equal(orc.spmv_ell(rows, width, pitch, eAj, eAx, x, y0=y0), g[f"{tag}_yacc_ell"])
    w = int(g[f"{tag}_hyb_width"])
    p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, w)
    assert np.array_equal(orc.spmv_hyb(rows, w, p, hAj, hAx, cAi, cAj, cAx, x), g[f"{tag}_y_hyb"])
    assert np.array_equal(orc.spmv_hyb(rows, w, p, hAj, hAx, cAi, cAj, cAx, x, y0=y0), g[f"{tag}_yacc_hyb"])
    assert orc.optimal_entries_per_row(Ap) == int(g[f"{tag}_heuristic_width"])


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_oracle_bit_exact_on_banded_dia_golden(orc, golden_banded, tag):
    g = golden_banded
    rows, cols, pitch = int(g["rows"]), int(g["cols"]), int(g["pitch"])
    off, vals, x, y0 = g["offsets"], g[f"{tag}_vals"], g[f"{tag}_x"], g[f"{tag}_y0"]
    assert np.array_equal(orc.spmv_dia(rows, cols, pitch, off, vals, x), g[f"{tag}_y"])
    assert np.array_equal(orc.spmv_dia(rows, cols, pitch, off, vals, x, y0=y0), g[f"{tag}_yacc"])


# ---- oracle vs the reference's own kernels, live (build container only) --------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_oracle_matches_live_reference_on_random_matrices(orc, ref, dtype):
    rng = np.random.default_rng(3)
    for trial in range(12):
        rows, cols = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        lens = rng.integers(0, min(cols, 9) + 1, size=rows)
        Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
        Aj = np.concatenate([np.sort(rng.choice(cols, size=l, replace=False)) for l in lens] + [np.empty(0, int)]).astype(np.int32)
        Ax = rng.standard_normal(len(Aj)).astype(dtype)
        x = rng.standard_normal(cols).astype(dtype)
        y0 = rng.standard_normal(rows).astype(dtype)
        for yy in (None, y0):
            assert np.array_equal(orc.spmv_csr(Ap, Aj, Ax, x, y0=yy), ref.spmv_csr(cols, Ap, Aj, Ax, x, y0=yy))
            Ai = orc.csr_row_indices(Ap)
            assert np.array_equal(orc.spmv_coo(rows, Ai, Aj, Ax, x, y0=yy), ref.spmv_coo(rows, cols, Ai, Aj, Ax, x, y0=yy))
            width = int(lens.max())
            pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width)
            assert np.array_equal(orc.spmv_ell(rows, width, pitch, eAj, eAx, x, y0=yy),
                                  ref.spmv_ell(rows, cols, width, pitch, eAj, eAx, x, y0=yy))
            if len(Ax):
                pd, off, vals = orc.csr_to_dia(rows, cols, Ap, Aj, Ax)
                assert np.array_equal(orc.spmv_dia(rows, cols, pd, off, vals, x, y0=yy),
                                      ref.spmv_dia(rows, cols, pd, off, vals, x, y0=yy))
            w = width // 2
            p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, w)
            assert np.array_equal(orc.spmv_hyb(rows, w, p, hAj, hAx, cAi, cAj, cAx, x, y0=yy),
                                  ref.spmv_hyb(rows, cols, w, p, hAj, hAx, cAi, cAj, cAx, x, y0=yy))


def test_empty_shapes(orc):
    # reference: empty matrices are legal (ell_spmv.h:117-121 fills y with init)
    Ap = np.zeros(4, np.int32)
    e = np.empty(0, np.int32)
    y = orc.spmv_csr(Ap, e, np.empty(0), np.ones(3))
    assert y.tolist() == [0, 0, 0]
    y = orc.spmv_coo(3, e, e, np.empty(0), np.ones(3), y0=np.array([1., 2., 3.]))
    assert y.tolist() == [1, 2, 3]


# ---- every data file the reference's tests hold: oracle SpMV (all formats) == dense product -------------
@pytest.mark.parametrize("path", reference_data_files(), ids=lambda p: os.path.basename(p))
def test_oracle_on_reference_data_files(orc, path):
    rows, cols, I, J, V = read_mtx(path)
    D = np.zeros((rows, cols))
    np.add.at(D, (I, J), V)
    Ap, Aj, Ax = coo_to_csr(rows, I, J, V)
    assert len(Aj) == len(I) and Ap[-1] == len(I)
    x = (np.arange(cols) % 7 - 3.0) * 0.5            # halves: with the files' small-integer entries every sum is exact
    exact = bool(np.all(V == np.round(V * 4) / 4))   # (coordinate_real_general holds 10.5, 250.5, 38.75: still exact)
    want = D @ x
    got = {"csr": orc.spmv_csr(Ap, Aj, Ax, x), "coo": orc.spmv_coo(rows, orc.csr_row_indices(Ap), Aj, Ax, x)}
    width = int(np.diff(Ap).max()) if len(Aj) else 0
    pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width)
    got["ell"] = orc.spmv_ell(rows, width, pitch, eAj, eAx, x)
    if len(Aj):
        pd, off, vals = orc.csr_to_dia(rows, cols, Ap, Aj, Ax)
        got["dia"] = orc.spmv_dia(rows, cols, pd, off, vals, x)
    hw = orc.optimal_entries_per_row(Ap)
    p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, hw)
    got["hyb"] = orc.spmv_hyb(rows, hw, p, hAj, hAx, cAi, cAj, cAx, x)
    for fmt, y in got.items():
        if exact:
            assert np.array_equal(y, want), fmt
        else:
            assert np.allclose(y, want, rtol=1e-13, atol=0), fmt
