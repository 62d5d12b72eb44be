"""GPU parity tests (-m gpu): every SpMV entry point of the C-ABI, every kernel variant, against
(1) the golden vectors produced by the reference's own kernels, (2) the known answers of the
reference's tests and (3) the CPU oracle on seeded inputs -- plus size-independent properties at
BASELINE.json's full size (poisson5pt 3162x3162).

Bars (written here, as the task demands):
  * integer outputs (row offsets, column indices, row lengths, ...): bit-exact;
  * one-lane-per-row kernels (csr_scalar, csr_stream, ell, dia): BIT-EXACT vs the reference host
    loop -- same summation order, library built with -ffp-contract=off;
  * kernels that re-associate the row sum (csr_vector, coo, hyb's coo half):
    |y_gpu - y_ref| <= TOL * sum_j |a_ij x_j| with TOL = 1e-6 for f64 (north_star) and 1e-5 for f32.
"""
import itertools
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = {np.dtype(np.float64): 1e-6, np.dtype(np.float32): 1e-5}
DT = {"f64": np.float64, "f32": np.float32}


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    return torch


def dev(a, torch):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def row_abs(orc, Ap, Aj, Ax, x):
    return orc.spmv_csr(Ap, Aj, np.abs(Ax), np.abs(x))


def assert_close(y, want, bound, dtype, what=""):
    err = np.abs(y.astype(np.float64) - want.astype(np.float64))
    lim = TOL[np.dtype(dtype)] * np.maximum(bound.astype(np.float64), np.finfo(dtype).tiny)
    bad = np.nonzero(err > lim)[0]
    assert bad.size == 0, f"{what}: {bad.size} rows out of tolerance, first {bad[:5]}, err {err[bad[:5]]}"


LONG_ROW = 512  # csr_stream, threads_per_row = 0: shorter rows are summed in storage order (kLongRowMin, spmv_csr.hip)


def assert_variant(got, want, bound, dtype, exact, Ap, what):
    """exact True: bit-equal; "short": bit-equal on rows shorter than LONG_ROW, tolerance on the others; False: tolerance."""
    if exact is True:
        assert np.array_equal(got, want), f"{what}: not bit-exact"
        return
    assert_close(got, want, bound, dtype, what)
    if exact == "short":
        short = np.diff(Ap) < LONG_ROW
        assert np.array_equal(got[short], want[short]), f"{what}: a row shorter than {LONG_ROW} entries is not bit-exact"


def spmv_csr_planned(cmi, rows, cols, dAp, dAj, dAx, dx, y, accumulate=False, cfg=None):
    """y = A x through a plan made for (matrix, cfg): the row-length profile is what lets csr_stream launch its long-row
    instance or switch to the merge-path kernel (a plan-less call never measures anything)."""
    plan = cmi.Plan(cmi.FORMAT_CSR, y.dtype, rows, cols, dAj.numel(), dAp, cfg)
    cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y, accumulate=accumulate)
    return plan


def csr_variants(cmi, small=False):
    """Every CSR kernel variant worth distinguishing (launch shape x load policy)."""
    out = []
    for blk, nt in itertools.product((64, 256, 1024), (0, 1, 2, 3)):
        out.append(("scalar", True, cmi.Config(kernel=cmi.CSR_SCALAR, block_size=blk, nontemporal=nt)))
    for tpr in (2, 4, 8, 16, 32, 64):
        for blk in ((256,) if small else (128, 256, 512)):
            out.append((f"vector{tpr}", False, cmi.Config(kernel=cmi.CSR_VECTOR, block_size=blk, threads_per_row=tpr)))
    out.append(("vector8nt", False, cmi.Config(kernel=cmi.CSR_VECTOR, block_size=256, threads_per_row=8, nontemporal=1)))
    for blk, ipt, rpb, nt, swz in ((256, 1, 0, 0, 1), (256, 1, 0, 1, 0), (256, 2, 0, 0, 1), (256, 4, 0, 0, 0),
                                   (128, 1, 0, 0, 1), (512, 1, 0, 0, 1), (64, 1, 7, 2, 1), (256, 1, 1, 0, 0),
                                   (256, 1, 1024, 3, 1), (128, 2, 512, 1, 1), (1024, 1, 300, 0, 1), (256, 1, 192, 2, 0),
                                   (256, 1, 192, 2, 16), (128, 1, 5, 0, 3), (256, 2, 0, 2, 64),
                                   # policy bit 4: the entry streams requested lane-strided (one entry per lane per instruction)
                                   (256, 1, 0, 4, 1), (256, 1, 0, 7, 64), (256, 2, 0, 6, 0), (256, 4, 0, 7, 16), (128, 1, 5, 5, 3),
                                   (64, 1, 7, 6, 1), (512, 1, 400, 7, 32), (256, 1, 192, 7, 0), (1024, 2, 300, 4, 0)):
        # threads_per_row = 1: storage order for every row; 0 (the table's value): rows of LONG_ROW entries or more
        # are streamed by the whole workgroup (re-associated), every shorter row stays bit-exact
        out.append((f"stream b{blk} i{ipt} r{rpb} nt{nt} x{swz} strict", True,
                    cmi.Config(kernel=cmi.CSR_STREAM, block_size=blk, items_per_thread=ipt, rows_per_block=rpb,
                               nontemporal=nt, xcd_swizzle=swz, threads_per_row=1)))
        out.append((f"stream b{blk} i{ipt} r{rpb} nt{nt} x{swz}", "short",
                    cmi.Config(kernel=cmi.CSR_STREAM, block_size=blk, items_per_thread=ipt, rows_per_block=rpb,
                               nontemporal=nt, xcd_swizzle=swz)))
    for blk, ipt, rpb, tpr in ((256, 1, 0, 2), (256, 2, 0, 4), (256, 4, 64, 16), (128, 1, 8, 64), (512, 4, 0, 32), (64, 1, 3, 8)):
        out.append((f"stream-tpr b{blk} i{ipt} r{rpb} t{tpr}", False,
                    cmi.Config(kernel=cmi.CSR_STREAM, block_size=blk, items_per_thread=ipt, rows_per_block=rpb,
                               threads_per_row=tpr, nontemporal=2)))
    for blk, rpb, nt, chunked, bpc in ((256, 0, 0, 0, 0), (256, 0, 1, 1, 4), (128, 0, 0, 0, 8), (512, 0, 0, 1, 2),
                                       (64, 7, 2, 0, 1), (256, 255, 3, 0, 16), (256, 1, 1, 1, 3), (1024, 300, 0, 0, 1), (256, 192, 2, 0, 3)):
        out.append((f"pipe b{blk} r{rpb} nt{nt} c{chunked} bpc{bpc}", True,
                    cmi.Config(kernel=cmi.CSR_STREAM_PIPE, block_size=blk, rows_per_block=rpb, nontemporal=nt,
                               xcd_swizzle=chunked, blocks_per_cu=bpc)))
    # wave-private tiles (CMI_CSR_STREAM_WAVE): items_per_thread = entries per lane (64 x that many hold a wave's tile, else the
    # wave sums its rows one lane per row from the arrays); rows_per_block = waves x rows per wave; storage order either way
    for blk, k, rpb, nt, swz in ((256, 5, 0, 0, 0), (256, 2, 0, 3, 64), (64, 10, 64, 2, 1), (1024, 4, 16 * 7, 1, 16), (256, 8, 4, 0, 0),
                                 (128, 3, 128, 3, 32), (512, 0, 0, 2, 8)):
        out.append((f"wave b{blk} k{k} r{rpb} nt{nt} x{swz}", True,
                    cmi.Config(kernel=cmi.CSR_STREAM_WAVE, block_size=blk, items_per_thread=k, rows_per_block=rpb, nontemporal=nt, xcd_swizzle=swz)))
    for bpc in (0, 1, 3):  # merge-path split: re-associates (lane groups + atomics on rows that span tiles)
        out.append((f"balanced bpc{bpc}", False, cmi.Config(kernel=cmi.CSR_BALANCED, block_size=256, blocks_per_cu=bpc)))
    return out


def run_all_formats(cmi, torch, orc, rows, cols, Ap, Aj, Ax, x, want, want_acc, y0, hyb_width, label):
    """CSR (all variants), COO, ELL (+ELLR, 1/2 rows per lane), HYB against `want` / `want_acc`."""
    dtype = Ax.dtype
    bound = row_abs(orc, Ap, Aj, Ax, x) + (np.abs(y0) if y0 is not None else 0)
    dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)

    def fresh(acc):
        return dev(y0, torch).clone() if acc else torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")

    for acc, w in ((False, want["csr"]), (True, want_acc["csr"])):
        for name, exact, cfg in csr_variants(cmi, small=True):
            y = fresh(acc)
            cmi.spmv_csr(rows, cols, dAp, dAj, dAx, dx, y, accumulate=acc, cfg=cfg)
            assert_variant(host(y), w, bound, dtype, exact, Ap, f"{label} csr {name} acc={acc}")
        y = fresh(acc)  # NULL config: tuning table / heuristics
        cmi.spmv_csr(rows, cols, dAp, dAj, dAx, dx, y, accumulate=acc)
        assert_close(host(y), w, bound, dtype, f"{label} csr auto acc={acc}")
        if rows:
            y = fresh(acc)  # the same through a plan: profile-steered (long-row instance / merge-path where it pays)
            plan = spmv_csr_planned(cmi, rows, cols, dAp, dAj, dAx, dx, y, acc)
            assert_close(host(y), w, bound, dtype, f"{label} csr plan acc={acc}")
            if plan.info()["storage_order_sums"]:
                assert np.array_equal(host(y), w), f"{label} csr plan claims storage-order sums"

    # COO (sorted, as csr_to_coo produces)
    Ai = orc.csr_row_indices(Ap)
    dAi = dev(Ai, torch)
    for acc, w in ((False, want["coo"]), (True, want_acc["coo"])):
        for kern, ipt, blk in ((cmi.COO_SEGMENTED, 1, 64), (cmi.COO_SEGMENTED, 4, 256), (cmi.COO_SEGMENTED, 16, 256),
                               (cmi.COO_SEGMENTED, 3, 128), (cmi.COO_LANE4, 1, 64), (cmi.COO_LANE4, 4, 256),
                               (cmi.COO_LANE4, 2, 512), (cmi.COO_LANE4, 7, 128)):
            y = fresh(acc)
            cmi.spmv_coo(rows, cols, dAi, dAj, dAx, dx, y, accumulate=acc,
                         cfg=cmi.Config(kernel=kern, block_size=blk, items_per_thread=ipt))
            assert_close(host(y), w, bound, dtype, f"{label} coo k{kern} ipt{ipt} acc={acc}")
        y = fresh(acc)
        cmi.spmv_coo(rows, cols, dAi, dAj, dAx, dx, y, accumulate=acc)  # NULL config
        assert_close(host(y), w, bound, dtype, f"{label} coo auto acc={acc}")
        # row-sorted entries through a plan: the tile kernel, storage-order sums = the host loop's bits
        plan = cmi.Plan(cmi.FORMAT_COO, dx.dtype, rows, cols, len(Aj), dAi)
        assert plan.info()["coo_sorted"] is True
        coo_exact = plan.info()["storage_order_sums"]  # sorted entries: the plan's row offsets + a CSR kernel (or the COO tile kernel)
        for swz, nt in ((0, 0), (1, 2), (3, 3), (32, 1)):
            y = fresh(acc)
            cmi.spmv_coo(rows, cols, dAi, dAj, dAx, dx, y, accumulate=acc, cfg=cmi.Config(kernel=cmi.COO_TILE, xcd_swizzle=swz, nontemporal=nt))
            assert np.array_equal(host(y), w), f"{label} coo tile x{swz} nt{nt} acc={acc}: not bit-exact"
        y = fresh(acc)
        cmi.spmv_coo_plan(plan, dAi, dAj, dAx, dx, y, accumulate=acc)
        if coo_exact and len(Aj) >= 4:
            assert np.array_equal(host(y), w), f"{label} coo plan acc={acc}: not bit-exact"
        else:
            assert_close(host(y), w, bound, dtype, f"{label} coo plan acc={acc}")

    # ELL / ELLR
    width = int(np.diff(Ap).max()) if rows else 0
    pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width)
    deAj, deAx = dev(eAj, torch), dev(eAx, torch)
    rl = torch.empty(rows, dtype=torch.int32, device="cuda")
    cmi.ell_row_lengths(rows, width, pitch, deAj, rl)
    assert np.array_equal(host(rl), np.diff(Ap).astype(np.int32)), f"{label}: ELLR row lengths"
    for acc, w in ((False, want["ell"]), (True, want_acc["ell"])):
        for rpl, nt, ellr, blk in itertools.product((1, 2), (0, 1, 2, 3), (False, True), (256,)):
            swz = (0, 1, 3, 32)[(rpl + nt + ellr) % 4]  # tiles dealt to the XCDs: launch order, eighths, chunks of 3 / 32
            y = fresh(acc)
            cmi.spmv_ell(rows, cols, width, pitch, deAj, deAx, dx, y, row_lengths=rl if ellr else None, accumulate=acc,
                         cfg=cmi.Config(kernel=cmi.ELL_ROW, block_size=blk, threads_per_row=1, items_per_thread=rpl, nontemporal=nt, xcd_swizzle=swz))
            assert np.array_equal(host(y), w), f"{label} ell rpl{rpl} nt{nt} ellr{ellr} x{swz} acc={acc}: not bit-exact"
        for blk, swz in ((64, 1), (64, 2), (128, 5), (1024, 8)):  # small workgroups: many tiles, padded chunk rounds
            y = fresh(acc)
            cmi.spmv_ell(rows, cols, width, pitch, deAj, deAx, dx, y, accumulate=acc,
                         cfg=cmi.Config(kernel=cmi.ELL_ROW, block_size=blk, threads_per_row=1, items_per_thread=1, xcd_swizzle=swz))
            assert np.array_equal(host(y), w), f"{label} ell b{blk} x{swz} acc={acc}: not bit-exact"
        # S lanes per row (slices of the slots, partial sums through LDS): re-associated, 1e-6 class
        for lanes, blk, ellr, nt in ((2, 256, False, 0), (4, 256, True, 3), (8, 128, False, 2), (16, 256, True, 1), (16, 64, False, 0),
                                     (4, 1024, False, 0), (3, 256, False, 0)):
            y = fresh(acc)
            cmi.spmv_ell(rows, cols, width, pitch, deAj, deAx, dx, y, row_lengths=rl if ellr else None, accumulate=acc,
                         cfg=cmi.Config(kernel=cmi.ELL_ROW, block_size=blk, threads_per_row=lanes, nontemporal=nt))
            assert_close(host(y), w, bound, dtype, f"{label} ell lanes{lanes} b{blk} ellr{ellr} acc={acc}")
        y = fresh(acc)  # NULL config: the table's shape, lanes per row by the auto rule
        cmi.spmv_ell(rows, cols, width, pitch, deAj, deAx, dx, y, accumulate=acc)
        assert_close(host(y), w, bound, dtype, f"{label} ell auto acc={acc}")

    # HYB at the fixture's split
    p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, hyb_width)
    args = [dev(a, torch) for a in (hAj, hAx, cAi, cAj, cAx)]
    for acc, w in ((False, want["hyb"]), (True, want_acc["hyb"])):
        y = fresh(acc)
        cmi.spmv_hyb(rows, cols, hyb_width, p, *args, dx, y, accumulate=acc)
        assert_close(host(y), w, bound, dtype, f"{label} hyb acc={acc}")


# ------------------------------------------------------------------------------------------------
# golden vectors from the reference's own kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_poisson_100x100_golden_all_formats(cmi, torch_cuda, orc, golden_poisson, tag):
    torch, g, dtype = torch_cuda, golden_poisson, DT[tag]
    m, n = int(g["m"]), int(g["n"])
    N = m * n
    off, vals, nnz = orc.poisson5pt_dia(m, n, dtype)
    Ap, Aj, Ax = orc.dia_to_csr(N, N, off, vals, nnz)
    x, y0 = g[f"{tag}_x"], g[f"{tag}_y0"]
    want = {k: g[f"{tag}_y_{k}"] for k in ("csr", "coo", "ell", "hyb", "dia")}
    want_acc = {k: g[f"{tag}_yacc_{k}"] for k in ("csr", "coo", "ell", "hyb", "dia")}
    run_all_formats(cmi, torch, orc, N, N, Ap, Aj, Ax, x, want, want_acc, y0, int(g[f"{tag}_hyb_width"]), "poisson100")
    # DIA: gallery layout (pitch = N), 1 and 2 rows per lane, both load policies -- bit-exact
    doff, dvals, dx = dev(off, torch), dev(vals, torch), dev(x, torch)
    for acc, w in ((False, want["dia"]), (True, want_acc["dia"])):
        for rpl, nt, blk in itertools.product((1, 2), (0, 1, 2, 3), (64, 256)):
            y = dev(y0, torch).clone() if acc else torch.full((N,), 10.0, dtype=dx.dtype, device="cuda")
            cmi.spmv_dia(N, N, 5, N, doff, dvals, dx, y, accumulate=acc,
                         cfg=cmi.Config(kernel=cmi.DIA_ROW, block_size=blk, items_per_thread=rpl, nontemporal=nt))
            assert np.array_equal(host(y), w), f"dia rpl{rpl} nt{nt} acc={acc}: not bit-exact"


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_irregular_golden_all_formats(cmi, torch_cuda, orc, golden_irregular, tag):
    torch, g = torch_cuda, golden_irregular
    rows, cols = int(g["rows"]), int(g["cols"])
    Ap, Aj, Ax, x, y0 = (g[f"{tag}_{k}"] for k in ("Ap", "Aj", "Ax", "x", "y0"))
    want = {k: g[f"{tag}_y_{k}"] for k in ("csr", "coo", "ell", "hyb")}
    want_acc = {k: g[f"{tag}_yacc_{k}"] for k in ("csr", "coo", "ell", "hyb")}
    run_all_formats(cmi, torch, orc, rows, cols, Ap, Aj, Ax, x, want, want_acc, y0, int(g[f"{tag}_hyb_width"]), "irregular")


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_banded_dia_golden(cmi, torch_cuda, golden_banded, tag):
    torch, g = torch_cuda, golden_banded
    rows, cols, pitch = int(g["rows"]), int(g["cols"]), int(g["pitch"])
    off, vals, x, y0 = g["offsets"], g[f"{tag}_vals"], g[f"{tag}_x"], g[f"{tag}_y0"]
    doff, dvals, dx = dev(off, torch), dev(vals, torch), dev(x, torch)
    for acc, w in ((False, g[f"{tag}_y"]), (True, g[f"{tag}_yacc"])):
        for rpl, nt, (blk, swz) in itertools.product((1, 2), (0, 2, 3), ((256, 0), (64, 1), (64, 3), (128, 32), (1024, 2))):
            y = dev(y0, torch).clone() if acc else torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
            cmi.spmv_dia(rows, cols, len(off), pitch, doff, dvals, dx, y, accumulate=acc,
                         cfg=cmi.Config(kernel=cmi.DIA_ROW, block_size=blk, items_per_thread=rpl, nontemporal=nt, xcd_swizzle=swz))
            assert np.array_equal(host(y), w), f"banded dia rpl{rpl} nt{nt} b{blk} x{swz} acc={acc}"


def test_dia_more_than_256_diagonals(cmi, torch_cuda, orc):
    """The LDS offset chunk is 256 wide (as the reference's, dia_spmv.h:84-97): cross it.
    Mirrors the 1024-diagonal matrices of testing/ktt.cu:274-281 at a smaller size."""
    torch = torch_cuda
    rng = np.random.default_rng(11)
    rows, cols, nd = 1000, 777, 600
    off = np.sort(rng.choice(np.arange(-rows + 1, cols), size=nd, replace=False)).astype(np.int32)
    pitch = 1024
    vals = rng.standard_normal(nd * pitch)
    x = rng.standard_normal(cols)
    want = orc.spmv_dia(rows, cols, pitch, off, vals, x)
    for rpl, swz in itertools.product((1, 2), (0, 1, 4)):
        y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
        cmi.spmv_dia(rows, cols, nd, pitch, dev(off, torch), dev(vals, torch), dev(x, torch), y,
                     cfg=cmi.Config(kernel=cmi.DIA_ROW, block_size=64, items_per_thread=rpl, xcd_swizzle=swz))
        assert np.array_equal(host(y), want)


# ------------------------------------------------------------------------------------------------
# known answers of the reference's tests (testing/multiply.cu:383-512, generalized_spmv.cu:20-70)
# ------------------------------------------------------------------------------------------------
def test_reference_known_answer_matrices(cmi, torch_cuda, orc, known):
    from conftest import dense_to_csr
    torch = torch_cuda
    for case in known["spmv"]:
        D = np.array(case["dense"])
        rows, cols = D.shape
        for dtype in (np.float64, np.float32):
            Ap, Aj, Ax = dense_to_csr(D, dtype)
            x = np.array(case["x"], dtype)
            want = np.array(case["y"], dtype)
            want_acc = np.array(case["y_accumulate_from_10"], dtype)
            dx = dev(x, torch)
            dAp, dAj, dAx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch)
            for _, _, cfg in csr_variants(cmi, small=True) + [("auto", True, None)]:
                if cfg is not None and cfg.kernel == cmi.CSR_STREAM_PIPE and len(Ax) < 4:
                    with pytest.raises(cmi.CmiError):  # documented precondition, reported -- never a silent fallback
                        cmi.spmv_csr(rows, cols, dAp, dAj, dAx, dx, torch.zeros(rows, dtype=dx.dtype, device="cuda"), cfg=cfg)
                    continue
                y = torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
                cmi.spmv_csr(rows, cols, dAp, dAj, dAx, dx, y, cfg=cfg)
                assert np.array_equal(host(y), want), (case["name"], cfg)  # small integers: exact in any order
                y = torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
                cmi.spmv_csr(rows, cols, dAp, dAj, dAx, dx, y, accumulate=True, cfg=cfg)
                assert np.array_equal(host(y), want_acc), (case["name"], cfg)
            Ai = orc.csr_row_indices(Ap)
            y = torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
            cmi.spmv_coo(rows, cols, dev(Ai, torch), dAj, dAx, dx, y)
            assert np.array_equal(host(y), want), case["name"]
            width = int(np.diff(Ap).max())
            pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width)
            y = torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
            cmi.spmv_ell(rows, cols, width, pitch, dev(eAj, torch), dev(eAx, torch), dx, y)
            assert np.array_equal(host(y), want), case["name"]
            if len(Ax):
                pd, off, vals = orc.csr_to_dia(rows, cols, Ap, Aj, Ax)
                y = torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
                cmi.spmv_dia(rows, cols, len(off), pd, dev(off, torch), dev(vals, torch), dx, y)
                assert np.array_equal(host(y), want), case["name"]
            for w in range(width + 1):
                p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, w)
                y = torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
                cmi.spmv_hyb(rows, cols, w, p, *[dev(a, torch) for a in (hAj, hAx, cAi, cAj, cAx)], dx, y)
                assert np.array_equal(host(y), want), (case["name"], w)
    g = known["generalized_spmv"]
    D = np.array([c for c in known["spmv"] if c["name"] == "A"][0]["dense"])
    Ap, Aj, Ax = dense_to_csr(D)
    y = dev(np.array(g["y"], np.float64), torch)
    cmi.spmv_csr(5, 4, dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(np.array(g["x"], np.float64), torch), y,
                 accumulate=True)
    assert host(y).tolist() == [183.0, 74.0, 325.0, 510.0, 131.0]


# ------------------------------------------------------------------------------------------------
# the data files the reference's own tests hold (testing/data/{test,laplacian,random_10x10}/*.mtx)
# ------------------------------------------------------------------------------------------------
def test_reference_data_files_all_formats(cmi, torch_cuda, orc):
    """Every file, every format and kernel variant, f64 and f32, plain and accumulate -- against the oracle's
    result for the same arrays (000_nonzeros.mtx is the empty matrix, 100_nonzeros.mtx the full one)."""
    from conftest import coo_to_csr, read_mtx, reference_data_files
    torch = torch_cuda
    files = reference_data_files()
    assert len(files) == 20  # testing/data (19) + examples/Preconditioners/A.mtx
    for path in files:
        rows, cols, I, J, V = read_mtx(path)
        for dtype in (np.float64, np.float32):
            Ap, Aj, Ax = coo_to_csr(rows, I, J, V, dtype)
            x = ((np.arange(cols) % 7 - 3.0) * 0.5).astype(dtype)
            y0 = ((np.arange(rows) % 5) * 0.25 + 1.0).astype(dtype)
            hw = orc.optimal_entries_per_row(Ap)
            width = int(np.diff(Ap).max()) if len(Aj) else 0
            pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width)
            p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, hw)
            Ai = orc.csr_row_indices(Ap)
            want, want_acc = {}, {}
            for w, yy in ((want, None), (want_acc, y0)):
                w["csr"] = orc.spmv_csr(Ap, Aj, Ax, x, yy)
                w["coo"] = orc.spmv_coo(rows, Ai, Aj, Ax, x, yy)
                w["ell"] = orc.spmv_ell(rows, width, pitch, eAj, eAx, x, yy)
                w["hyb"] = orc.spmv_hyb(rows, hw, p, hAj, hAx, cAi, cAj, cAx, x, yy)
            if len(Aj) >= 4:  # csr_stream_pipe's documented precondition
                run_all_formats(cmi, torch, orc, rows, cols, Ap, Aj, Ax, x, want, want_acc, y0, hw, os.path.basename(path))
            else:             # 0, 1, 2 entries: the table-selected paths
                dx = dev(x, torch)
                y = torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
                cmi.spmv_csr(rows, cols, dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dx, y)
                assert np.array_equal(host(y), want["csr"]), path
                y = torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
                cmi.spmv_coo(rows, cols, dev(Ai, torch), dev(Aj, torch), dev(Ax, torch), dx, y)
                assert np.array_equal(host(y), want["coo"]), path
            if len(Aj):
                pd, off, vals = orc.csr_to_dia(rows, cols, Ap, Aj, Ax)
                y = torch.full((rows,), 10.0, dtype=dev(x, torch).dtype, device="cuda")
                cmi.spmv_dia(rows, cols, len(off), pd, dev(off, torch), dev(vals, torch), dev(x, torch), y)
                assert np.array_equal(host(y), orc.spmv_dia(rows, cols, pd, off, vals, x)), path


# ------------------------------------------------------------------------------------------------
# seeded random matrices: shapes from 1x1 to a few thousand, rectangular both ways, empty rows and
# columns, row lengths from 0 to dense -- every format and kernel variant against the oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(10))
def test_random_matrices_all_formats(cmi, torch_cuda, orc, seed):
    torch = torch_cuda
    rng = np.random.default_rng(1000 + seed)
    rows, cols = [(7, 129), (3001, 1000), (257, 64), (2049, 4097), (1000, 1), (64, 2500), (65, 64), (1, 1), (1000, 1000), (2, 4097)][seed]
    kind = seed % 5
    if kind == 0:
        lens = rng.integers(0, min(cols, 9) + 1, size=rows)                 # short rows, some empty
    elif kind == 1:
        lens = np.where(rng.random(rows) < 0.7, 0, rng.integers(1, min(cols, 40) + 1, size=rows))  # mostly empty
    elif kind == 2:
        lens = np.full(rows, min(cols, 33))                                  # uniform, ELL-friendly
    elif kind == 3:
        lens = np.minimum((rng.pareto(1.2, size=rows) * 3).astype(np.int64), cols)  # heavy tail
    else:
        lens = rng.integers(0, cols + 1, size=rows) if rows * cols <= 300_000 else rng.integers(0, 60, size=rows)  # up to dense
    lens = np.minimum(lens, cols).astype(np.int64)
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    Aj = np.concatenate([np.sort(rng.choice(cols, size=int(l), replace=False)) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    nnz = int(Ap[-1])
    for dtype in (np.float64, np.float32):
        Ax = rng.standard_normal(nnz).astype(dtype)
        x = rng.standard_normal(cols).astype(dtype)
        y0 = rng.standard_normal(rows).astype(dtype)
        width = int(lens.max()) if rows else 0
        hw = orc.optimal_entries_per_row(Ap)
        pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width)
        p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, hw)
        Ai = orc.csr_row_indices(Ap)
        want, want_acc = {}, {}
        for w, yy in ((want, None), (want_acc, y0)):
            w["csr"] = orc.spmv_csr(Ap, Aj, Ax, x, yy)
            w["coo"] = orc.spmv_coo(rows, Ai, Aj, Ax, x, yy)
            w["ell"] = orc.spmv_ell(rows, width, pitch, eAj, eAx, x, yy)
            w["hyb"] = orc.spmv_hyb(rows, hw, p, hAj, hAx, cAi, cAj, cAx, x, yy)
        if nnz >= 4:
            run_all_formats(cmi, torch, orc, rows, cols, Ap, Aj, Ax, x, want, want_acc, y0, hw, f"random seed {seed} {rows}x{cols} kind {kind}")
        if 0 < nnz and len(np.unique(Aj.astype(np.int64) - np.repeat(np.arange(rows), lens))) * rows <= 3_000_000:
            pd, off, vals = orc.csr_to_dia(rows, cols, Ap, Aj, Ax)
            for rpl in (1, 2):
                y = torch.full((rows,), 10.0, dtype=dev(x, torch).dtype, device="cuda")
                cmi.spmv_dia(rows, cols, len(off), pd, dev(off, torch), dev(vals, torch), dev(x, torch), y,
                             cfg=cmi.Config(kernel=cmi.DIA_ROW, items_per_thread=rpl))
                assert np.array_equal(host(y), orc.spmv_dia(rows, cols, pd, off, vals, x)), (seed, rpl)


# ------------------------------------------------------------------------------------------------
# FEM-like long rows (SuiteSparse surrogates of BASELINE.json configs[3]): 27-point stencil (~27/row,
# nlpkkt120-like) and 27-point x 3 dof (~79/row, ldoor-like), through the tuning table and explicit
# lane-group configurations
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dof", [1, 3])
def test_fem_like_long_rows(cmi, torch_cuda, orc, dof):
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1] / "tools"))
    import autotune
    torch = torch_cuda
    g = 24
    Ap, Aj, Ax = autotune.stencil_csr(g, g, g, autotune.stencil_points(27), np.float64)
    if dof > 1:
        Ap, Aj, Ax = autotune.block_expand(Ap, Aj, Ax, dof, np.float64)
    rows = len(Ap) - 1
    x = np.random.default_rng(4).standard_normal(rows)
    want = orc.spmv_csr(Ap, Aj, Ax, x)
    bound = row_abs(orc, Ap, Aj, Ax, x)
    dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
    sel = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, rows, rows, len(Ax))
    assert sel.kernel == cmi.CSR_STREAM
    cfgs = [None, cmi.Config(kernel=cmi.CSR_STREAM, items_per_thread=2), cmi.Config(kernel=cmi.CSR_STREAM, items_per_thread=4, block_size=512),
            cmi.Config(kernel=cmi.CSR_STREAM, items_per_thread=2, threads_per_row=4, block_size=256),
            cmi.Config(kernel=cmi.CSR_STREAM, items_per_thread=2, threads_per_row=8, block_size=512),
            cmi.Config(kernel=cmi.CSR_STREAM, items_per_thread=2, threads_per_row=32, block_size=512, rows_per_block=16),
            cmi.Config(kernel=cmi.CSR_VECTOR, threads_per_row=16), cmi.Config(kernel=cmi.CSR_SCALAR)]
    for cfg in cfgs:
        y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
        cmi.spmv_csr(rows, rows, dAp, dAj, dAx, dx, y, cfg=cfg)
        got = host(y)
        exact = cfg is not None and cfg.kernel in (cmi.CSR_STREAM, cmi.CSR_SCALAR) and cfg.threads_per_row <= 1
        if exact or (cfg is None and sel.threads_per_row <= 1):
            assert np.array_equal(got, want), f"dof{dof} {cfg}"
        else:
            assert_close(got, want, bound, np.float64, f"dof{dof} {cfg}")


# ------------------------------------------------------------------------------------------------
# edge cases
# ------------------------------------------------------------------------------------------------
def test_empty_and_degenerate_shapes(cmi, torch_cuda):
    torch = torch_cuda
    i32, f64 = torch.int32, torch.float64
    # rows but no entries: y = 0 (or untouched when accumulating)
    Ap = torch.zeros(6, dtype=i32, device="cuda")
    e_i, e_v = torch.empty(0, dtype=i32, device="cuda"), torch.empty(0, dtype=f64, device="cuda")
    x = torch.ones(3, dtype=f64, device="cuda")
    for cfg in (None, cmi.Config(kernel=cmi.CSR_SCALAR), cmi.Config(kernel=cmi.CSR_VECTOR, threads_per_row=4),
                cmi.Config(kernel=cmi.CSR_STREAM)):
        y = torch.full((5,), 10.0, dtype=f64, device="cuda")
        cmi.spmv_csr(5, 3, Ap, e_i, e_v, x, y, cfg=cfg)
        assert host(y).tolist() == [0] * 5
        y = torch.full((5,), 10.0, dtype=f64, device="cuda")
        cmi.spmv_csr(5, 3, Ap, e_i, e_v, x, y, accumulate=True, cfg=cfg)
        assert host(y).tolist() == [10] * 5
    y = torch.full((5,), 10.0, dtype=f64, device="cuda")
    cmi.spmv_coo(5, 3, e_i, e_i, e_v, x, y)
    assert host(y).tolist() == [0] * 5
    y = torch.full((5,), 10.0, dtype=f64, device="cuda")
    cmi.spmv_ell(5, 3, 0, 32, e_i, e_v, x, y)  # reference: empty ELL -> y = init (ell_spmv.h:117-121)
    assert host(y).tolist() == [0] * 5
    y = torch.full((5,), 10.0, dtype=f64, device="cuda")
    cmi.spmv_dia(5, 3, 0, 32, e_i, e_v, x, y)
    assert host(y).tolist() == [0] * 5
    # zero rows: nothing happens, no error
    cmi.spmv_csr(0, 3, torch.zeros(1, dtype=i32, device="cuda"), e_i, e_v, x, e_v)
    # a 1x1 matrix
    y = torch.zeros(1, dtype=f64, device="cuda")
    one_i = torch.zeros(1, dtype=i32, device="cuda")
    cmi.spmv_csr(1, 1, torch.tensor([0, 1], dtype=i32, device="cuda"), one_i,
                 torch.tensor([2.5], dtype=f64, device="cuda"), torch.tensor([4.0], dtype=f64, device="cuda"), y)
    assert host(y).tolist() == [10.0]


@pytest.mark.parametrize("shape", ["one_dense_row", "dense_last_row", "mostly_empty", "unsorted_duplicates"])
def test_pathological_row_length_distributions(cmi, torch_cuda, orc, shape):
    """Extreme skew: every CSR kernel (multi-pass tiles, lane groups, persistent variant) and the
    COO kernels must still match the host order / tolerance."""
    torch = torch_cuda
    rng = np.random.default_rng(17)
    rows, cols = 20000, 30000
    lens = np.zeros(rows, np.int64)
    if shape == "one_dense_row":
        lens[:] = rng.integers(0, 4, size=rows)
        lens[1234] = 250_000
    elif shape == "dense_last_row":
        lens[-1] = 70_001
        lens[0] = 3
    elif shape == "mostly_empty":
        lens[rng.integers(0, rows, size=300)] = rng.integers(1, 50, size=300)
    else:
        lens[:] = rng.integers(0, 9, size=rows)
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    nnz = int(Ap[-1])
    Aj = rng.integers(0, cols, size=nnz).astype(np.int32)  # unsorted, duplicates allowed
    if shape != "unsorted_duplicates":
        for i in np.nonzero(lens > 1)[0][:50]:
            Aj[Ap[i]:Ap[i + 1]] = np.sort(Aj[Ap[i]:Ap[i + 1]])
    Ax = rng.standard_normal(nnz)
    if shape == "unsorted_duplicates":
        Ax[rng.integers(0, nnz, size=nnz // 10)] = 0.0  # explicit zeros stay entries
    x = rng.standard_normal(cols)
    want = orc.spmv_csr(Ap, Aj, Ax, x)
    bound = row_abs(orc, Ap, Aj, Ax, x)
    dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
    for name, exact, cfg in csr_variants(cmi, small=True) + [("auto", False, None)]:
        y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
        cmi.spmv_csr(rows, cols, dAp, dAj, dAx, dx, y, cfg=cfg)
        assert_variant(host(y), want, bound, np.float64, exact, Ap, f"{shape} {name}")
    Ai = orc.csr_row_indices(Ap)
    for kern in (cmi.COO_SEGMENTED, cmi.COO_LANE4):
        y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
        cmi.spmv_coo(rows, cols, dev(Ai, torch), dAj, dAx, dx, y, cfg=cmi.Config(kernel=kern, items_per_thread=2))
        assert_close(host(y), want, bound, np.float64, f"{shape} coo{kern}")


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_long_rows_streamed_by_the_workgroup(cmi, torch_cuda, orc, tag):
    """csr_stream's cooperative long-row path at its edges: rows of exactly 511 / 512 / 513 entries (the threshold), long
    rows first and last in the matrix and in a tile, two long rows side by side, long rows whose first entry sits at every
    offset modulo 4 (head / tail handling of the 16-byte vector loop), unaligned arrays, y += A x, the fused dot, lane
    groups (threshold 128 entries per lane of the group) -- shorter rows must stay bit-exact next to them."""
    torch = torch_cuda
    dt = np.float64 if tag == "f64" else np.float32
    rng = np.random.default_rng(77)
    rows, cols = 3000, 5000
    lens = rng.integers(0, 9, size=rows)
    special = {0: 700, 1: 512, 2: 511, 3: 513, 175: 2048, 176: 1500, 177: 3, 178: 4097, 1000: 515, 1001: 514, 1002: 513,
               1003: 512, 2047: 9000, 2999: 1234}
    for r, l in special.items():
        lens[r] = l
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    nnz = int(Ap[-1])
    Aj = rng.integers(0, cols, size=nnz).astype(np.int32)
    Ax = rng.standard_normal(nnz).astype(dt)
    x = rng.standard_normal(cols).astype(dt)
    y0 = rng.standard_normal(rows).astype(dt)
    want, want_acc = orc.spmv_csr(Ap, Aj, Ax, x), orc.spmv_csr(Ap, Aj, Ax, x, y0)
    bound = row_abs(orc, Ap, Aj, Ax, x)
    dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
    tdt = torch.float64 if tag == "f64" else torch.float32
    shapes = [(256, 1, 0), (256, 1, 176), (128, 2, 64), (512, 4, 0), (64, 1, 7), (1024, 1, 300), (256, 2, 1)]
    for blk, ipt, rpb in shapes:
        for acc in (False, True):
            cfg = cmi.Config(kernel=cmi.CSR_STREAM, block_size=blk, items_per_thread=ipt, rows_per_block=rpb, nontemporal=2)
            y = dev(y0, torch).clone() if acc else torch.full((rows,), 10.0, dtype=tdt, device="cuda")
            plan = spmv_csr_planned(cmi, rows, cols, dAp, dAj, dAx, dx, y, acc, cfg)
            assert plan.info()["max_row_length"] == 9000 and not plan.info()["storage_order_sums"]
            assert_variant(host(y), want_acc if acc else want, bound + (np.abs(y0) if acc else 0), dt, "short", Ap,
                           f"long rows b{blk} i{ipt} r{rpb} acc={acc}")
            # without a plan nothing is known about the rows: the ordinary instance, storage order for every row
            y = dev(y0, torch).clone() if acc else torch.full((rows,), 10.0, dtype=tdt, device="cuda")
            cmi.spmv_csr(rows, cols, dAp, dAj, dAx, dx, y, accumulate=acc, cfg=cfg)
            assert np.array_equal(host(y), want_acc if acc else want), f"long rows, no plan b{blk} i{ipt} r{rpb} acc={acc}"
    # lane groups: the threshold moves to 128 entries per lane (tpr 8 -> 1024): rows of 1500+ are streamed, 700 is not
    for tpr, rpb in ((8, 64), (32, 16), (2, 0)):
        y = torch.full((rows,), 10.0, dtype=tdt, device="cuda")
        spmv_csr_planned(cmi, rows, cols, dAp, dAj, dAx, dx, y, False,
                         cmi.Config(kernel=cmi.CSR_STREAM, block_size=256, items_per_thread=2, rows_per_block=rpb, threads_per_row=tpr))
        assert_close(host(y), want, bound, dt, f"long rows tpr{tpr}")
    # unaligned column / value arrays: the scalar-load instances have the same path
    bj = torch.zeros(nnz + 8, dtype=torch.int32, device="cuda")
    bv = torch.zeros(nnz + 8, dtype=tdt, device="cuda")
    bj[1:1 + nnz].copy_(dAj)
    bv[3:3 + nnz].copy_(dAx)
    y = torch.full((rows,), 10.0, dtype=tdt, device="cuda")
    pu = cmi.Plan(cmi.FORMAT_CSR, tdt, rows, cols, nnz, dAp, cmi.Config(kernel=cmi.CSR_STREAM))
    cmi.spmv_csr_plan(pu, dAp, bj[1:1 + nnz], bv[3:3 + nnz], dx, y)
    assert_variant(host(y), want, bound, dt, "short", Ap, "long rows, unaligned arrays")
    # storage order on request: every row bit-exact, and the plan says so
    y = torch.full((rows,), 10.0, dtype=tdt, device="cuda")
    ps = spmv_csr_planned(cmi, rows, cols, dAp, dAj, dAx, dx, y, False, cmi.Config(kernel=cmi.CSR_STREAM, threads_per_row=1))
    assert np.array_equal(host(y), want) and ps.info()["storage_order_sums"]
    # the streamed rows are re-associated but deterministic (fixed fold order: lanes, then waves): same bits every time
    ya = torch.full((rows,), 10.0, dtype=tdt, device="cuda")
    pa = spmv_csr_planned(cmi, rows, cols, dAp, dAj, dAx, dx, ya)
    for _ in range(20):
        yb = torch.full((rows,), -1.0, dtype=tdt, device="cuda")
        cmi.spmv_csr_plan(pa, dAp, dAj, dAx, dx, yb)
        assert torch.equal(ya, yb)
    # the fused dot runs the same instances (f64 and f32): y identical to the plain call, dot within rounding
    wv = dev(rng.standard_normal(rows).astype(dt), torch)
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    y1 = torch.full((rows,), 10.0, dtype=tdt, device="cuda")
    y2 = torch.full((rows,), -3.0, dtype=tdt, device="cuda")
    cfg = cmi.Config(kernel=cmi.CSR_STREAM, block_size=256, rows_per_block=176)
    pd = spmv_csr_planned(cmi, rows, cols, dAp, dAj, dAx, dx, y1, False, cfg)
    cmi.spmv_csr_dot(rows, cols, dAp, dAj, dAx, dx, y2, wv, res, cmi.blas_workspace(), plan=pd)
    assert torch.equal(y1, y2)
    ref = float((y2.double() * wv.double()).sum())
    assert abs(float(res) - ref) <= 1e-11 * float((y2.double() * wv.double()).abs().sum())


def test_row_length_profile_selects_the_balanced_kernel(cmi, torch_cuda, orc):
    """With no explicit kernel the library measures the longest row once per matrix and leaves the row-tile
    kernel for the merge-path one when that row would dominate: results stay within tolerance, and the
    multiply no longer takes milliseconds per long row."""
    import time
    torch = torch_cuda
    rng = np.random.default_rng(23)
    rows = cols = 300_000
    lens = rng.integers(2, 9, size=rows)
    lens[[5, 77_777, 299_999]] = 400_000
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    nnz = int(Ap[-1])
    Aj = rng.integers(0, cols, size=nnz).astype(np.int32)
    Ax = rng.standard_normal(nnz)
    x = rng.standard_normal(cols)
    dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
    L = cmi.lib()
    import ctypes
    got = ctypes.c_int64()
    cmi.check(L.cmi_csr_max_row_length(rows, ctypes.c_void_p(dAp.data_ptr()), ctypes.byref(got), None))
    assert got.value == 400_000
    want = orc.spmv_csr(Ap, Aj, Ax, x)
    bound = row_abs(orc, Ap, Aj, Ax, x)
    y = torch.empty(rows, dtype=torch.float64, device="cuda")

    def timed(cfg):
        plan = cmi.Plan(cmi.FORMAT_CSR, torch.float64, rows, cols, nnz, dAp, cfg)  # measures the rows: the one synchronisation
        assert plan.info()["max_row_length"] == 400_000 and plan.info()["entries_in_long_rows"] == 1_200_000
        cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y)  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, plan

    t_auto, p_auto = timed(None)
    assert p_auto.config().kernel == cmi.CSR_BALANCED and not p_auto.info()["storage_order_sums"]
    assert_close(host(y), want, bound, np.float64, "auto (profile)")
    table = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, rows, cols, nnz)   # what the mean row length alone selects
    assert table.kernel == cmi.CSR_STREAM and table.threads_per_row == 0
    # the row-tile kernel as the table shapes it: the three long rows are streamed by their whole workgroup
    # (re-associated), every other row is bit-exact
    t_table, p_table = timed(table)
    assert p_table.config().kernel == cmi.CSR_STREAM  # an explicit kernel is kept; the profile picks its long-row instance
    assert_variant(host(y), want, bound, np.float64, "short", Ap, "table config")
    # threads_per_row = 1: storage order for EVERY row -- bit-exact, but one lane sums each long row
    table.threads_per_row = 1
    t_strict, p_strict = timed(table)
    assert np.array_equal(host(y), want) and p_strict.info()["storage_order_sums"]
    # no plan, no config: nothing is measured inside a multiply -- the table's row-tile kernel, storage order everywhere
    cmi.spmv_csr(rows, cols, dAp, dAj, dAx, dx, y)
    assert np.array_equal(host(y), want)
    assert t_auto * 5 < t_strict and t_table * 3 < t_strict, (t_auto, t_table, t_strict)
    # accumulate mode through the balanced kernel: no zero fill, y += A x
    y0 = rng.standard_normal(rows)
    dy = dev(y0, torch)
    cmi.spmv_csr(rows, cols, dAp, dAj, dAx, dx, dy, accumulate=True, cfg=cmi.Config(kernel=cmi.CSR_BALANCED))
    assert_close(host(dy), orc.spmv_csr(Ap, Aj, Ax, x, y0), bound + np.abs(y0), np.float64, "balanced accumulate")


def test_multiply_inside_hip_graph_capture(cmi, torch_cuda, orc):
    """cmi_spmv_* is capturable: launches only, nothing allocates or synchronises.  A matrix that has no plan yet gets
    none inside a capture (making one needs a read-back): the plan-less entry point records the table's kernel; a
    matrix planned before the capture records its plan's kernel.  Replays reproduce the eager result in every format."""
    torch = torch_cuda
    Ap, Aj, Ax = orc.poisson5pt_csr(83, 61)
    n = 83 * 61
    x = np.random.default_rng(5).standard_normal(n)
    want = orc.spmv_csr(Ap, Aj, Ax, x)
    A = cmi.CsrMatrix(n, n, len(Aj), dev(Ap, torch), dev(Aj, torch), dev(Ax, torch))  # fresh arrays: no cached profile
    mats = {"csr": A, "ell": cmi.convert(A, "ell"), "coo": cmi.convert(A, "coo"), "hyb": cmi.convert(A, "hyb", num_entries_per_row=3),
            "dia": cmi.convert(A, "dia")}
    dx = dev(x, torch)
    side = torch.cuda.Stream()
    mats["coo planned"] = cmi.convert(A, "coo")
    mats["coo planned"].plan()
    tile_planned = mats["coo planned"].plan().info()["storage_order_sums"]  # (through its plan a sorted COO multiply is exact)
    for fmt, M in mats.items():
        y = torch.full((n,), 7.0, dtype=torch.float64, device="cuda")
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            cmi.multiply(M, dx, y)
        y.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        got = host(y)
        if fmt in ("csr", "ell", "dia") or (fmt == "coo planned" and tile_planned):
            assert np.array_equal(got, want), fmt
            if fmt in ("csr", "coo"):
                assert getattr(M, "_plan", None) is None, "a capture must not have made a plan"
        else:
            assert np.allclose(got, want, rtol=1e-12, atol=1e-12), fmt
        dx2 = dx * 2.0            # same graph, new input values in the same buffers
        dx.copy_(dx2)
        g.replay()
        torch.cuda.synchronize()
        assert np.allclose(host(y), 2.0 * want, rtol=1e-12, atol=1e-12), fmt
        dx.copy_(dev(x, torch))


def test_bad_config_is_an_error_not_a_fallback(cmi, torch_cuda):
    torch = torch_cuda
    Ap = torch.tensor([0, 1], dtype=torch.int32, device="cuda")
    Aj = torch.zeros(1, dtype=torch.int32, device="cuda")
    v = torch.ones(1, dtype=torch.float64, device="cuda")
    for cfg in (cmi.Config(kernel=cmi.ELL_ROW), cmi.Config(kernel=77), cmi.Config(kernel=cmi.DIA_ROW)):
        with pytest.raises(cmi.CmiError) as e:
            cmi.spmv_csr(1, 1, Ap, Aj, v, v, v.clone(), cfg=cfg)
        assert e.value.status == 3  # CMI_ERROR_NOT_SUPPORTED
    with pytest.raises(cmi.CmiError):
        cmi.spmv_csr(1, 1, Ap, Aj, v, v, v.clone(), cfg=cmi.Config(kernel=cmi.CSR_STREAM, block_size=64, rows_per_block=9999))


def test_unaligned_views_take_the_scalar_load_paths(cmi, torch_cuda, orc, golden_irregular):
    """Aj/Ax that are not 16-byte aligned (e.g. slices of larger buffers) must still be exact."""
    torch, g = torch_cuda, golden_irregular
    rows, cols = int(g["rows"]), int(g["cols"])
    Ap, Aj, Ax, x = (g[f"f64_{k}"] for k in ("Ap", "Aj", "Ax", "x"))
    want = g["f64_y_csr"]
    for shift_i, shift_v in ((1, 0), (0, 1), (3, 1), (2, 0)):
        bufj = torch.zeros(len(Aj) + 8, dtype=torch.int32, device="cuda")
        bufv = torch.zeros(len(Ax) + 8, dtype=torch.float64, device="cuda")
        dAj = bufj[shift_i:shift_i + len(Aj)]
        dAx = bufv[shift_v:shift_v + len(Ax)]
        dAj.copy_(dev(Aj, torch))
        dAx.copy_(dev(Ax, torch))
        bound = row_abs(orc, Ap, Aj, Ax, x)
        # (nontemporal 4: the lane-strided request shape, which assumes no alignment at all)
        for ipt, tpr, pol in itertools.product((1, 2, 4), (1, 0), (0, 4)):  # 1: storage order everywhere; 0: the 5000-entry row is streamed
            y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
            cmi.spmv_csr(rows, cols, dev(Ap, torch), dAj, dAx, dev(x, torch), y,
                         cfg=cmi.Config(kernel=cmi.CSR_STREAM, items_per_thread=ipt, threads_per_row=tpr, nontemporal=pol))
            assert_variant(host(y), want, bound, np.float64, True if tpr == 1 else "short", Ap, f"{shift_i} {shift_v} {ipt} {tpr} {pol}")
    # ELL with an odd pitch: the two-rows-per-lane request silently uses one row per lane (same result)
    width = int(np.diff(Ap).max())
    pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width, alignment=1)
    if pitch % 2 == 0:
        pitch, eAj, eAx = orc.csr_to_ell(np.r_[Ap, Ap[-1]].astype(np.int32), Aj, Ax, width, alignment=1)
    y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
    cmi.spmv_ell(rows, cols, width, pitch, dev(eAj, torch), dev(eAx, torch), dev(x, torch), y,
                 cfg=cmi.Config(kernel=cmi.ELL_ROW, threads_per_row=1, items_per_thread=2))
    assert np.array_equal(host(y), g["f64_y_ell"])


def test_coo_unsorted_entries(cmi, torch_cuda, orc, golden_irregular):
    """The host loop accepts any entry order (coo_spmv.h:59-67); so does the kernel."""
    torch, g = torch_cuda, golden_irregular
    rows, cols = int(g["rows"]), int(g["cols"])
    Ap, Aj, Ax, x = (g[f"f64_{k}"] for k in ("Ap", "Aj", "Ax", "x"))
    Ai = orc.csr_row_indices(Ap)
    perm = np.random.default_rng(5).permutation(len(Ax))
    bound = row_abs(orc, Ap, Aj, Ax, x)
    for kern, ipt in ((cmi.COO_SEGMENTED, 1), (cmi.COO_SEGMENTED, 4), (cmi.COO_LANE4, 1), (cmi.COO_LANE4, 3), (None, 0)):
        y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
        cmi.spmv_coo(rows, cols, dev(Ai[perm], torch), dev(Aj[perm], torch), dev(Ax[perm], torch), dev(x, torch), y,
                     cfg=cmi.Config(kernel=kern, items_per_thread=ipt) if kern is not None else None)
        assert_close(host(y), g["f64_y_coo"], bound, np.float64, f"coo unsorted k{kern} ipt{ipt}")
    # a single swapped pair at an interval boundary, at a lane boundary and inside a lane
    for pos in (255, 256, 3, 5, 1023, len(Ai) - 2):
        I2, J2, V2 = Ai.copy(), Aj.copy(), Ax.copy()
        q = pos + 1
        while q < len(Ai) - 1 and I2[q] == I2[pos]:
            q += 1
        for arr in (I2, J2, V2):
            arr[[pos, q]] = arr[[q, pos]]
        y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
        cmi.spmv_coo(rows, cols, dev(I2, torch), dev(J2, torch), dev(V2, torch), dev(x, torch), y,
                     cfg=cmi.Config(kernel=cmi.COO_LANE4, items_per_thread=1, block_size=64))
        assert_close(host(y), g["f64_y_coo"], bound, np.float64, f"coo one swap at {pos}")
    # unaligned index / value arrays: the lane4 request runs the segmented kernel
    buf_i = torch.zeros(len(Ai) + 4, dtype=torch.int32, device="cuda")
    buf_i[1:1 + len(Ai)].copy_(dev(Ai, torch))
    y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
    cmi.spmv_coo(rows, cols, buf_i[1:1 + len(Ai)], dev(Aj, torch), dev(Ax, torch), dev(x, torch), y,
                 cfg=cmi.Config(kernel=cmi.COO_LANE4))
    assert_close(host(y), g["f64_y_coo"], bound, np.float64, "coo lane4 unaligned")


@pytest.mark.parametrize("shape", ["gaps", "one_row", "long_rows", "all_rows_shared", "tiny", "dense_rows_and_deserts"])
def test_coo_row_sorted_shapes(cmi, torch_cuda, orc, shape):
    """Row-sorted COO (the reference's contract, cusp/coo_matrix.h:72) in the shapes that stress run handling: rows
    without entries (singles, runs, deserts of 10^5 rows, both ends of the matrix), one row holding everything, rows
    of several wave intervals, every row straddling an interval boundary -- y starts poisoned, so a row nobody
    wrote shows, and rows without entries must come out as +0 exactly."""
    torch = torch_cuda
    rng = np.random.default_rng(41)
    if shape == "gaps":           # empty rows everywhere: singles, runs of 2..8, runs of hundreds, both ends
        rows, cols = 60000, 700
        lens = rng.integers(0, 7, size=rows)
        lens[rng.random(rows) < 0.3] = 0
        lens[:37] = 0
        lens[-1500:] = 0
        lens[20000:20900] = 0
        lens[40000:40009] = 0
    elif shape == "one_row":      # one row holds everything: every interval shares it with its neighbours
        rows, cols = 9, 5000
        lens = np.zeros(rows, np.int64)
        lens[4] = 30001
    elif shape == "long_rows":    # rows of several intervals between short ones
        rows, cols = 3000, 4000
        lens = rng.integers(1, 6, size=rows)
        lens[[0, 17, 1500, 2999]] = [777, 2600, 257, 1025]
    elif shape == "all_rows_shared":  # 256-entry rows shifted by one entry: every row straddles a boundary
        rows, cols = 400, 300
        lens = np.full(rows, 256, np.int64)
        lens[0] = 1
    elif shape == "tiny":
        rows, cols = 5, 5
        lens = np.array([0, 2, 0, 1, 0])
    else:                         # dense rows separated by deserts of 10^5 empty rows
        rows, cols = 400000, 1000
        lens = np.zeros(rows, np.int64)
        lens[[5, 100000, 100001, 250000, 399990]] = [900, 3, 1, 4000, 2]
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    nnz = int(Ap[-1])
    Ai = orc.csr_row_indices(Ap)
    Aj = rng.integers(0, cols, size=nnz).astype(np.int32)   # unsorted columns, duplicates: allowed
    Ax = rng.standard_normal(nnz)
    x = rng.standard_normal(cols)
    want = orc.spmv_csr(Ap, Aj, Ax, x)
    bound = row_abs(orc, Ap, Aj, Ax, x)
    dAi, dAj, dAx, dx = dev(Ai, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
    for kern, blk, ipt in itertools.product((cmi.COO_LANE4, cmi.COO_SEGMENTED), (64, 256, 512), (1, 2, 3, 8)):
        y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
        cmi.spmv_coo(rows, cols, dAi, dAj, dAx, dx, y, cfg=cmi.Config(kernel=kern, block_size=blk, items_per_thread=ipt))
        got = host(y)
        assert_close(got, want, bound, np.float64, f"coo {shape} k{kern} b{blk} i{ipt}")
        assert np.array_equal(got[lens == 0], np.zeros(int((lens == 0).sum()))), f"{shape}: an empty row is not +0"
    # the tile kernel (what a plan selects for sorted entries): the host COO loop's bits, rows without entries +0
    want_coo = orc.spmv_coo(rows, Ai, Aj, Ax, x)
    plan = cmi.Plan(cmi.FORMAT_COO, torch.float64, rows, cols, nnz, dAi)
    assert plan.info()["coo_sorted"]
    for swz in (0, 1, 2, 64):
        y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
        cmi.spmv_coo(rows, cols, dAi, dAj, dAx, dx, y, cfg=cmi.Config(kernel=cmi.COO_TILE, xcd_swizzle=swz, nontemporal=2))
        assert np.array_equal(host(y), want_coo), f"coo tile {shape} x{swz}"
    y0t = rng.standard_normal(rows)
    y = dev(y0t, torch)
    cmi.spmv_coo(rows, cols, dAi, dAj, dAx, dx, y, accumulate=True, cfg=cmi.Config(kernel=cmi.COO_TILE, xcd_swizzle=16))
    assert np.array_equal(host(y), orc.spmv_coo(rows, Ai, Aj, Ax, x, y0t)), f"coo tile {shape} accumulate"
    y = dev(y0t, torch)
    cmi.spmv_coo_plan(plan, dAi, dAj, dAx, dx, y, accumulate=True)   # whatever the table's sorted-COO key holds
    assert_close(host(y), orc.spmv_coo(rows, Ai, Aj, Ax, x, y0t), bound + np.abs(y0t), np.float64, f"coo plan {shape} accumulate")
    # the same through the table (NULL config), on another stream, and accumulating (order-agnostic kernel)
    y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        cmi.spmv_coo(rows, cols, dAi, dAj, dAx, dx, y)
    s.synchronize()
    assert_close(host(y), want, bound, np.float64, f"coo {shape} table")
    y0 = rng.standard_normal(rows)
    y = dev(y0, torch)
    cmi.spmv_coo(rows, cols, dAi, dAj, dAx, dx, y, accumulate=True)
    assert_close(host(y), orc.spmv_csr(Ap, Aj, Ax, x, y0), bound + np.abs(y0), np.float64, f"coo {shape} accumulate")


def test_device_coo_to_csr(cmi, torch_cuda, orc, golden_irregular):
    """cmi_coo_row_offsets: row offsets from row-sorted row indices on the device (empty rows at both ends and inside,
    a 5000-entry row), identical to the CSR the COO came from; unsorted or out-of-range indices are reported, not used."""
    torch, g = torch_cuda, golden_irregular
    rows, cols = int(g["rows"]), int(g["cols"])
    Ap, Aj, Ax = g["f64_Ap"], g["f64_Aj"], g["f64_Ax"]
    Ai = orc.csr_row_indices(Ap)
    C = cmi.CooMatrix(rows, cols, len(Aj), dev(Ai, torch), dev(Aj, torch), dev(Ax, torch))
    A = cmi.convert(C, "csr")
    assert np.array_equal(host(A.row_offsets), Ap) and A.num_entries == len(Aj)
    assert A.column_indices.data_ptr() == C.column_indices.data_ptr()       # the entry arrays are shared, not copied
    y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
    cmi.multiply(A, dev(g["f64_x"], torch), y, cfg=cmi.Config(kernel=cmi.CSR_STREAM, threads_per_row=1))
    assert np.array_equal(host(y), g["f64_y_csr"])
    # shapes: no entries at all, one entry, entries only in the last row, deserts of empty rows
    for rws, idx in ((7, []), (7, [3]), (7, [6, 6, 6]), (100000, [0, 5, 5, 99998]), (1, [0, 0])):
        ai = torch.tensor(idx, dtype=torch.int32, device="cuda")
        ap = torch.full((rws + 1,), -7, dtype=torch.int32, device="cuda")
        assert cmi.coo_row_offsets(rws, ai, ap)
        want = np.searchsorted(np.asarray(idx, dtype=np.int64), np.arange(rws + 1), side="left")
        assert np.array_equal(host(ap), want.astype(np.int32)), (rws, idx)
    # not sorted / out of range: reported
    for rws, idx in ((7, [3, 2]), (7, [0, 7]), (7, [-1, 2]), (7, [1, 2, 3, 2, 5])):
        ai = torch.tensor(idx, dtype=torch.int32, device="cuda")
        ap = torch.empty(rws + 1, dtype=torch.int32, device="cuda")
        assert not cmi.coo_row_offsets(rws, ai, ap), (rws, idx)
    # entries in any order: sorted by row on the device first (a copy; stable), as the reference's coo -> csr sorts first
    perm = np.random.default_rng(3).permutation(len(Aj))
    U = cmi.CooMatrix(rows, cols, len(Aj), dev(Ai[perm], torch), dev(Aj[perm], torch), dev(Ax[perm], torch))
    S = cmi.convert(U, "csr")
    assert np.array_equal(host(S.row_offsets), Ap) and np.array_equal(host(U.row_indices), Ai[perm])   # the source is untouched
    order = np.argsort(Ai[perm], kind="stable")
    assert np.array_equal(host(S.column_indices), Aj[perm][order]) and np.array_equal(host(S.values), Ax[perm][order])
    with pytest.raises(Exception):
        bad = Ai[perm].copy(); bad[5] = rows
        cmi.convert(cmi.CooMatrix(rows, cols, len(Aj), dev(bad, torch), dev(Aj[perm], torch), dev(Ax[perm], torch)), "csr")


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_device_ell_and_dia_to_csr(cmi, torch_cuda, orc, golden_irregular, golden_banded, tag):
    """ELL -> CSR and DIA -> CSR built on the device (count per row, exclusive scan over more than one scan tile, scatter):
    the round trips CSR -> ELL -> CSR and CSR -> DIA -> CSR give the matrix back (empty rows, a 5000-entry row, a banded
    rectangular matrix); padding and explicit zeros of the DIA array are dropped as the reference does."""
    torch, g = torch_cuda, golden_irregular
    rows, cols = int(g["rows"]), int(g["cols"])
    Ap, Aj, Ax = g[f"{tag}_Ap"], g[f"{tag}_Aj"], g[f"{tag}_Ax"]
    A = cmi.CsrMatrix(rows, cols, len(Aj), dev(Ap, torch), dev(Aj, torch), dev(Ax, torch))
    E = cmi.convert(A, "ell")
    back = cmi.convert(E, "csr")
    assert back.num_entries == len(Aj)
    assert np.array_equal(host(back.row_offsets), Ap) and np.array_equal(host(back.column_indices), Aj) and np.array_equal(host(back.values), Ax)
    # a bigger matrix: more rows than one scan tile holds several times over, through DIA and ELL
    m, n = 301, 127
    P = cmi.poisson5pt(m, n, "csr", dtype=torch.float64 if tag == "f64" else torch.float32)
    for fmt in ("dia", "ell"):
        Q = cmi.convert(cmi.convert(P, fmt), "csr")
        assert torch.equal(Q.row_offsets, P.row_offsets) and torch.equal(Q.column_indices, P.column_indices) and torch.equal(Q.values, P.values)
    # the gallery's own DIA (built directly) converts to the gallery's CSR
    Q = cmi.convert(cmi.poisson5pt(m, n, "dia", dtype=P.values.dtype), "csr")
    assert torch.equal(Q.row_offsets, P.row_offsets) and torch.equal(Q.column_indices, P.column_indices) and torch.equal(Q.values, P.values)
    # explicit zeros inside the band are dropped (dia_to_other.h: copy_if value != 0); an all-empty matrix works
    D = cmi.poisson5pt(9, 7, "dia", dtype=P.values.dtype)
    D.values[: D.pitch].zero_()      # the first diagonal (offset -m) becomes explicit zeros
    Z = cmi.convert(D, "csr")
    want_cols = [c for c in host(cmi.poisson5pt(9, 7, "csr").column_indices)]
    Pp, Pj = host(cmi.poisson5pt(9, 7, "csr").row_offsets), host(cmi.poisson5pt(9, 7, "csr").column_indices)
    keep = np.concatenate([[j for j in Pj[Pp[i]:Pp[i + 1]] if j != i - 9] for i in range(63)]).astype(np.int32)
    assert np.array_equal(host(Z.column_indices), keep) and Z.num_entries == len(keep) and len(want_cols) > len(keep)
    # HYB -> CSR: a row's ELL entries, then its COO entries -- the CSR it came from, at every split width
    for w in (0, 1, 3, 7, 5000):
        H = cmi.convert(A, "hyb", num_entries_per_row=w)
        back = cmi.convert(H, "csr")
        assert np.array_equal(host(back.row_offsets), Ap) and np.array_equal(host(back.column_indices), Aj) and np.array_equal(host(back.values), Ax), w
    empty = cmi.EllMatrix(5, 5, 0, 0, 32, torch.empty(0, dtype=torch.int32, device="cuda"), torch.empty(0, dtype=P.values.dtype, device="cuda"))
    Zc = cmi.convert(empty, "csr")
    assert Zc.num_entries == 0 and host(Zc.row_offsets).tolist() == [0] * 6


def test_non_default_stream(cmi, torch_cuda, golden_poisson, orc):
    torch, g = torch_cuda, golden_poisson
    Ap, Aj, Ax = orc.poisson5pt_csr(100, 100)
    s = torch.cuda.Stream()
    dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(g["f64_x"], torch)
    y = torch.full((10000,), 10.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        cmi.spmv_csr(10000, 10000, dAp, dAj, dAx, dx, y)
        cmi.spmv_csr(10000, 10000, dAp, dAj, dAx, dx, y, stream=s)
    s.synchronize()
    assert np.array_equal(host(y), g["f64_y_csr"])


# ------------------------------------------------------------------------------------------------
# on-device builders: integer outputs bit-exact vs the oracle's restatement of the reference
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n", [(2, 3), (10, 10), (1, 1), (1, 9), (9, 1), (117, 113), (100, 100)])
def test_device_poisson_builder_matches_oracle(cmi, torch_cuda, orc, m, n):
    torch = torch_cuda
    N = m * n
    Ap, Aj, Ax = orc.poisson5pt_csr(m, n)
    A = cmi.poisson5pt(m, n, "csr")
    assert A.num_entries == len(Ax) == 5 * m * n - 2 * m - 2 * n
    assert np.array_equal(host(A.row_offsets), Ap) and np.array_equal(host(A.column_indices), Aj)
    assert np.array_equal(host(A.values), Ax)
    off, vals, nnz = orc.poisson5pt_dia(m, n)
    D = cmi.poisson5pt(m, n, "dia")
    assert D.num_entries == nnz and np.array_equal(host(D.diagonal_offsets), off) and np.array_equal(host(D.values), vals)
    # row-block shards carry global columns and concatenate to the whole matrix
    cuts = sorted({0, N // 3, N // 2, N - 1 if N > 1 else N, N})
    parts = [cmi.poisson5pt(m, n, "csr", row_begin=a, row_end=b) for a, b in zip(cuts, cuts[1:])]
    assert np.array_equal(np.concatenate([host(p.column_indices) for p in parts]), Aj)
    assert np.array_equal(np.concatenate([host(p.row_offsets)[1:] + Ap[a] for p, a in zip(parts, cuts)]), Ap[1:])
    # conversions
    C = cmi.convert(A, "coo")
    assert np.array_equal(host(C.row_indices), orc.csr_row_indices(Ap))
    width = int(np.diff(Ap).max())
    E = cmi.convert(A, "ell")
    pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width)
    assert E.pitch == pitch and np.array_equal(host(E.column_indices), eAj) and np.array_equal(host(E.values), eAx)
    for w in range(0, width + 1):
        H = cmi.convert(A, "hyb", num_entries_per_row=w)
        p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, w)
        assert np.array_equal(host(H.ell.column_indices), hAj) and np.array_equal(host(H.ell.values), hAx)
        assert np.array_equal(host(H.coo.row_indices), cAi) and np.array_equal(host(H.coo.column_indices), cAj)
        assert np.array_equal(host(H.coo.values), cAx)


# ------------------------------------------------------------------------------------------------
# BASELINE.json full size: poisson5pt 3162 x 3162 (N = 9 998 244, nnz = 49 978 572), fp64
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big(cmi, torch_cuda, orc):
    import oracle
    torch = torch_cuda
    m = n = 3162
    A = cmi.poisson5pt(m, n, "csr")
    N = m * n
    x = oracle.fill_x(N)
    Ap, Aj, Ax = host(A.row_offsets), host(A.column_indices), host(A.values)
    want = orc.spmv_csr(Ap, Aj, Ax, x, omp=True)
    return dict(m=m, n=n, N=N, A=A, x=x, dx=dev(x, torch), want=want, Ap=Ap, Aj=Aj, Ax=Ax)


def test_full_size_structure(cmi, big):
    assert big["N"] == 9998244 and big["A"].num_entries == 49978572
    Ap, Aj = big["Ap"], big["Aj"]
    lens = np.diff(Ap)
    assert lens.min() == 3 and lens.max() == 5 and Ap[-1] == 49978572
    # spot-check rows against the closed form: corner, edge, interior
    m = big["m"]
    for r, cols in ((0, [0, 1, m]), (1, [0, 1, 2, m + 1]), (m + 5, [5, m + 4, m + 5, m + 6, 2 * m + 5])):
        assert Aj[Ap[r]:Ap[r + 1]].tolist() == cols
    assert big["Ax"].sum() == 4.0 * big["N"] - (Ap[-1] - big["N"])


def test_full_size_csr_bit_exact_and_formats_agree(cmi, torch_cuda, orc, big):
    torch = torch_cuda
    A, dx, N, want = big["A"], big["dx"], big["N"], big["want"]
    # sampled values + checksum recorded from the oracle run (pins the on-box oracle itself)
    y = torch.full((N,), 10.0, dtype=torch.float64, device="cuda")
    cmi.multiply(A, dx, y)  # NULL config -> tuning table
    assert np.array_equal(host(y), want), "default CSR kernel is not bit-exact at full size"
    for cfg in (cmi.Config(kernel=cmi.CSR_STREAM, items_per_thread=2, xcd_swizzle=1, nontemporal=1),
                cmi.Config(kernel=cmi.CSR_STREAM, items_per_thread=4, block_size=512),
                cmi.Config(kernel=cmi.CSR_STREAM_PIPE), cmi.Config(kernel=cmi.CSR_STREAM_PIPE, xcd_swizzle=1, blocks_per_cu=4),
                cmi.Config(kernel=cmi.CSR_STREAM_PIPE, block_size=512, nontemporal=1, blocks_per_cu=3),
                cmi.Config(kernel=cmi.CSR_SCALAR)):
        y.fill_(10.0)
        cmi.multiply(A, dx, y, cfg=cfg)
        assert np.array_equal(host(y), want)
    bound = 8.0 * np.abs(big["x"]).max() * np.ones(1)
    for tpr in (4, 8, 64):
        y.fill_(10.0)
        cmi.multiply(A, dx, y, cfg=cmi.Config(kernel=cmi.CSR_VECTOR, threads_per_row=tpr))
        assert np.max(np.abs(host(y) - want)) <= 1e-6 * bound[0]
    # the other formats of the same matrix: ELL K=5 pitch 9 998 272, DIA, COO, HYB
    E = cmi.convert(A, "ell")
    assert E.pitch == 9998272 and E.num_entries_per_row == 5
    y.fill_(10.0)
    cmi.multiply(E, dx, y)
    assert np.array_equal(host(y), want)
    y.fill_(10.0)
    cmi.multiply(E, dx, y, cfg=cmi.Config(kernel=cmi.ELL_ROW, items_per_thread=2, nontemporal=1))
    assert np.array_equal(host(y), want)
    del E
    D = cmi.poisson5pt(big["m"], big["n"], "dia")
    for rpl in (1, 2):
        y.fill_(10.0)
        cmi.multiply(D, dx, y, cfg=cmi.Config(kernel=cmi.DIA_ROW, items_per_thread=rpl))
        assert np.array_equal(host(y), want)
    del D
    # sorted COO through its plan (row offsets + the CSR kernel) and HYB through its plan (one launch for the light splits, ELL + the
    # COO part's plan for the heavy ones): per row the same chain of adds as the CSR loop -- the same bits at full size
    C = cmi.convert(A, "coo")
    y.fill_(10.0)
    cmi.multiply(C, dx, y)
    assert C.plan().info()["storage_order_sums"] and np.array_equal(host(y), want)
    y.fill_(10.0)
    cmi.multiply(C, dx, y, cfg=cmi.Config(kernel=cmi.COO_TILE, nontemporal=3, xcd_swizzle=32))   # the COO format's own kernel
    assert np.array_equal(host(y), want)
    y.fill_(10.0)
    cmi.spmv_coo(N, N, C.row_indices, C.column_indices, C.values, dx, y)                         # plan-less: any order, atomics
    assert np.max(np.abs(host(y) - want)) <= 1e-6 * bound[0]
    del C
    for w, launches in ((1, 2), (3, 1), (4, 1), (5, 1)):
        H = cmi.convert(A, "hyb", num_entries_per_row=w)
        assert H.coo.num_entries == int(np.maximum(np.diff(big["Ap"]) - w, 0).sum())
        y.fill_(10.0)
        cmi.multiply(H, dx, y)
        assert H.plan().hyb_launches() == launches, (w, H.plan().hyb_launches())
        assert H.plan().info()["storage_order_sums"] and np.array_equal(host(y), want), w
        del H


def test_full_size_properties(cmi, torch_cuda, big):
    """Size-independent properties: linearity, checksum through column sums, idempotence."""
    torch = torch_cuda
    A, N, dx = big["A"], big["N"], big["dx"]
    y1 = torch.empty(N, dtype=torch.float64, device="cuda")
    y2 = torch.empty_like(y1)
    cmi.multiply(A, dx, y1)
    cmi.multiply(A, dx, y2)
    assert torch.equal(y1, y2)  # deterministic / idempotent
    # A is symmetric with known column sums: sum(y) = sum_j colsum_j x_j, colsum_j = 4 - (#neighbours)
    colsum = torch.zeros(N, dtype=torch.float64, device="cuda")
    ones = torch.ones(N, dtype=torch.float64, device="cuda")
    cmi.multiply(A, ones, colsum)  # row sums == column sums (symmetric)
    lhs = float(y1.sum())
    rhs = float((colsum * dx).sum())
    assert abs(lhs - rhs) <= 1e-9 * float(dx.abs().sum())
    # linearity: A(2x + 3*1) = 2 A x + 3 A 1, exact up to rounding of the scaled inputs
    z = 2.0 * dx + 3.0 * ones
    yz = torch.empty_like(y1)
    cmi.multiply(A, z, yz)
    assert float((yz - (2.0 * y1 + 3.0 * colsum)).abs().max()) <= 1e-12 * 32
    # y += A x twice == 2 A x
    acc = torch.zeros_like(y1)
    cmi.multiply(A, dx, acc, accumulate=True)
    cmi.multiply(A, dx, acc, accumulate=True)
    assert float((acc - 2.0 * y1).abs().max()) <= 1e-12


# ------------------------------------------------------------------------------------------------
# BASELINE.json configs[4] size on ONE device: poisson5pt 10000 x 10000 (N = 1e8, nnz = 499 960 000) -- value arrays
# of 4 GB, i.e. byte offsets beyond 2^32 and entry positions near the int32 limit, in every format and kernel.
# Checked against the stencil's closed form (bench.stencil_expected: bit-identical to the oracle, tests/
# test_bench_helpers.py) so that no 1e8-row host oracle run is needed.
# ------------------------------------------------------------------------------------------------
def test_1e8_rows_every_format_and_kernel(cmi, torch_cuda):
    import bench
    torch = torch_cuda
    if torch.cuda.get_device_properties(0).total_memory < 64 * 2**30:
        pytest.skip("needs 64 GiB of device memory")
    m = n = 10000
    N = m * n
    want = bench.stencil_expected(torch, cmi, m, n, 0, N, "cuda")
    bound = 8.0 * 0.51
    dx = cmi.fill_x(N).to("cuda")
    y = torch.full((N,), 10.0, dtype=torch.float64, device="cuda")

    def check(M, cfg, exact, what):
        y.fill_(10.0)
        cmi.multiply(M, dx, y, cfg=cfg)
        if exact:
            assert torch.equal(y, want), what
        else:
            assert float((y - want).abs().max()) <= 1e-6 * bound, what

    A = cmi.poisson5pt(m, n, "csr")
    assert A.num_rows == N and A.num_entries == 499960000 == cmi.poisson5pt_num_entries(m, n)
    assert int(A.row_offsets[-1]) == 499960000 and int(A.column_indices[-1]) == N - 1
    check(A, None, True, "csr default")
    for cfg in (cmi.Config(kernel=cmi.CSR_SCALAR), cmi.Config(kernel=cmi.CSR_STREAM, items_per_thread=2, block_size=512),
                cmi.Config(kernel=cmi.CSR_STREAM_PIPE), cmi.Config(kernel=cmi.CSR_STREAM_PIPE, xcd_swizzle=1, blocks_per_cu=4)):
        check(A, cfg, True, f"csr {cfg.as_dict()}")
    for cfg in (cmi.Config(kernel=cmi.CSR_VECTOR, threads_per_row=4), cmi.Config(kernel=cmi.CSR_BALANCED)):
        check(A, cfg, False, f"csr {cfg.as_dict()}")
    # fused SpMV + dot at this size (more tiles than the partial list holds -> the plain dot follows the SpMV)
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    ws = cmi.blas_workspace("cuda")
    y.fill_(10.0)
    cmi.spmv_csr_dot(N, N, A.row_offsets, A.column_indices, A.values, dx, y, dx, res, ws)
    assert torch.equal(y, want)
    ref_dot = float((want * dx).sum())
    assert abs(float(res.item()) - ref_dot) <= 1e-9 * float((want.abs() * dx.abs()).sum())
    C = cmi.convert(A, "coo")
    assert C.num_entries == A.num_entries and int(C.row_indices[-1]) == N - 1
    check(C, None, False, "coo default")
    check(C, cmi.Config(kernel=cmi.COO_SEGMENTED), False, "coo segmented")
    del C
    H = cmi.convert(A, "hyb", num_entries_per_row=4)
    assert H.coo.num_entries == (m - 2) * (n - 2)
    check(H, None, False, "hyb K=4")
    del H
    E = cmi.convert(A, "ell")
    assert E.num_entries_per_row == 5 and E.pitch >= N
    check(E, None, True, "ell default")
    check(E, cmi.Config(kernel=cmi.ELL_ROW, items_per_thread=2), True, "ell 2 rows per lane")
    del E, A
    D = cmi.poisson5pt(m, n, "dia")
    for rpl in (1, 2):
        check(D, cmi.Config(kernel=cmi.DIA_ROW, items_per_thread=rpl), True, f"dia {rpl} rows per lane")


# ------------------------------------------------------------------------------------------------
# CSR -> DIA on the device
# ------------------------------------------------------------------------------------------------
def test_device_csr_to_dia_matches_oracle(cmi, torch_cuda, orc, golden_irregular):
    """cmi_csr_diagonals + cmi_csr_to_dia_* reproduce the arrays of the reference conversion
    (csr_to_other.h:73-153: occupied diagonals ascending, zero fill) bit for bit; the SpMV on the result
    equals the CSR SpMV; the fill-in guard rejects a matrix with too many diagonals."""
    torch = torch_cuda
    cases = []
    Ap, Aj, Ax = orc.poisson5pt_csr(61, 47)
    cases.append((61 * 47, 61 * 47, Ap, Aj, Ax))
    # rectangular banded matrix with an empty row (built here; float32 and float64)
    rng = np.random.default_rng(5)
    rows, cols, offs = 700, 900, (-450, -3, 0, 1, 17, 400, 880)
    ap, aj, ax = [0], [], []
    for i in range(rows):
        if i != 123:
            for o in offs:
                if 0 <= i + o < cols:
                    aj.append(i + o)
                    ax.append(rng.standard_normal())
        ap.append(len(aj))
    for dt in (np.float64, np.float32):
        cases.append((rows, cols, np.array(ap, np.int32), np.array(aj, np.int32), np.array(ax, dt)))
    for rows, cols, Ap, Aj, Ax in cases:
        A = cmi.CsrMatrix(rows, cols, len(Aj), dev(Ap, torch), dev(Aj, torch), dev(Ax, torch))
        D = cmi.convert(A, "dia")
        pitch, off, vals = orc.csr_to_dia(rows, cols, Ap, Aj, Ax)
        assert D.pitch == pitch and np.array_equal(host(D.diagonal_offsets), off)
        assert np.array_equal(host(D.values), vals)
        x = np.random.default_rng(1).standard_normal(cols).astype(Ax.dtype)
        y = torch.empty(rows, dtype=A.values.dtype, device="cuda")
        cmi.multiply(D, dev(x, torch), y)
        assert np.array_equal(host(y), orc.spmv_dia(rows, cols, pitch, off, vals, x))
    # the irregular golden matrix occupies 2365 diagonals: 3.5e6 slots for 15378 entries -> refused
    g = golden_irregular
    A = cmi.CsrMatrix(int(g["rows"]), int(g["cols"]), len(g["f64_Aj"]), dev(g["f64_Ap"], torch), dev(g["f64_Aj"], torch), dev(g["f64_Ax"], torch))
    with pytest.raises(ValueError, match="fill-in"):
        cmi.convert(A, "dia")


# ------------------------------------------------------------------------------------------------
# fused y = A x, <y, w>  (the CG step cg.inl:80-83 in one pass)
# ------------------------------------------------------------------------------------------------
def test_spmv_csr_dot_every_variant(cmi, torch_cuda, golden_irregular):
    """cmi_spmv_csr_dot_f64: y bit-identical to cmi_spmv_csr_f64 under the same config, the dot within
    1e-12 * sum|y_i w_i| of numpy's -- for the kernels that fuse it (csr_stream) and the ones that fall
    back to SpMV + dot alike."""
    torch = torch_cuda
    g = golden_irregular
    rows, cols = int(g["rows"]), int(g["cols"])
    Ap, Aj, Ax, x = dev(g["f64_Ap"], torch), dev(g["f64_Aj"], torch), dev(g["f64_Ax"], torch), dev(g["f64_x"], torch)
    w = np.random.default_rng(7).standard_normal(rows)
    dw = dev(w, torch)
    ws = cmi.blas_workspace()
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    for name, _, cfg in csr_variants(cmi, small=True) + [("table", True, None)]:
        y_plain = torch.full((rows,), 3.0, dtype=torch.float64, device="cuda")
        y_fused = torch.full((rows,), -5.0, dtype=torch.float64, device="cuda")
        cmi.spmv_csr(rows, cols, Ap, Aj, Ax, x, y_plain, cfg=cfg)
        res.fill_(float("nan"))
        cmi.spmv_csr_dot(rows, cols, Ap, Aj, Ax, x, y_fused, dw, res, ws, cfg=cfg)
        if cfg is None or cfg.kernel == cmi.CSR_BALANCED:  # (the 5000-entry row makes the profile pick csr_balanced too)
            # atomics on rows that span tiles: order-dependent rounding
            assert torch.allclose(y_plain, y_fused, rtol=1e-12, atol=1e-12), name
        else:
            assert torch.equal(y_plain, y_fused), name
        yh = host(y_fused)
        assert abs(float(res) - float(np.dot(yh, w))) <= 1e-12 * float(np.abs(yh * w).sum()), name
    # twice the same call: the same bits (fixed reduction tree, no atomics)
    cmi.spmv_csr_dot(rows, cols, Ap, Aj, Ax, x, y_fused, dw, res, ws)
    first = float(res)
    cmi.spmv_csr_dot(rows, cols, Ap, Aj, Ax, x, y_fused, dw, res, ws)
    assert float(res) == first


@pytest.mark.parametrize("case", ["folded", "over_capacity", "w_is_x"])
def test_spmv_csr_dot_partial_list_lengths(cmi, torch_cuda, case):
    """Long per-tile partial lists take the middle fold stage (> 2048 tiles); more tiles than the
    workspace holds (> 131072) fall back to SpMV + dot; w == x is the CG call."""
    torch = torch_cuda
    m = 700
    A = cmi.poisson5pt(m, m, "csr")
    n = A.num_rows
    x = cmi.fill_x(n).cuda()
    w = x if case == "w_is_x" else torch.from_numpy(np.random.default_rng(3).standard_normal(n)).cuda()
    cfg = {"folded": cmi.Config(kernel=cmi.CSR_STREAM, block_size=256, rows_per_block=64, nontemporal=2),      # 7657 tiles
           "over_capacity": cmi.Config(kernel=cmi.CSR_STREAM, block_size=64, rows_per_block=2),                # 245000 tiles
           "w_is_x": None}[case]
    ws = cmi.blas_workspace()
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    y0, y1 = torch.empty(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda")
    cmi.spmv_csr(n, n, A.row_offsets, A.column_indices, A.values, x, y0, cfg=cfg)
    cmi.spmv_csr_dot(n, n, A.row_offsets, A.column_indices, A.values, x, y1, w, res, ws, cfg=cfg)
    assert torch.equal(y0, y1)
    yh, wh = host(y1), host(w)
    assert abs(float(res) - float(np.dot(yh, wh))) <= 1e-12 * float(np.abs(yh * wh).sum())


def test_spmv_csr_dot_degenerate(cmi, torch_cuda):
    torch = torch_cuda
    ws = cmi.blas_workspace()
    res = torch.full((1,), 9.0, dtype=torch.float64, device="cuda")
    e_i, e_d = torch.zeros(0, dtype=torch.int32, device="cuda"), torch.zeros(0, dtype=torch.float64, device="cuda")
    cmi.spmv_csr_dot(0, 0, torch.zeros(1, dtype=torch.int32, device="cuda"), e_i, e_d, e_d, e_d, e_d, res, ws)
    assert float(res) == 0.0                                     # empty matrix: <y, w> = 0 is still written
    Ap = torch.zeros(6, dtype=torch.int32, device="cuda")         # 5 empty rows
    y = torch.full((5,), 2.0, dtype=torch.float64, device="cuda")
    w = torch.ones(5, dtype=torch.float64, device="cuda")
    res.fill_(9.0)
    cmi.spmv_csr_dot(5, 3, Ap, e_i, e_d, torch.ones(3, dtype=torch.float64, device="cuda"), y, w, res, ws)
    assert float(res) == 0.0 and host(y).tolist() == [0.0] * 5
    with pytest.raises(Exception):
        cmi.spmv_csr_dot(5, 3, Ap, e_i, e_d, torch.ones(3, dtype=torch.float64, device="cuda"), y, w[:4], res, ws)


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_spmv_ell_and_dia_dot(cmi, torch_cuda, orc, tag):
    """cmi_spmv_{ell,dia}_dot_{f64,f32}: y bit-identical to the plain multiply (every launch shape, ELLR too), the dot (a double
    for both value types) within 1e-12 * sum|y_i w_i| of numpy's double dot of the same y, repeatable bit for bit."""
    torch = torch_cuda
    dtype = np.float64 if tag == "f64" else np.float32
    m, n = 301, 199
    Ap, Aj, Ax = orc.poisson5pt_csr(m, n)
    Ax = Ax.astype(dtype)
    N = m * n
    rng = np.random.default_rng(9)
    x, w = rng.standard_normal(N).astype(dtype), rng.standard_normal(N).astype(dtype)
    dx, dw = dev(x, torch), dev(w, torch)
    ws = cmi.blas_workspace()
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, 5)
    deAj, deAx = dev(eAj, torch), dev(eAx, torch)
    rl = torch.empty(N, dtype=torch.int32, device="cuda")
    cmi.ell_row_lengths(N, 5, pitch, deAj, rl)
    for rpl, nt, ellr, blk in itertools.product((1, 2), (0, 3), (False, True), (64, 256, 1024)):
        cfg = cmi.Config(kernel=cmi.ELL_ROW, block_size=blk, items_per_thread=rpl, nontemporal=nt,
                         xcd_swizzle=(0, 1, 5, 32)[(rpl + nt + blk // 64) % 4])  # partial index = tile, whatever the dealing
        y0 = torch.full((N,), 3.0, dtype=dx.dtype, device="cuda")
        y1 = torch.full((N,), -3.0, dtype=dx.dtype, device="cuda")
        cmi.spmv_ell(N, N, 5, pitch, deAj, deAx, dx, y0, row_lengths=rl if ellr else None, cfg=cfg)
        res.fill_(float("nan"))
        cmi.spmv_ell_dot(N, N, 5, pitch, deAj, deAx, dx, y1, dw, res, ws, row_lengths=rl if ellr else None, cfg=cfg)
        assert torch.equal(y0, y1), (rpl, nt, ellr, blk)
        yh = host(y1).astype(np.float64)
        assert abs(float(res) - float(np.dot(yh, w.astype(np.float64)))) <= 1e-12 * float(np.abs(yh * w).sum()), (rpl, nt, ellr, blk)
    pd, off, vals = orc.csr_to_dia(N, N, Ap, Aj, Ax)
    doff, dvals = dev(off, torch), dev(vals, torch)
    for rpl, nt, blk in itertools.product((1, 2), (0, 2), (64, 256, 512)):
        cfg = cmi.Config(kernel=cmi.DIA_ROW, block_size=blk, items_per_thread=rpl, nontemporal=nt,
                         xcd_swizzle=(0, 1, 5, 32)[(rpl + nt + blk // 64) % 4])
        y0 = torch.full((N,), 3.0, dtype=dx.dtype, device="cuda")
        y1 = torch.full((N,), -3.0, dtype=dx.dtype, device="cuda")
        cmi.spmv_dia(N, N, len(off), pd, doff, dvals, dx, y0, cfg=cfg)
        cmi.spmv_dia_dot(N, N, len(off), pd, doff, dvals, dx, y1, dw, res, ws, cfg=cfg)
        assert torch.equal(y0, y1), (rpl, nt, blk)
        yh = host(y1).astype(np.float64)
        assert abs(float(res) - float(np.dot(yh, w.astype(np.float64)))) <= 1e-12 * float(np.abs(yh * w).sum()), (rpl, nt, blk)
        first = float(res)
        cmi.spmv_dia_dot(N, N, len(off), pd, doff, dvals, dx, y1, dw, res, ws, cfg=cfg)
        assert float(res) == first
    # more workgroups than the workspace holds partials (block 64 on 5M rows): falls back to SpMV + dot
    big = cmi.poisson5pt(2300, 2300, "dia", dtype=dx.dtype)
    nb = big.num_rows
    xb = cmi.fill_x(nb, dx.dtype, "cuda")
    yb0, yb1 = torch.empty(nb, dtype=dx.dtype, device="cuda"), torch.empty(nb, dtype=dx.dtype, device="cuda")
    cfg = cmi.Config(kernel=cmi.DIA_ROW, block_size=64, items_per_thread=1)
    cmi.multiply(big, xb, yb0, cfg=cfg)
    cmi.spmv_dia_dot(nb, nb, 5, big.pitch, big.diagonal_offsets, big.values, xb, yb1, xb, res, ws, cfg=cfg)
    assert torch.equal(yb0, yb1)
    assert abs(float(res) - float(torch.dot(yb1.double(), xb.double()))) <= 1e-10 * float((yb1.double() * xb.double()).abs().sum())


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_ell_wide_rows_lanes_per_row(cmi, torch_cuda, orc, tag):
    """Few, wide rows: the auto rule gives a row several lanes (ref: THREADS_PER_ROW of ktt kernels/ell_kernel.h:102-109); every
    lane count agrees with the host loop within 1e-6 of sum|a_ij x_j|, one lane per row bit for bit; the fused dot falls back
    to the separate dot and still returns <y, w>."""
    torch = torch_cuda
    dtype = np.float64 if tag == "f64" else np.float32
    rng = np.random.default_rng(21)
    rows, cols = 3001, 40000
    lens = rng.integers(0, 200, size=rows)
    lens[7] = 0
    lens[11] = 199
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    Aj = np.concatenate([np.sort(rng.choice(cols, size=l, replace=False)) for l in lens]).astype(np.int32)
    Ax = rng.standard_normal(len(Aj)).astype(dtype)
    x = rng.standard_normal(cols).astype(dtype)
    width = int(lens.max())
    pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width)
    want = orc.spmv_ell(rows, width, pitch, eAj, eAx, x)
    y0 = rng.standard_normal(rows).astype(dtype)
    want_acc = orc.spmv_ell(rows, width, pitch, eAj, eAx, x, y0)
    bound = row_abs(orc, Ap, Aj, Ax, x)
    deAj, deAx, dx = dev(eAj, torch), dev(eAx, torch), dev(x, torch)
    rl = torch.empty(rows, dtype=torch.int32, device="cuda")
    cmi.ell_row_lengths(rows, width, pitch, deAj, rl)
    plan = cmi.Plan(cmi.FORMAT_ELL, dx.dtype, rows, cols, rows * width, None)
    assert plan.info()["storage_order_sums"] is False  # 3001 rows x 199 slots: the auto rule splits the rows
    one = cmi.Plan(cmi.FORMAT_ELL, dx.dtype, rows, cols, rows * width, None, cfg=cmi.Config(kernel=cmi.ELL_ROW, threads_per_row=1))
    assert one.info()["storage_order_sums"] is True
    for lanes, ellr, acc in itertools.product((0, 1, 2, 4, 8, 16), (False, True), (False, True)):
        y = dev(y0, torch).clone() if acc else torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
        cmi.spmv_ell(rows, cols, width, pitch, deAj, deAx, dx, y, row_lengths=rl if ellr else None, accumulate=acc,
                     cfg=None if lanes == 0 else cmi.Config(kernel=cmi.ELL_ROW, threads_per_row=lanes))
        w_ = want_acc if acc else want
        if lanes == 1:
            assert np.array_equal(host(y), w_), (lanes, ellr, acc)
        else:
            assert_close(host(y), w_, bound + (np.abs(y0) if acc else 0), dtype, f"ell lanes{lanes} ellr{ellr} acc{acc}")
    if tag == "f64":
        wv = rng.standard_normal(rows)
        res = torch.zeros(1, dtype=torch.float64, device="cuda")
        y = torch.zeros(rows, dtype=torch.float64, device="cuda")
        cmi.spmv_ell_dot(rows, cols, width, pitch, deAj, deAx, dx, y, dev(wv, torch), res, cmi.blas_workspace(),
                         cfg=cmi.Config(kernel=cmi.ELL_ROW, threads_per_row=8))
        yh = host(y)
        assert_close(yh, want, bound, dtype, "ell lanes8 dot: y")
        assert abs(float(res) - float(np.dot(yh, wv))) <= 1e-12 * float(np.abs(yh * wv).sum())


def test_cg_update_long_partial_list(cmi, torch_cuda):
    """cmi_cg_update_f64 on 3M+1 elements leaves > 2048 partials: the folded reduction tree."""
    torch = torch_cuda
    rng = np.random.default_rng(11)
    n = 3_000_001
    p, y, x, r = (rng.standard_normal(n) for _ in range(4))
    dp, dy, dx, dr = (dev(a, torch) for a in (p, y, x, r))
    rz = torch.tensor([3.0], dtype=torch.float64, device="cuda")
    yp = torch.tensor([-1.5], dtype=torch.float64, device="cuda")
    rr = torch.zeros(1, dtype=torch.float64, device="cuda")
    ws = cmi.blas_workspace()
    ws.fill_(float("nan"))  # nothing in the workspace (partials, folded list, ticket) may need initialising
    mirror = cmi.HostScalar()
    cmi.cg_update(rz, yp, dp, dy, dx, dr, rr, ws, mirror=mirror)
    assert mirror.wait() == float(rr)  # the reduction's second destination: page-locked host memory
    first = float(rr)
    dx2, dr2 = dev(x, torch), dev(r, torch)
    cmi.cg_update(rz, yp, dp, dy, dx2, dr2, rr, ws)  # ticket left at 0 by the previous fold; same tree, same bits
    assert float(rr) == first and mirror.wait() == first
    mirror.close()
    alpha = 3.0 / -1.5
    r_new = (-alpha) * y + r
    assert np.array_equal(host(dx), alpha * p + x)
    assert np.array_equal(host(dr), r_new)
    assert abs(float(rr) - float(np.dot(r_new, r_new))) <= 1e-12 * float(np.dot(r_new, r_new))


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 3_000_001])
def test_cg_eight_pass_split_is_the_nine_pass_iteration(cmi, torch_cuda, n):
    """cmi_cg_update(x = NULL) + cmi_cg_direction_x == cmi_cg_update + cmi_cg_direction, bit for bit: the x update moved
    to the pass that already holds p (one read of p less), with the update kernel's own expressions; odd lengths and
    unaligned views take the scalar paths."""
    torch = torch_cuda
    rng = np.random.default_rng(n)
    for shift in (0, 1):  # 1: views that are not 16-byte aligned
        def vec():
            buf = dev(rng.standard_normal(n + 2), torch)
            return buf[shift:shift + n]
        p, y, x, r = vec(), vec(), vec(), vec()
        p2, x2, r2 = p.clone(), x.clone(), r.clone()
        if shift:  # clone() re-aligns: put the copies back on odd offsets
            def odd(t):
                buf = torch.empty(n + 2, dtype=torch.float64, device="cuda")
                buf[1:1 + n].copy_(t)
                return buf[1:1 + n]
            p2, x2, r2 = odd(p), odd(x), odd(r)
        rz = torch.tensor([1.75], dtype=torch.float64, device="cuda")
        yp = torch.tensor([0.6], dtype=torch.float64, device="cuda")
        rr_a = torch.zeros(1, dtype=torch.float64, device="cuda")
        rr_b = torch.zeros(1, dtype=torch.float64, device="cuda")
        ws = cmi.blas_workspace()
        cmi.cg_update(rz, yp, p, y, x, r, rr_a, ws)
        cmi.cg_direction(rr_a, rz, r, p)
        cmi.cg_update(rz, yp, None, y, None, r2, rr_b, ws)
        cmi.cg_direction_x(rr_b, rz, yp, r2, p2, x2)
        assert float(rr_a) == float(rr_b)
        assert torch.equal(r, r2) and torch.equal(x, x2) and torch.equal(p, p2)


# ------------------------------------------------------------------------------------------------
# BLAS-1 used by cg
# ------------------------------------------------------------------------------------------------
def test_blas1(cmi, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(1)
    ws = cmi.blas_workspace()
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    for n in (1, 2, 63, 64, 1000, 100003, 4_000_001):
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        dx, dy = dev(x, torch), dev(y, torch)
        cmi.blas_dot(dx, dy, res, ws)
        assert abs(float(res) - float(np.dot(x, y))) <= 1e-12 * float(np.abs(x * y).sum()) + 1e-300
        cmi.blas_nrm2(dx, res, ws)
        assert abs(float(res) - float(np.linalg.norm(x))) <= 1e-12 * float(np.linalg.norm(x))
        cmi.blas_axpy(0.75, dx, dy)
        assert np.array_equal(host(dy), 0.75 * x + y)  # one multiply + one add per element, unfused
        dz = torch.empty_like(dx)
        cmi.blas_axpby(2.0, dx, -0.5, dy, dz)
        assert np.array_equal(host(dz), 2.0 * x + (-0.5) * (0.75 * x + y))
        cmi.blas_copy(dz, dy)
        assert torch.equal(dy, dz)
        cmi.blas_fill(3.25, dy)
        assert host(dy).tolist() == [3.25] * n if n < 100 else float(dy.min()) == 3.25 == float(dy.max())
    # f32: same entry points, products and partial sums of the reductions kept in double
    res32 = torch.zeros(1, dtype=torch.float32, device="cuda")
    for n in (3, 64, 100003):
        x, y = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
        dx, dy = dev(x, torch), dev(y, torch)
        cmi.blas_dot(dx, dy, res32, ws)
        assert float(res32) == float(np.float32(np.dot(x.astype(np.float64), y.astype(np.float64)))) or \
            abs(float(res32) - float(np.dot(x.astype(np.float64), y.astype(np.float64)))) <= 1e-6 * float(np.abs(x * y).sum())
        cmi.blas_nrm2(dx, res32, ws)
        assert abs(float(res32) - float(np.linalg.norm(x.astype(np.float64)))) <= 1e-6 * float(np.linalg.norm(x))
        cmi.blas_axpy(0.75, dx, dy)
        assert np.array_equal(host(dy), np.float32(0.75) * x + y)
        dz = torch.empty_like(dx)
        cmi.blas_axpby(2.0, dx, -0.5, dy, dz)
        assert np.array_equal(host(dz), np.float32(2.0) * x + np.float32(-0.5) * (np.float32(0.75) * x + y))
        cmi.blas_fill(1.5, dz)
        assert float(dz.min()) == 1.5 == float(dz.max())
    # unaligned views fall back to scalar accesses
    buf = torch.arange(1001, dtype=torch.float64, device="cuda")
    v = buf[1:]
    cmi.blas_dot(v, v, res, ws)
    assert float(res) == float((np.arange(1, 1001, dtype=np.float64) ** 2).sum())


# ------------------------------------------------------------------------------------------------
# BASELINE.json configs[3]: the SuiteSparse irregular set (nlpkkt120, ldoor, thermal2)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,scale", [("thermal2", 0.03), ("ldoor", 0.03), ("nlpkkt120", 0.01)])
def test_suitesparse_like_matrices_every_kernel(cmi, torch_cuda, orc, name, scale):
    """The three matrices of the reference's CSR-vector threads-per-row sweep (performance/csr_vector/csr_vector.cu:41-62,
    86-110 over testing/UF downloads): the real file when CMI_SUITESPARSE_DIR provides it (dimensions from the file), else a
    seeded stand-in with the collection's published row-length statistics (tools/suitesparse_like.py) -- every CSR variant,
    the table, a plan, COO / ELL / HYB against the oracle."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import suitesparse_like as ssl
    torch = torch_cuda
    Ap, Aj, Ax, source = ssl.load(name, scale)
    st = ssl.stats(Ap, Aj)
    pub = ssl.PUBLISHED[name]
    print(f"{name}: {source}: {st}")
    if source.startswith("seeded"):  # the stand-in must look like the matrix it stands for
        assert abs(st["mean"] - pub["mean"]) <= 0.08 * pub["mean"], (st, pub)
        assert st["max"] <= pub["max"] and st["max"] >= 0.85 * pub["max"], (st, pub)
    rows = cols = st["rows"]
    rng = np.random.default_rng(5)
    x = rng.standard_normal(cols)
    y0 = rng.standard_normal(rows)
    Ai = orc.csr_row_indices(Ap)
    width = int(np.diff(Ap).max())
    pitch, eAj, eAx = orc.csr_to_ell(Ap, Aj, Ax, width)
    hw = max(1, int(round(st["mean"])))
    p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, hw)

    def all_of(y0_):
        return {"csr": orc.spmv_csr(Ap, Aj, Ax, x, y0_), "coo": orc.spmv_coo(rows, Ai, Aj, Ax, x, y0_),
                "ell": orc.spmv_ell(rows, width, pitch, eAj, eAx, x, y0_), "hyb": orc.spmv_hyb(rows, hw, p, hAj, hAx, cAi, cAj, cAx, x, y0_)}

    run_all_formats(cmi, torch, orc, rows, cols, Ap, Aj, Ax, x, all_of(None), all_of(y0), y0, hw, f"{name}-like")
    # the reference's sweep itself: csr_vector at every threads-per-row, against the oracle
    bound = row_abs(orc, Ap, Aj, Ax, x)
    dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
    want = orc.spmv_csr(Ap, Aj, Ax, x)
    for tpr in (2, 4, 8, 16, 32, 64):
        y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
        cmi.spmv_csr(rows, cols, dAp, dAj, dAx, dx, y, cfg=cmi.Config(kernel=cmi.CSR_VECTOR, threads_per_row=tpr))
        assert_close(host(y), want, bound, np.float64, f"{name}-like csr_vector tpr{tpr}")


def test_fold_handoff_is_stable_under_load(cmi, torch_cuda):
    """The multi-workgroup fold of a long partial list hands its chunk sums to the last-arriving workgroup with write-through
    stores, a drained counter add and sc1 loads -- no agent-scope fences (blas1.hip dot_fold_final_kernel).  A stale or torn
    hand-off would change the scalar: thousands of folds, idle and beside a streaming kernel on another stream (uneven load,
    consumer caches warm), must all return the first call's bits; so must the fused SpMV + <y, w> and cg_update."""
    torch = torch_cuda
    rng = np.random.default_rng(77)
    n = 6_000_001                                   # > 2048 partials for every producer below
    a, b = dev(rng.standard_normal(n), torch), dev(rng.standard_normal(n), torch)
    ws = cmi.blas_workspace()
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    A = cmi.poisson5pt(1500, 1500, "csr")
    N = A.num_rows
    x = cmi.fill_x(N).cuda()
    y = torch.empty(N, dtype=torch.float64, device="cuda")
    ws2 = cmi.blas_workspace()
    res2 = torch.zeros(1, dtype=torch.float64, device="cuda")
    rz = torch.tensor([3.0], dtype=torch.float64, device="cuda")
    yp = torch.tensor([-1.5], dtype=torch.float64, device="cuda")
    rr = torch.zeros(1, dtype=torch.float64, device="cuda")
    r0 = dev(rng.standard_normal(n), torch)
    noise_in = torch.ones(1 << 25, dtype=torch.float64, device="cuda")
    noise_out = torch.empty_like(noise_in)
    side = torch.cuda.Stream()

    def one_round():
        cmi.blas_dot(a, b, res, ws)                                   # dot_partial -> fold
        d = float(res)
        cmi.spmv_csr_dot(N, N, A.row_offsets, A.column_indices, A.values, x, y, x, res2, ws2)   # per-tile partials -> fold
        s = float(res2)
        r = r0.clone()
        cmi.cg_update(rz, yp, None, b, None, r, rr, ws)               # near one-shot grid: tens of thousands of partials
        return d, s, float(rr)

    first = one_round()
    for phase in ("idle", "beside a streaming kernel"):
        for it in range(400):
            if phase != "idle" and it % 4 == 0:
                with torch.cuda.stream(side):
                    cmi.blas_axpby(1.0, noise_in, 2.0, noise_in, noise_out)
            assert one_round() == first, (phase, it)
    side.synchronize()
