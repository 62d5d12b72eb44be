"""Round-3 additions on an MI355X, through the C-ABI: the wave-private 16-byte-vector kernel for FEM-like rows (CMI_CSR_STREAM_WAVEV),
cmi_plan_validate / the row-offset check of cmi_plan_create, the fenced fold hand-off of the reductions."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    return torch


def dev(a, torch):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _random_csr(rng, rows, cols, lens, dtype):
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    nnz = int(Ap[-1])
    Aj = rng.integers(0, cols, size=nnz).astype(np.int32)  # (unsorted, duplicates allowed: the sums do not care)
    Ax = rng.standard_normal(nnz).astype(dtype)
    return Ap, Aj, Ax


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("vectors", [0, 1, 2, 4])
def test_wavev_kernel_bit_exact_on_irregular_rows(cmi, torch_cuda, orc, tag, vectors):
    """reference arithmetic: cusp/system/detail/sequential/multiply/csr_spmv.h:42-74 (storage-order sums) -- every row of every
    matrix below must have the host loop's bits: FEM-like lengths, empty rows, a stretch of 300 empty rows (more rows than lanes
    in one wave tile), the longest row the tile admits, an entry count that is not a multiple of four (the arrays' last vector),
    accumulate, and the fused <y, w>."""
    torch = torch_cuda
    dtype, tdt = (np.float64, torch.float64) if tag == "f64" else (np.float32, torch.float32)
    cases = [(1, 6000, 20, 72), (2, 3000, 1, 100), (3, 20000, 5, 28), (4, 4099, 0, 40), (5, 900, 60, 125)]
    for seed, rows, lo, hi in cases:
        rng = np.random.default_rng(100 * seed + vectors)
        v_rule = vectors
        lens = rng.integers(lo, hi + 1, size=rows)
        lens[rows // 3:rows // 3 + 300] = 0
        lens[-1] = hi
        if (int(lens.sum()) % 4) == 0:
            lens[0] += 1
        cols = rows + 33
        Ap, Aj, Ax = _random_csr(rng, rows, cols, lens, dtype)
        nnz = int(Ap[-1])
        mean, longest = nnz / rows, int(lens.max())
        if v_rule == 0:
            v_rule = 4 if mean >= 20 else 2 if mean >= 8 else 1
            while v_rule < 4 and 2 * (longest + 3) > 256 * v_rule:
                v_rule *= 2
        x = rng.standard_normal(cols).astype(dtype)
        y0 = rng.standard_normal(rows).astype(dtype)
        dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
        cfg = cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=vectors)
        if 2 * (longest + 3) > 256 * v_rule:  # the longest row takes more than half a tile: refused, by name
            with pytest.raises(cmi.CmiError):
                cmi.Plan(cmi.FORMAT_CSR, tdt, rows, cols, nnz, dAp, cfg)
            continue
        plan = cmi.Plan(cmi.FORMAT_CSR, tdt, rows, cols, nnz, dAp, cfg)
        c = plan.config()
        assert (c.kernel, c.items_per_thread, c.rows_per_block) == (cmi.CSR_STREAM_WAVEV, v_rule, 0), (seed, c)
        assert plan.info()["storage_order_sums"] is True
        want, want_acc = orc.spmv_csr(Ap, Aj, Ax, x), orc.spmv_csr(Ap, Aj, Ax, x, y0)
        y = torch.full((rows,), 9.0, dtype=tdt, device="cuda")
        cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y)
        assert np.array_equal(y.cpu().numpy(), want), (seed, vectors)
        y = dev(y0, torch)
        cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y, accumulate=True)
        assert np.array_equal(y.cpu().numpy(), want_acc), (seed, vectors, "accumulate")
        w = rng.standard_normal(rows).astype(dtype)
        res = torch.zeros(1, dtype=torch.float64, device="cuda")
        y = torch.zeros(rows, dtype=tdt, device="cuda")
        cmi.spmv_csr_dot(rows, cols, dAp, dAj, dAx, dx, y, dev(w, torch), res, cmi.blas_workspace(), plan=plan)
        assert np.array_equal(y.cpu().numpy(), want), (seed, vectors, "dot")
        ref = float(np.dot(want.astype(np.float64), w.astype(np.float64)))
        assert abs(float(res) - ref) <= 1e-12 * float(np.abs(want.astype(np.float64) * w).sum()) + (0 if tag == "f64" else 1e-6 * abs(ref)), seed
        # every XCD dealing and cache policy the plan may carry
        for pol, swz in ((0, 0), (1, 1), (3, 4), (2, 32)):
            p2 = cmi.Plan(cmi.FORMAT_CSR, tdt, rows, cols, nnz, dAp, cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=vectors, nontemporal=pol, xcd_swizzle=swz))
            y = torch.full((rows,), 3.0, dtype=tdt, device="cuda")
            cmi.spmv_csr_plan(p2, dAp, dAj, dAx, dx, y)
            assert np.array_equal(y.cpu().numpy(), want), (seed, vectors, pol, swz)
    # a row of 512+ entries: not this kernel's matrix
    lens = np.full(500, 30)
    lens[7] = 600
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    with pytest.raises(cmi.CmiError):
        cmi.Plan(cmi.FORMAT_CSR, tdt, 500, 700, int(Ap[-1]), dev(Ap, torch), cmi.Config(kernel=cmi.CSR_STREAM_WAVEV))
    # without a plan the kernel cannot run
    with pytest.raises(cmi.CmiError):
        Ap, Aj, Ax = _random_csr(np.random.default_rng(0), 100, 100, np.full(100, 20), dtype)
        y = torch.zeros(100, dtype=tdt, device="cuda")
        cmi.spmv_csr(100, 100, dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), torch.zeros(100, dtype=tdt, device="cuda"), y, cfg=cmi.Config(kernel=cmi.CSR_STREAM_WAVEV))


def test_plan_validate_and_row_offset_check(cmi, torch_cuda, orc):
    """The plan contract of include/cusp_mi355x.h: arrays a plan was made from must not change in place; cmi_plan_validate tells."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    rows, cols = 3000, 3100
    lens = rng.integers(0, 9, size=rows)
    Ap, Aj, Ax = _random_csr(rng, rows, cols, lens, np.float64)
    nnz = int(Ap[-1])
    dAp, dAj = dev(Ap, torch), dev(Aj, torch)

    def valid(plan, index, columns=None):
        return int(plan.validate(index, columns))

    plan = cmi.Plan(cmi.FORMAT_CSR, torch.float64, rows, cols, nnz, dAp)
    assert valid(plan, dAp) == 1
    # the same lengths in another order: same sizes, same ends, same multiset of offsets differences -- the checksum is order-sensitive
    lens2 = lens.copy()
    i = int(np.argmax(lens[:-1] != lens[1:]))  # two neighbouring rows of different length, swapped
    lens2[i], lens2[i + 1] = lens[i + 1], lens[i]
    Ap2 = np.r_[0, np.cumsum(lens2)].astype(np.int32)
    assert Ap2[-1] == Ap[-1] and not np.array_equal(Ap2, Ap)
    dAp.copy_(dev(Ap2, torch))
    assert valid(plan, dAp) == 0
    dAp.copy_(dev(Ap, torch))
    assert valid(plan, dAp) == 1
    # sorted COO: the plan caches row offsets derived from the row indices -- an in-place edit is what the check is for
    Ai = orc.csr_row_indices(Ap)
    dAi = dev(Ai, torch)
    cplan = cmi.Plan(cmi.FORMAT_COO, torch.float64, rows, cols, nnz, dAi)
    assert valid(cplan, dAi) == 1
    Ai2 = Ai.copy()
    k = int(np.argmax(np.diff(Ai) > 0))  # first position where the row changes: move one entry to the next row
    Ai2[k] = Ai[k + 1]
    dAi.copy_(dev(Ai2, torch))
    assert valid(cplan, dAi) == 0
    # the 16-bit column copy depends on the columns too
    Aj_local = np.sort(np.clip(np.repeat(np.arange(rows), lens) + rng.integers(-30, 30, size=nnz), 0, cols - 1)).astype(np.int32)
    dAj2 = dev(Aj_local, torch)
    p16 = cmi.Plan.csr(torch.float64, rows, cols, dAp, dAj2, cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
    if p16.config().kernel == cmi.CSR_STREAM_C16:
        assert valid(p16, dAp, dAj2) == 1
        dAj2[5] += 1
        assert valid(p16, dAp, dAj2) == 0
    # ELL / DIA plans hold nothing derived from the arrays
    eplan = cmi.Plan(cmi.FORMAT_ELL, torch.float64, rows, cols, rows * 8, None)
    assert valid(eplan, dAp) == 1
    # row offsets that do not span [0, num_entries]: refused at creation (the plan-owned arrays are sized from num_entries)
    for bad in (nnz - 1, nnz + 5):
        with pytest.raises(cmi.CmiError) as ei:
            cmi.Plan(cmi.FORMAT_CSR, torch.float64, rows, cols, bad, dAp)
        assert "row offsets run from" in str(ei.value)
    Ap3 = Ap.copy()
    Ap3[0] = 1
    with pytest.raises(cmi.CmiError):
        cmi.Plan(cmi.FORMAT_CSR, torch.float64, rows, cols, nnz, dev(Ap3, torch))


def test_container_replans_after_in_place_structure_change(cmi, torch_cuda, orc):
    """a different matrix of the same shape and entry count written into the SAME tensors must not multiply through
    the old plan (a sorted-COO plan caches row offsets).  The Python containers key their plan on the tensors' version counters."""
    torch = torch_cuda
    rng = np.random.default_rng(11)
    rows = cols = 2000
    lensA = rng.integers(1, 8, size=rows)
    lensB = lensA[::-1].copy()
    ApA, AjA, AxA = _random_csr(rng, rows, cols, lensA, np.float64)
    ApB, AjB, AxB = _random_csr(rng, rows, cols, lensB, np.float64)
    assert ApA[-1] == ApB[-1]
    nnz = int(ApA[-1])
    x = rng.standard_normal(cols)
    dx = dev(x, torch)
    C = cmi.CooMatrix(rows, cols, nnz, dev(orc.csr_row_indices(ApA), torch), dev(AjA, torch), dev(AxA, torch))
    y = torch.zeros(rows, dtype=torch.float64, device="cuda")
    cmi.multiply(C, dx, y)
    assert np.array_equal(y.cpu().numpy(), orc.spmv_csr(ApA, AjA, AxA, x))
    C.row_indices.copy_(dev(orc.csr_row_indices(ApB), torch))
    C.column_indices.copy_(dev(AjB, torch))
    C.values.copy_(dev(AxB, torch))
    cmi.multiply(C, dx, y)
    assert np.array_equal(y.cpu().numpy(), orc.spmv_csr(ApB, AjB, AxB, x)), "the COO container multiplied through a stale plan"
    A = cmi.CsrMatrix(rows, cols, nnz, dev(ApA, torch), dev(AjA, torch), dev(AxA, torch))
    cmi.multiply(A, dx, y)
    A.row_offsets.copy_(dev(ApB, torch))
    A.column_indices.copy_(dev(AjB, torch))
    A.values.copy_(dev(AxB, torch))
    cmi.multiply(A, dx, y)
    assert np.array_equal(y.cpu().numpy(), orc.spmv_csr(ApB, AjB, AxB, x))


def test_fold_handoff_stress(cmi, torch_cuda):
    """the multi-workgroup fold of a long partial list (dot_fold_final_kernel) hands its chunk sums to the last
    workgroup through an agent-scope release / acquire pair now; 3000 repetitions at a size that needs every chunk, each compared
    with the first -- the fold is a fixed tree, so every repetition must return the same bits -- and with a host sum."""
    torch = torch_cuda
    n = 9_000_017  # fused update kernels leave ~17 600 partials here: 18 folding workgroups
    rng = np.random.default_rng(3)
    r = dev(rng.standard_normal(n), torch)
    yv = dev(rng.standard_normal(n), torch)
    ws = cmi.blas_workspace()
    rz = torch.tensor([0.0], dtype=torch.float64, device="cuda")  # alpha = 0: r stays what it is, <r, r> is recomputed every time
    yp = torch.tensor([1.0], dtype=torch.float64, device="cuda")
    rr = torch.zeros(1, dtype=torch.float64, device="cuda")
    lib = cmi.lib()
    first = None
    vals = []
    for it in range(3000):
        cmi.check(lib.cmi_cg_update_f64(n, ctypes.c_void_p(rz.data_ptr()), ctypes.c_void_p(yp.data_ptr()), None, ctypes.c_void_p(yv.data_ptr()), None,
                                        ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(rr.data_ptr()), None, ctypes.c_void_p(ws.data_ptr()),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        if it % 50 == 0 or it > 2900:
            vals.append(float(rr))
    assert len(set(vals)) == 1, f"the fold returned {len(set(vals))} different values"
    host = float((r.double() * r.double()).sum())
    assert abs(vals[0] - host) <= 1e-10 * host
