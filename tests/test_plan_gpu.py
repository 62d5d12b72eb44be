"""cmi_plan, the sorted-COO table key and the tuned HYB split rule, through the C-ABI on an MI355X."""
import ctypes
import itertools
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


def test_plan_lifecycle_and_errors(cmi, torch_cuda, orc):
    torch = torch_cuda
    Ap, Aj, Ax = orc.poisson5pt_csr(37, 29)
    n = 37 * 29
    dAp, dAj, dAx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch)
    x = orc.fill_x(n)
    want = orc.spmv_csr(Ap, Aj, Ax, x)
    plan = cmi.Plan(cmi.FORMAT_CSR, torch.float64, n, n, len(Aj), dAp)
    info = plan.info()
    assert info == {"max_row_length": 5, "entries_in_long_rows": 0, "coo_sorted": None, "storage_order_sums": True}
    table = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, n, n, len(Aj))
    # the profile (every row 3..5 entries, mean within 7 % of the longest) turns the table's csr_stream entry into the wave-tile
    # kernel: 64 rows per wave, 5 entries per lane; cache policy and XCD dealing stay the table's
    c = plan.config()
    assert (c.kernel, c.block_size, c.rows_per_block, c.items_per_thread) == (cmi.CSR_STREAM_WAVE, 256, 256, 5)
    assert table.kernel == cmi.CSR_STREAM and (c.nontemporal, c.xcd_swizzle) == (table.nontemporal & 3, table.xcd_swizzle)
    y = torch.full((n,), 10.0, dtype=torch.float64, device="cuda")
    cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dev(x, torch), y)
    assert np.array_equal(y.cpu().numpy(), want)
    # a plan made with an explicit config keeps it
    cfg = cmi.Config(kernel=cmi.CSR_VECTOR, threads_per_row=4)
    pv = cmi.Plan(cmi.FORMAT_CSR, torch.float64, n, n, len(Aj), dAp, cfg)
    assert pv.config().kernel == cmi.CSR_VECTOR and not pv.info()["storage_order_sums"]
    # wrong value type / format for the entry point
    L = cmi.lib()
    yf = torch.zeros(n, dtype=torch.float32, device="cuda")
    st = L.cmi_spmv_csr_plan_f32(plan.handle, ctypes.c_void_p(dAp.data_ptr()), ctypes.c_void_p(dAj.data_ptr()), ctypes.c_void_p(yf.data_ptr()),
                                 ctypes.c_void_p(yf.data_ptr()), ctypes.c_void_p(yf.data_ptr()), 0, None)
    assert st == 1  # CMI_ERROR_INVALID_VALUE
    st = L.cmi_spmv_coo_plan_f64(plan.handle, ctypes.c_void_p(dAp.data_ptr()), ctypes.c_void_p(dAj.data_ptr()), ctypes.c_void_p(dAx.data_ptr()),
                                 ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(y.data_ptr()), 0, None)
    assert st == 1
    with pytest.raises(cmi.CmiError):
        cmi.Plan(cmi.FORMAT_CSR, torch.float64, n, n, len(Aj), None)   # CSR needs its row offsets
    with pytest.raises(cmi.CmiError):
        cmi.Plan(7, torch.float64, n, n, len(Aj), dAp)
    # ELL / DIA / HYB: the resolved launch shape only
    pe = cmi.Plan(cmi.FORMAT_ELL, torch.float64, n, n, 5 * n)
    assert pe.config().kernel == cmi.ELL_ROW and pe.info()["storage_order_sums"] and pe.info()["max_row_length"] == -1
    assert cmi.Plan(cmi.FORMAT_DIA, torch.float32, n, n, 5 * n).config().kernel == cmi.DIA_ROW
    # an empty matrix plans (and multiplies) without touching the device arrays
    p0 = cmi.Plan(cmi.FORMAT_CSR, torch.float64, 0, 0, 0, None)
    assert p0.info()["max_row_length"] == -1


def test_coo_plan_sorted_and_unsorted(cmi, torch_cuda, orc, golden_irregular):
    torch, g = torch_cuda, golden_irregular
    rows, cols = int(g["rows"]), int(g["cols"])
    Ap, Aj, Ax, x = (g[f"f64_{k}"] for k in ("Ap", "Aj", "Ax", "x"))
    Ai = orc.csr_row_indices(Ap)
    want = orc.spmv_coo(rows, Ai, Aj, Ax, x)
    dx = dev(x, torch)
    ps = cmi.Plan(cmi.FORMAT_COO, torch.float64, rows, cols, len(Aj), dev(Ai, torch))
    # sorted entries: the plan builds the row offsets the row indices imply and runs what a CSR plan of them runs (the row
    # indices are never read again: 12 instead of 16 bytes per entry) -- same kernel choice, same result class
    assert ps.info()["coo_sorted"] is True
    pc = cmi.Plan(cmi.FORMAT_CSR, torch.float64, rows, cols, len(Aj), dev(Ap, torch))
    assert ps.config().as_dict() == pc.config().as_dict()
    assert ps.info()["storage_order_sums"] == pc.info()["storage_order_sums"]
    # with $CMI_COO_PLAN_OFFSETS=0 the plan keeps the COO kernels: the table's sorted-COO key (the tile kernel) -- unless a row
    # holds more than one tile of entries (this fixture has a 5000-entry row: its tail would be walked serially)
    os.environ["CMI_COO_PLAN_OFFSETS"] = "0"
    try:
        pk = cmi.Plan(cmi.FORMAT_COO, torch.float64, rows, cols, len(Aj), dev(Ai, torch))
        key = cmi.FORMAT_COO if (np.diff(Ap) > 1024).any() else cmi.TABLE_COO_SORTED
        assert pk.config().as_dict() == cmi.tuning_select(key, cmi.F64, rows, cols, len(Aj)).as_dict()
        assert pk.info()["storage_order_sums"] == (pk.config().kernel == cmi.COO_TILE)
        Ap_s, Aj_s, Ax_s = orc.poisson5pt_csr(37, 29)   # short rows only: the sorted-COO key
        Ai_s = orc.csr_row_indices(Ap_s)
        pss = cmi.Plan(cmi.FORMAT_COO, torch.float64, 37 * 29, 37 * 29, len(Aj_s), dev(Ai_s, torch))
        assert pss.config().as_dict() == cmi.tuning_select(cmi.TABLE_COO_SORTED, cmi.F64, 37 * 29, 37 * 29, len(Aj_s)).as_dict()
    finally:
        os.environ.pop("CMI_COO_PLAN_OFFSETS", None)
    y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
    cmi.spmv_coo_plan(ps, dev(Ai, torch), dev(Aj, torch), dev(Ax, torch), dx, y)
    bound0 = orc.spmv_csr(Ap, Aj, np.abs(Ax), np.abs(x))
    assert np.all(np.abs(y.cpu().numpy() - want) <= 1e-6 * np.maximum(bound0, 1e-300))
    # asked for explicitly, the tile kernel gives the host loop's bits, and the plan says so
    pt = cmi.Plan(cmi.FORMAT_COO, torch.float64, rows, cols, len(Aj), dev(Ai, torch), cmi.Config(kernel=cmi.COO_TILE, xcd_swizzle=8))
    assert pt.config().kernel == cmi.COO_TILE and pt.info()["storage_order_sums"]
    y.fill_(10.0)
    cmi.spmv_coo_plan(pt, dev(Ai, torch), dev(Aj, torch), dev(Ax, torch), dx, y)
    assert np.array_equal(y.cpu().numpy(), want)
    # shuffled entries: the plan says so and runs an order-agnostic kernel (tolerance class)
    perm = np.random.default_rng(8).permutation(len(Aj))
    dAi, dAj, dAx = dev(Ai[perm], torch), dev(Aj[perm], torch), dev(Ax[perm], torch)
    pu = cmi.Plan(cmi.FORMAT_COO, torch.float64, rows, cols, len(Aj), dAi)
    assert pu.info()["coo_sorted"] is False and pu.config().kernel in (cmi.COO_LANE4, cmi.COO_SEGMENTED)
    assert not pu.info()["storage_order_sums"]
    y.fill_(10.0)
    cmi.spmv_coo_plan(pu, dAi, dAj, dAx, dx, y)
    bound = orc.spmv_csr(Ap, Aj, np.abs(Ax), np.abs(x))
    assert np.all(np.abs(y.cpu().numpy() - want) <= 1e-6 * np.maximum(bound, 1e-300))
    # asking for the tile kernel on unsorted entries is refused where it is known
    with pytest.raises(cmi.CmiError):
        cmi.Plan(cmi.FORMAT_COO, torch.float64, rows, cols, len(Aj), dAi, cmi.Config(kernel=cmi.COO_TILE))
    # out-of-range row index: not "sorted"
    bad = Ai.copy()
    bad[-1] = rows
    assert cmi.Plan(cmi.FORMAT_COO, torch.float64, rows, cols, len(Aj), dev(bad, torch)).info()["coo_sorted"] is False
    # the plan-less table key never holds the tile kernel
    assert cmi.tuning_select(cmi.FORMAT_COO, cmi.F64, rows, cols, len(Aj)).kernel != cmi.COO_TILE
    with pytest.raises(cmi.CmiError):
        cmi.tuning_set(cmi.FORMAT_COO, cmi.F64, 5.0, cmi.Config(kernel=cmi.COO_TILE))


def test_containers_plan_once_and_replan_when_the_arrays_change(cmi, torch_cuda, orc):
    torch = torch_cuda
    A = cmi.poisson5pt(40, 30, "csr")
    x = cmi.fill_x(1200).cuda()
    y = torch.empty(1200, dtype=torch.float64, device="cuda")
    assert A._plan is None
    cmi.multiply(A, x, y)
    p1 = A._plan
    assert p1 is not None
    cmi.multiply(A, x, y)
    assert A._plan is p1                       # made once
    cmi.multiply(A, x, y, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
    assert A._plan is p1                       # an explicit config bypasses the plan, it does not replace it
    A.row_offsets = A.row_offsets.clone()      # new structure array: a new plan
    cmi.multiply(A, x, y)
    assert A._plan is not p1
    A.invalidate()
    assert A._plan is None
    # HYB: its own plan (COO part sorted by construction -> the one-launch kernel, the host loops' bits)
    H = cmi.convert(A, "hyb", num_entries_per_row=3)
    Ap, Aj, Ax = (t.cpu().numpy() for t in (A.row_offsets, A.column_indices, A.values))
    y.fill_(10.0)
    cmi.multiply(H, x, y)
    assert H._plan is not None and H._plan.info()["coo_sorted"] and H._plan.info()["storage_order_sums"]
    p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, 3)
    want = orc.spmv_hyb(1200, 3, p, hAj, hAx, cAi, cAj, cAx, x.cpu().numpy())
    assert np.array_equal(y.cpu().numpy(), want)
    hp = H._plan
    cmi.multiply(H, x, y)
    assert H._plan is hp
    H.invalidate()
    assert H._plan is None


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("shape", ["irregular", "heavy_tail", "one_entry_rows", "desert", "width0"])
def test_hyb_plan_one_launch(cmi, torch_cuda, orc, tag, shape):
    """cmi_spmv_hyb_plan_*: with the COO part sorted by row ONE kernel finishes every row -- ELL slots, then the row's COO
    entries, one accumulator (sequential/multiply/hyb_spmv.h:55-56 is that chain) -- bit for bit the host loops, overwrite and
    accumulate; tiles whose COO range needs several 256-entry chunks, rows whose run straddles chunks, tiles without entries.
    A COO part that is not sorted runs the two launches (1e-6 class); an empty one the ELL kernel."""
    torch = torch_cuda
    dtype = np.float64 if tag == "f64" else np.float32
    rng = np.random.default_rng(31)
    rows, cols = 3000, 2500
    if shape == "irregular":
        lens, width = rng.integers(0, 12, size=rows), 5
    elif shape == "heavy_tail":  # rows of up to 2000 entries: a tile's COO range spans many chunks, one row several
        lens = rng.integers(1, 6, size=rows)
        lens[[5, 300, 301, 1500, 2999]] = [2000, 700, 900, 1999, 1200]
        width = 3
    elif shape == "one_entry_rows":
        lens, width = np.full(rows, 5), 4
    elif shape == "desert":  # long runs of rows without any COO entry (and without any entry at all)
        lens = np.zeros(rows, np.int64)
        lens[::97] = 9
        lens[1000:1010] = 40
        width = 2
    else:
        lens, width = rng.integers(0, 7, size=rows), 0
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    Aj = np.concatenate([np.sort(rng.choice(cols, size=l, replace=False)) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    Ax = rng.standard_normal(len(Aj)).astype(dtype)
    x = rng.standard_normal(cols).astype(dtype)
    y0 = rng.standard_normal(rows).astype(dtype)
    p, hAj, hAx, cAi, cAj, cAx = orc.csr_to_hyb(Ap, Aj, Ax, width)
    want = orc.spmv_hyb(rows, width, p, hAj, hAx, cAi, cAj, cAx, x)
    want_acc = orc.spmv_hyb(rows, width, p, hAj, hAx, cAi, cAj, cAx, x, y0)
    d = [torch.from_numpy(a).cuda() for a in (hAj, hAx, cAi, cAj, cAx)]
    dx = torch.from_numpy(x).cuda()
    plan = cmi.Plan.hyb(dx.dtype, rows, cols, width, d[2])
    assert plan.info()["coo_sorted"] is True
    bound = orc.spmv_hyb(rows, width, p, hAj, np.abs(hAx), cAi, cAj, np.abs(cAx), np.abs(x)) + np.abs(y0)
    tol = (1e-6 if tag == "f64" else 1e-4) * bound + 1e-30
    # one launch (forced: also where the plan would not choose it -- a tile's COO range then takes many 256-entry chunks), two
    # launches (forced: ELL kernel + COO tile kernel accumulating: the same chain per row), and the plan's own choice by the
    # COO part's weight: every one the host loops' bits
    for force, (swz, nt) in itertools.product(("1", "0", None), ((0, 0), (1, 1), (3, 2), (64, 3))):
        old = os.environ.pop("CMI_HYB_ONE_LAUNCH", None)
        try:
            if force is not None:
                os.environ["CMI_HYB_ONE_LAUNCH"] = force
            pl = cmi.Plan.hyb(dx.dtype, rows, cols, width, d[2], cfg_ell=cmi.Config(kernel=cmi.ELL_ROW, threads_per_row=1, xcd_swizzle=swz, nontemporal=nt))
        finally:
            os.environ.pop("CMI_HYB_ONE_LAUNCH", None)
            if old is not None:
                os.environ["CMI_HYB_ONE_LAUNCH"] = old
        exact = pl.info()["storage_order_sums"]  # two launches: only when the table's sorted-COO kernel is the tile kernel
        if force == "1" and len(cAi):
            assert exact is True, (shape, force)
        if force is not None and len(cAi):
            assert pl.hyb_launches() == (1 if force == "1" else 2), (shape, force)
        elif len(cAi):  # the weight rule, restated
            per_tile = np.bincount(cAi // 256, minlength=(rows + 255) // 256)
            assert pl.hyb_launches() == (1 if len(cAi) <= 3.0 * rows and per_tile.max() <= 4096 else 2), (shape, len(cAi), per_tile.max())
        y = torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
        cmi.spmv_hyb_plan(pl, p, *d, dx, y)
        ya = torch.from_numpy(y0).cuda()
        cmi.spmv_hyb_plan(pl, p, *d, dx, ya, accumulate=True)
        if exact:
            assert np.array_equal(y.cpu().numpy(), want), (shape, force, swz, nt)
            assert np.array_equal(ya.cpu().numpy(), want_acc), (shape, force, swz, nt, "acc")
        else:
            assert np.all(np.abs(y.cpu().numpy() - want) <= tol) and np.all(np.abs(ya.cpu().numpy() - want_acc) <= tol), (shape, force, swz, nt)
    # <y, w> in the same pass (the CG step): fused in the one-launch kernel, a separate dot behind the two launches
    wv = rng.standard_normal(rows).astype(dtype)
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    for force in ("1", "0"):
        os.environ["CMI_HYB_ONE_LAUNCH"] = force
        try:
            pl = cmi.Plan.hyb(dx.dtype, rows, cols, width, d[2], cfg_ell=cmi.Config(kernel=cmi.ELL_ROW, threads_per_row=1))
        finally:
            os.environ.pop("CMI_HYB_ONE_LAUNCH", None)
        args = cmi.binding.hyb_plan_args(pl, p, *d)
        y = torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
        res.fill_(float("nan"))
        cmi.binding.spmv_hyb_dot_plan_args(args, dx, y, torch.from_numpy(wv).cuda(), res, cmi.blas_workspace())
        yh = y.cpu().numpy()
        if pl.info()["storage_order_sums"]:
            assert np.array_equal(yh, want), (shape, force, "dot: y")
        else:
            assert np.all(np.abs(yh - want) <= tol), (shape, force, "dot: y")
        ref = float(np.dot(yh.astype(np.float64), wv.astype(np.float64)))
        assert abs(float(res) - ref) <= 1e-12 * float(np.abs(yh.astype(np.float64) * wv).sum()) + 1e-300, (shape, force, float(res), ref)
    # COO part shuffled: the plan sees it, two launches, sums re-associated
    if len(cAi) > 1:
        perm = rng.permutation(len(cAi))
        ds = [torch.from_numpy(a[perm].copy()).cuda() for a in (cAi, cAj, cAx)]
        pu = cmi.Plan.hyb(dx.dtype, rows, cols, width, ds[0])
        assert pu.info()["coo_sorted"] is False and pu.info()["storage_order_sums"] is False
        y = torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
        cmi.spmv_hyb_plan(pu, p, d[0], d[1], *ds, dx, y)
        assert np.all(np.abs(y.cpu().numpy() - want) <= tol)
    # empty COO part
    w2 = int(lens.max())
    p2, hAj2, hAx2, cAi2, cAj2, cAx2 = orc.csr_to_hyb(Ap, Aj, Ax, w2)
    assert len(cAi2) == 0
    d2 = [torch.from_numpy(a).cuda() for a in (hAj2, hAx2, cAi2, cAj2, cAx2)]
    pe = cmi.Plan.hyb(dx.dtype, rows, cols, w2, d2[2], cfg_ell=cmi.Config(kernel=cmi.ELL_ROW, threads_per_row=1))
    y = torch.full((rows,), 10.0, dtype=dx.dtype, device="cuda")
    cmi.spmv_hyb_plan(pe, p2, *d2, dx, y)
    assert np.array_equal(y.cpu().numpy(), orc.spmv_hyb(rows, w2, p2, hAj2, hAx2, cAi2, cAj2, cAx2, x))
    with pytest.raises(ValueError):
        cmi.spmv_hyb_plan(plan, p, *d, dx, y[:-1])
    with pytest.raises(cmi.CmiError):  # a HYB plan is not a COO plan
        cmi.check(cmi.lib().cmi_spmv_coo_plan_f64(plan.handle, None, None, None, None, None, 0, None))


def test_hyb_width_rule(cmi, torch_cuda, orc, golden_irregular):
    """cmi_hyb_entries_per_row: the reference rule kind with the reference's pair reproduces
    cusp::compute_optimal_entries_per_row (oracle restatement, pinned to the reference tests' examples by
    tests/test_oracle_golden.py); the launch-cost kind is the argmin of its model (numpy restatement here); the tuned rule
    comes from the table."""
    import os
    torch, g = torch_cuda, golden_irregular
    rng = np.random.default_rng(77)
    cases = [g["f64_Ap"]]
    for rows, lo, hi in ((5000, 0, 9), (20000, 1, 40), (300, 2, 3), (100000, 0, 4)):
        lens = rng.integers(lo, hi + 1, size=rows)
        lens[rng.integers(0, rows, size=3)] = 5000          # a few rows beyond the histogram's last bin
        cases.append(np.r_[0, np.cumsum(lens)].astype(np.int32))

    def cost_width(Ap, rs, th):
        lens = np.minimum(np.diff(Ap).astype(np.int64), 4096)
        N, mx = len(lens), int(lens.max())
        ks = np.arange(mx + 1)
        coo = np.array([np.maximum(lens - k, 0).sum() for k in ks], dtype=np.float64)
        cost = N * ks.astype(np.float64) + np.where(coo > 0, th + rs * coo, 0.0)
        return int(ks[cost == cost.min()].max())            # ties: the wider ELL part

    def cost2_width(Ap, rs, th, light):                     # two regimes: light COO part (<= 3 entries per row) = one launch
        lens = np.minimum(np.diff(Ap).astype(np.int64), 4096)
        N, mx = len(lens), int(lens.max())
        ks = np.arange(mx + 1)
        coo = np.array([np.maximum(lens - k, 0).sum() for k in ks], dtype=np.float64)
        cost = N * ks.astype(np.float64) + np.where(coo > 0, np.where(coo <= 3.0 * N, light * coo, th + rs * coo), 0.0)
        return int(ks[cost == cost.min()].max())

    shipped_rule = {dt: (cmi.tuning_hyb_rule(dt), cmi.tuning_hyb_light_speed(dt)) for dt in (cmi.F64, cmi.F32)}
    for Ap in cases:
        rows = len(Ap) - 1
        dAp = dev(Ap, torch)
        for rs, th, light in ((1.0, 2_000_000, 3.0), (1.1, 0, 1.5), (2.0, 50_000, 1.0)):
            cmi.tuning_set_hyb_light_speed(cmi.F64, light)
            got = cmi.hyb_entries_per_row(cmi.F64, rows, dAp, cmi.HYB_RULE_COST2, rs, th)
            assert got == cost2_width(Ap, rs, th, light), (rows, rs, th, light, got, cost2_width(Ap, rs, th, light))
        cmi.tuning_set_hyb_light_speed(cmi.F64, shipped_rule[cmi.F64][1])
        for rs, be in ((3.0, 4096), (3.0, 0), (1.5, 100), (10.0, 1), (1.0, 0)):
            got = cmi.hyb_entries_per_row(cmi.F64, rows, dAp, cmi.HYB_RULE_REFERENCE, rs, be)
            want = min(orc.optimal_entries_per_row(Ap, rs, be), 4096)
            assert got == want, (rows, rs, be, got, want)
        for rs, th in ((1.3, 5_000_000), (1.3, 0), (2.0, 1000), (1.0, 10**9), (4.0, 50_000)):
            got = cmi.hyb_entries_per_row(cmi.F64, rows, dAp, cmi.HYB_RULE_COST, rs, th)
            assert got == cost_width(Ap, rs, th), (rows, rs, th, got, cost_width(Ap, rs, th))
    # the tuned rule: what the table (or, with none, the reference's constants) says
    kind, rs, th = cmi.tuning_hyb_rule(cmi.F64)
    assert kind in (cmi.HYB_RULE_REFERENCE, cmi.HYB_RULE_COST, cmi.HYB_RULE_COST2) and rs > 0 and th >= 0
    Ap = cases[2]
    expect = (cost2_width(Ap, rs, th, cmi.tuning_hyb_light_speed(cmi.F64)) if kind == cmi.HYB_RULE_COST2 else
              cost_width(Ap, rs, th) if kind == cmi.HYB_RULE_COST else min(orc.optimal_entries_per_row(Ap, rs, th), 4096))
    assert cmi.hyb_entries_per_row(cmi.F64, len(Ap) - 1, dev(Ap, torch)) == expect
    # set / get / clear
    shipped = os.path.join(os.path.dirname(cmi.lib_path()), "..", "tuned", "gfx950.json")
    try:
        cmi.tuning_set_hyb_rule(cmi.F32, cmi.HYB_RULE_COST, 2.25, 123)
        assert cmi.tuning_hyb_rule(cmi.F32) == (cmi.HYB_RULE_COST, 2.25, 123)
        cmi.tuning_clear()
        assert cmi.tuning_hyb_rule(cmi.F32) == (cmi.HYB_RULE_REFERENCE, 3.0, 4096)
        # convert(..., "hyb") without a width uses the rule, and the result multiplies to the CSR result
        A = cmi.CsrMatrix(len(Ap) - 1, len(Ap) - 1, int(Ap[-1]), dev(Ap, torch),
                          dev(rng.integers(0, len(Ap) - 1, size=int(Ap[-1])).astype(np.int32), torch), dev(rng.standard_normal(int(Ap[-1])), torch))
        H = cmi.convert(A, "hyb")
        assert H.ell.num_entries_per_row == min(orc.optimal_entries_per_row(Ap, 3.0, 4096), 4096)   # table cleared above
        cmi.tuning_load(shipped)
        H2 = cmi.convert(A, "hyb")                                                                    # the shipped rule
        assert H2.ell.num_entries_per_row == expect
        x = dev(rng.standard_normal(len(Ap) - 1), torch)
        y0 = torch.empty(len(Ap) - 1, dtype=torch.float64, device="cuda")
        cmi.multiply(A, x, y0, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
        for M in (H, H2):
            y1 = torch.full_like(y0, 10.0)
            cmi.multiply(M, x, y1)
            assert torch.allclose(y1, y0, rtol=1e-9, atol=1e-9)
    finally:
        cmi.tuning_load(shipped)   # the rest of the session runs on the shipped table again


def test_plans_free_the_device_memory_they_own(cmi, torch_cuda, orc):
    """COO plans (row offsets + a CSR sub-plan), HYB plans (tile ranges, or a COO plan for the second launch) and 16-bit-column CSR
    plans own device memory: cmi_plan_destroy gives all of it back (500 create / destroy rounds leave free memory where it was)."""
    import gc
    torch = torch_cuda
    A = cmi.poisson5pt(400, 300, "csr")
    N, nnz = A.num_rows, A.num_entries
    C = cmi.convert(A, "coo")
    H1 = cmi.convert(A, "hyb", num_entries_per_row=4)     # light COO part: one launch (tile ranges)
    H0 = cmi.convert(A, "hyb", num_entries_per_row=1)     # heavy COO part: two launches (nested COO plan)

    def make_all():
        ps = [cmi.Plan(cmi.FORMAT_COO, torch.float64, N, N, nnz, C.row_indices),
              cmi.Plan.hyb(torch.float64, N, N, 4, H1.coo.row_indices),
              cmi.Plan.hyb(torch.float64, N, N, 1, H0.coo.row_indices),
              cmi.Plan.csr(torch.float64, N, N, A.row_offsets, A.column_indices, cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))]
        assert ps[1].hyb_launches() == 1 and ps[2].hyb_launches() == 2 and ps[3].config().kernel == cmi.CSR_STREAM_C16
        return ps

    ps = make_all()
    del ps
    gc.collect()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(500):
        ps = make_all()
        del ps
    gc.collect()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert abs(free1 - free0) <= 8 << 20, (free0, free1)   # one round owns ~1.2 MB: 500 leaked rounds would be 600 MB


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_csr_plan_selects_wave_tiles_for_stencil_rows(cmi, torch_cuda, orc, tag):
    """Rows that all have (nearly) the same short length: the plan runs CMI_CSR_STREAM_WAVE with as many entries per lane as the
    longest row has -- bit-exact, also accumulating, through the fused dot, for a sorted COO matrix (row offsets built by its
    plan) -- and keeps csr_stream where the lengths vary, where a caller names a kernel, and under $CMI_CSR_WAVE=0 (child process)."""
    torch = torch_cuda
    dtype = np.float64 if tag == "f64" else np.float32
    tdt = torch.float64 if tag == "f64" else torch.float32
    rng = np.random.default_rng(11)
    for m, n in ((257, 131), (64, 64), (1, 300), (300, 1)):
        N = m * n
        Ap, Aj, Ax = orc.poisson5pt_csr(m, n)
        Ax = (Ax * rng.standard_normal(len(Ax))).astype(dtype)
        x = rng.standard_normal(N).astype(dtype)
        y0 = rng.standard_normal(N).astype(dtype)
        want, want_acc = orc.spmv_csr(Ap, Aj, Ax, x), orc.spmv_csr(Ap, Aj, Ax, x, y0)
        dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
        plan = cmi.Plan(cmi.FORMAT_CSR, tdt, N, N, len(Aj), dAp)
        c = plan.config()
        longest, mean = int(np.diff(Ap).max()), len(Aj) / N
        if 2 <= longest <= 10 and mean >= 0.93 * longest:
            assert c.kernel == cmi.CSR_STREAM_WAVE and c.items_per_thread == longest and c.rows_per_block == c.block_size == 256, (m, n, c)
            assert plan.info()["storage_order_sums"] is True
        else:
            assert c.kernel == cmi.CSR_STREAM, (m, n, c)
        y = torch.full((N,), 9.0, dtype=tdt, device="cuda")
        cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y)
        assert np.array_equal(y.cpu().numpy(), want), (m, n)
        y = dev(y0, torch)
        cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y, accumulate=True)
        assert np.array_equal(y.cpu().numpy(), want_acc), (m, n)
        w = rng.standard_normal(N).astype(dtype)
        res = torch.zeros(1, dtype=torch.float64, device="cuda")
        y = torch.zeros(N, dtype=tdt, device="cuda")
        cmi.spmv_csr_dot(N, N, dAp, dAj, dAx, dx, y, dev(w, torch), res, cmi.blas_workspace(), plan=plan)
        assert np.array_equal(y.cpu().numpy(), want), (m, n)
        ref = float(np.dot(want.astype(np.float64), w.astype(np.float64)))
        assert abs(float(res) - ref) <= 1e-12 * float(np.abs(want.astype(np.float64) * w).sum()) + (0 if tag == "f64" else 1e-6 * abs(ref)), (m, n)
        # the opt-in 16-bit column copy on such rows is tiled per wave and read by the wave-tile kernel's twin
        p16 = cmi.Plan.csr(tdt, N, N, dAp, dAj, cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
        c16 = p16.config()
        assert c16.kernel == cmi.CSR_STREAM_C16
        if c.kernel == cmi.CSR_STREAM_WAVE:
            assert (c16.block_size, c16.rows_per_block, c16.items_per_thread) == (256, 64, longest), c16
        for acc, ref_y in ((False, want), (True, want_acc)):
            y = dev(y0, torch) if acc else torch.full((N,), 9.0, dtype=tdt, device="cuda")
            cmi.spmv_csr_plan(p16, dAp, dAj, dAx, dx, y, accumulate=acc)
            assert np.array_equal(y.cpu().numpy(), ref_y), (m, n, "c16", acc)
        res.fill_(float("nan"))
        y = torch.zeros(N, dtype=tdt, device="cuda")
        cmi.spmv_csr_dot(N, N, dAp, dAj, dAx, dx, y, dev(w, torch), res, cmi.blas_workspace(), plan=p16)
        assert np.array_equal(y.cpu().numpy(), want), (m, n, "c16 dot")
        assert abs(float(res) - ref) <= 1e-12 * float(np.abs(want.astype(np.float64) * w).sum()) + (0 if tag == "f64" else 1e-6 * abs(ref)), (m, n, "c16 dot")
        # a caller's kernel is kept
        explicit = cmi.Plan(cmi.FORMAT_CSR, tdt, N, N, len(Aj), dAp, cmi.Config(kernel=cmi.CSR_STREAM))
        assert explicit.config().kernel == cmi.CSR_STREAM
        # sorted COO: the plan's row offsets + the same CSR kernel
        Ai = orc.csr_row_indices(Ap)
        cplan = cmi.Plan(cmi.FORMAT_COO, tdt, N, N, len(Aj), dev(Ai, torch))
        assert cplan.config().kernel == c.kernel
        y = torch.full((N,), 9.0, dtype=tdt, device="cuda")
        cmi.spmv_coo_plan(cplan, dev(Ai, torch), dAj, dAx, dx, y)
        assert np.array_equal(y.cpu().numpy(), want), (m, n)
    # irregular short rows: csr_stream by default; ASKED FOR (kernel CMI_CSR_STREAM_WAVE, rows_per_block < 0) the same kernel runs on a
    # partition of the rows the plan builds (wave tile = the rows whose first entry falls into one quantum of 64 K - longest entries;
    # rows_per_block then reads 0) -- with empty rows, a stretch of 300 empty rows (more rows than lanes in one tile: another turn
    # of the row-sum loop) and a last row that ends exactly at the arrays' end
    for seed, rows, hi in ((1, 5000, 11), (2, 20000, 7), (3, 4096, 16), (4, 70000, 9)):
        r2 = np.random.default_rng(seed)
        lens = r2.integers(0, hi, size=rows)
        lens[1000:1300] = 0
        lens[-1] = hi - 1
        cols = rows + 17
        Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
        nnz = int(Ap[-1])
        Aj = r2.integers(0, cols, size=nnz).astype(np.int32)  # (unsorted, duplicates allowed)
        Ax = r2.standard_normal(nnz).astype(dtype)
        x = r2.standard_normal(cols).astype(dtype)
        y0 = r2.standard_normal(rows).astype(dtype)
        want, want_acc = orc.spmv_csr(Ap, Aj, Ax, x), orc.spmv_csr(Ap, Aj, Ax, x, y0)
        dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
        assert cmi.Plan(cmi.FORMAT_CSR, tdt, rows, cols, nnz, dAp).config().kernel == cmi.CSR_STREAM
        mean, longest = nnz / rows, int(lens.max())
        k = max(2, int(np.floor(mean + longest / 64.0)))
        assert 2 <= k <= 10 and longest <= 6.4 * k
        plan = cmi.Plan(cmi.FORMAT_CSR, tdt, rows, cols, nnz, dAp, cmi.Config(kernel=cmi.CSR_STREAM_WAVE, rows_per_block=-1))
        c = plan.config()
        assert (c.kernel, c.rows_per_block, c.items_per_thread) == (cmi.CSR_STREAM_WAVE, 0, k), (seed, c)
        assert plan.info()["storage_order_sums"] is True
        if seed == 2:  # entries per lane given by the caller
            p8 = cmi.Plan(cmi.FORMAT_CSR, tdt, rows, cols, nnz, dAp, cmi.Config(kernel=cmi.CSR_STREAM_WAVE, rows_per_block=-1, items_per_thread=8))
            assert p8.config().items_per_thread == 8
            y = torch.full((rows,), 9.0, dtype=tdt, device="cuda")
            cmi.spmv_csr_plan(p8, dAp, dAj, dAx, dx, y)
            assert np.array_equal(y.cpu().numpy(), want), "k 8"
        y = torch.full((rows,), 9.0, dtype=tdt, device="cuda")
        cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y)
        assert np.array_equal(y.cpu().numpy(), want), seed
        y = dev(y0, torch)
        cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y, accumulate=True)
        assert np.array_equal(y.cpu().numpy(), want_acc), seed
        w = r2.standard_normal(rows).astype(dtype)
        res = torch.zeros(1, dtype=torch.float64, device="cuda")
        y = torch.zeros(rows, dtype=tdt, device="cuda")
        cmi.spmv_csr_dot(rows, cols, dAp, dAj, dAx, dx, y, dev(w, torch), res, cmi.blas_workspace(), plan=plan)
        assert np.array_equal(y.cpu().numpy(), want), seed
        ref = float(np.dot(want.astype(np.float64), w.astype(np.float64)))
        assert abs(float(res) - ref) <= 1e-12 * float(np.abs(want.astype(np.float64) * w).sum()) + (0 if tag == "f64" else 1e-6 * abs(ref)), seed
        del plan
    # rows longer than 10: csr_stream; a partition cannot be asked for either where the rows are too long for it
    lens = np.full(4000, 11)
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    assert cmi.Plan(cmi.FORMAT_CSR, tdt, 4000, 4000, int(Ap[-1]), dev(Ap, torch)).config().kernel == cmi.CSR_STREAM
    with pytest.raises(cmi.CmiError):
        cmi.Plan(cmi.FORMAT_CSR, tdt, 4000, 4000, int(Ap[-1]), dev(Ap, torch), cmi.Config(kernel=cmi.CSR_STREAM_WAVE, rows_per_block=-1))
    if tag == "f64":
        import subprocess, sys
        code = ("import numpy as np, torch, cusp_autotuned_amd as cmi\n"
                "A = cmi.poisson5pt(100, 100, 'csr')\n"
                "print(A.plan().config().kernel)\n")
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, "CMI_CSR_WAVE": "0"},
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip().splitlines()[-1] == str(cmi.CSR_STREAM)
