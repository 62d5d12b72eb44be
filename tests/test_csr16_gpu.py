"""CMI_CSR_STREAM_C16 (csrc/spmv_csr16.hip): the opt-in plan that keeps a 16-bit copy of the column indices.  Same products
and same storage-order sums as csr_stream, so every result must equal the host loop's
(/root/reference cusp/system/detail/sequential/multiply/csr_spmv.h:56-73, restated in oracle/) BIT FOR BIT; the copy is
granted only when every row tile spans < 65536 columns and fits one LDS pass -- otherwise the plan says CMI_CSR_STREAM."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch


def dev(a, torch):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def banded(rng, rows, cols, band, max_len, dtype):
    lens = rng.integers(0, max_len + 1, size=rows)
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    parts = []
    for i, l in enumerate(lens):
        lo = max(0, min(cols - 1, i * cols // rows) - band)
        hi = min(cols, lo + 2 * band + 1)
        parts.append(np.sort(rng.choice(np.arange(lo, hi), size=min(l, hi - lo), replace=False)))
        lens[i] = len(parts[-1])
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    Aj = np.concatenate(parts + [np.zeros(0, np.int64)]).astype(np.int32)
    Ax = rng.standard_normal(len(Aj)).astype(dtype)
    return Ap, Aj, Ax


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_c16_poisson_every_shape_bit_exact(cmi, torch_cuda, orc, tag):
    torch = torch_cuda
    dtype = np.float64 if tag == "f64" else np.float32
    m, n = 301, 199
    N = m * n
    Ap, Aj, Ax = orc.poisson5pt_csr(m, n)
    Ax = Ax.astype(dtype)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(N).astype(dtype)
    y0 = rng.standard_normal(N).astype(dtype)
    want, want_acc = orc.spmv_csr(Ap, Aj, Ax, x), orc.spmv_csr(Ap, Aj, Ax, x, y0)
    dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
    # the table's shape
    plan = cmi.Plan.csr(dx.dtype, N, N, dAp, dAj, cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
    assert plan.config().kernel == cmi.CSR_STREAM_C16 and plan.info()["storage_order_sums"] is True
    c0 = plan.config()  # stencil rows + the table's shape: the copy is tiled per wave (64 rows, 5 entries per lane) for the wave-tile kernel
    assert (c0.block_size, c0.rows_per_block, c0.items_per_thread) == (256, 64, 5)
    y = torch.full((N,), 10.0, dtype=dx.dtype, device="cuda")
    cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y)
    assert np.array_equal(y.cpu().numpy(), want)
    # explicit shapes: block x vectors per lane x rows per tile x cache policy x XCD dealing
    for blk, ipt, nt, swz in itertools.product((128, 256, 512), (1, 2, 4), (0, 3, 4, 7), (0, 1, 32)):  # (4: lane-strided entry streams)
        fit = (blk * ipt * 4 - 3) // 5
        for rpb in {min(blk, fit), min(blk, max(1, fit // 2)), 1 if blk == 128 and ipt == 1 else min(blk, 64)}:
            cfg = cmi.Config(kernel=cmi.CSR_STREAM_C16, block_size=blk, items_per_thread=ipt, rows_per_block=rpb, nontemporal=nt, xcd_swizzle=swz)
            p = cmi.Plan.csr(dx.dtype, N, N, dAp, dAj, cfg=cfg)
            assert p.config().kernel == cmi.CSR_STREAM_C16, (blk, ipt, rpb)
            assert p.config().rows_per_block == rpb  # an explicit shape is kept
            y = torch.full((N,), 10.0, dtype=dx.dtype, device="cuda")
            cmi.spmv_csr_plan(p, dAp, dAj, dAx, dx, y)
            assert np.array_equal(y.cpu().numpy(), want), (blk, ipt, rpb, nt, swz)
            y = dev(y0, torch)
            cmi.spmv_csr_plan(p, dAp, dAj, dAx, dx, y, accumulate=True)
            assert np.array_equal(y.cpu().numpy(), want_acc), (blk, ipt, rpb, nt, swz, "acc")
    # fused <y, w>
    w = rng.standard_normal(N).astype(dtype)
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    y = torch.zeros(N, dtype=dx.dtype, device="cuda")
    cmi.spmv_csr_dot(N, N, dAp, dAj, dAx, dx, y, dev(w, torch), res, cmi.blas_workspace(), plan=plan)
    assert np.array_equal(y.cpu().numpy(), want)
    ref = float(np.dot(want.astype(np.float64), w.astype(np.float64)))
    assert abs(float(res) - ref) <= 1e-12 * float(np.abs(want.astype(np.float64) * w).sum()) + (0 if tag == "f64" else 1e-6 * abs(ref))


@pytest.mark.parametrize("seed", range(8))
def test_c16_banded_irregular_matrices(cmi, torch_cuda, orc, seed):
    """rows of 0..max_len entries inside a band: granted where every tile qualifies (checked against the rule restated here),
    bit-exact either way, whatever nnz % 4 is; rectangular too."""
    torch = torch_cuda
    rng = np.random.default_rng(100 + seed)
    rows = int(rng.integers(500, 30000))
    cols = rows if seed % 2 == 0 else int(rows * 1.7)
    Ap, Aj, Ax = banded(rng, rows, cols, band=int(rng.integers(3, 4000)), max_len=int(rng.integers(1, 40)) if seed < 6 else 6, dtype=np.float64)
    x = rng.standard_normal(cols)
    want = orc.spmv_csr(Ap, Aj, Ax, x)
    dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
    plan = cmi.Plan.csr(torch.float64, rows, cols, dAp, dAj, cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
    # granted exactly when every tile of the shape the plan would use spans < 65536 columns and fits one LDS pass
    plain = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, rows, cols, len(Aj))  # the table's csr_stream shape (what the copy is tiled for)
    pass_entries = plain.block_size * plain.items_per_thread * 4
    rpb = plain.rows_per_block
    up = -(-rpb // 64) * 64
    if up <= plain.block_size and up * (len(Aj) / rows) + 3.0 <= pass_entries:
        rpb = up
    ok = plain.kernel == cmi.CSR_STREAM and plain.threads_per_row <= 1 and len(Aj) >= 4 and rpb <= plain.block_size
    for r0 in range(0, rows, rpb):
        a, b = int(Ap[r0]), int(Ap[min(r0 + rpb, rows)])
        if b - (a & ~3) > pass_entries or (b > a and int(Aj[a:b].max()) - int(Aj[a:b].min()) > 65535):
            ok = False
    assert (plan.config().kernel == cmi.CSR_STREAM_C16) == ok, (plan.config(), plain, rpb)
    if not ok and plan.config().kernel == cmi.CSR_STREAM:
        assert plan.config().rows_per_block == plain.rows_per_block  # refused: csr_stream exactly as tuned (or the plan's wave tiles)
    y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
    cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y)
    assert np.array_equal(y.cpu().numpy(), want)


def test_c16_is_refused_where_it_does_not_fit(cmi, torch_cuda, orc):
    torch = torch_cuda
    rng = np.random.default_rng(8)
    # (a) columns scattered over 300000: a tile spans more than 65535 columns
    rows, cols = 4000, 300000
    lens = rng.integers(1, 9, size=rows)
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    Aj = np.concatenate([np.sort(rng.choice(cols, size=l, replace=False)) for l in lens]).astype(np.int32)
    Ax = rng.standard_normal(len(Aj))
    x = rng.standard_normal(cols)
    d = [dev(a, torch) for a in (Ap, Aj, Ax, x)]
    plan = cmi.Plan.csr(torch.float64, rows, cols, d[0], d[1], cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
    assert plan.config().kernel == cmi.CSR_STREAM  # not granted: csr_stream itself
    y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
    cmi.spmv_csr_plan(plan, *d[:3], d[3], y)
    assert np.array_equal(y.cpu().numpy(), orc.spmv_csr(Ap, Aj, Ax, x))
    # (b) one row longer than the LDS pass
    rows, cols = 3000, 5000
    lens = rng.integers(1, 6, size=rows)
    lens[1234] = 4500
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    Aj = np.concatenate([np.sort(rng.choice(cols, size=l, replace=False)) for l in lens]).astype(np.int32)
    Ax = rng.standard_normal(len(Aj))
    x = rng.standard_normal(cols)
    d = [dev(a, torch) for a in (Ap, Aj, Ax, x)]
    plan = cmi.Plan.csr(torch.float64, rows, cols, d[0], d[1], cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
    assert plan.config().kernel != cmi.CSR_STREAM_C16
    y = torch.full((rows,), 10.0, dtype=torch.float64, device="cuda")
    cmi.spmv_csr_plan(plan, *d[:3], d[3], y)
    got, want = y.cpu().numpy(), orc.spmv_csr(Ap, Aj, Ax, x)
    bound = orc.spmv_csr(Ap, Aj, np.abs(Ax), np.abs(x))
    assert np.all(np.abs(got - want) <= 1e-6 * bound + 1e-30)
    # (c) fewer than four entries; an empty matrix
    Ap3, Aj3, Ax3 = np.array([0, 1, 1, 3], np.int32), np.array([2, 0, 1], np.int32), np.array([1.5, -2.0, 4.0])
    d = [dev(a, torch) for a in (Ap3, Aj3, Ax3, np.array([1.0, 2.0, 3.0]))]
    plan = cmi.Plan.csr(torch.float64, 3, 3, d[0], d[1], cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
    assert plan.config().kernel != cmi.CSR_STREAM_C16
    y = torch.full((3,), 10.0, dtype=torch.float64, device="cuda")
    cmi.spmv_csr_plan(plan, *d[:3], d[3], y)
    assert y.cpu().numpy().tolist() == [4.5, 0.0, 6.0]
    # (d) it cannot be asked for without a plan, without the columns, or per table entry
    with pytest.raises(cmi.CmiError):
        cmi.spmv_csr(3, 3, d[0], d[1], d[2], d[3], y, cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
    with pytest.raises(cmi.CmiError):
        cmi.Plan(cmi.FORMAT_CSR, torch.float64, 3, 3, 3, d[0], cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
    with pytest.raises(cmi.CmiError):
        cmi.tuning_set(cmi.FORMAT_CSR, cmi.F64, 5.0, cmi.Config(kernel=cmi.CSR_STREAM_C16))


def test_c16_as_the_process_default_and_per_matrix(cmi, torch_cuda, orc):
    torch = torch_cuda
    A = cmi.poisson5pt(120, 90, "csr")
    N = A.num_rows
    x = cmi.fill_x(N).cuda()
    y = torch.empty(N, dtype=torch.float64, device="cuda")
    Ap, Aj, Ax = (t.cpu().numpy() for t in (A.row_offsets, A.column_indices, A.values))
    want = orc.spmv_csr(Ap, Aj, Ax, x.cpu().numpy())
    assert cmi.get_index_compression() is False
    cmi.multiply(A, x, y)
    plain_kernel = A.plan().config().kernel      # (the 5-point rows: the wave-tile kernel; csr_stream under $CMI_CSR_WAVE=0)
    assert plain_kernel in (cmi.CSR_STREAM, cmi.CSR_STREAM_WAVE)
    try:
        cmi.set_index_compression(True)
        cmi.multiply(A, x, y)                       # the default changed: the matrix re-plans
        assert A.plan().config().kernel == cmi.CSR_STREAM_C16
        assert np.array_equal(y.cpu().numpy(), want)
    finally:
        cmi.set_index_compression(False)
    cmi.multiply(A, x, y)
    assert A.plan().config().kernel == plain_kernel
    A.plan(compress=True)                           # per matrix, sticky
    y.fill_(10.0)
    cmi.multiply(A, x, y)
    assert A.plan().config().kernel == cmi.CSR_STREAM_C16
    assert np.array_equal(y.cpu().numpy(), want)
    # CG on the compressed matrix: same iterates as on the plain one (same products, same sums; the fused dot's tiling may
    # differ, so the residual norms agree to rounding, not to the bit)
    b = torch.ones(N, dtype=torch.float64, device="cuda")
    x1, x2 = torch.zeros_like(b), torch.zeros_like(b)
    h1 = cmi.krylov.cg(A, x1, b, iteration_limit=60, relative_tolerance=1e-10)
    A2 = cmi.poisson5pt(120, 90, "csr")
    h2 = cmi.krylov.cg(A2, x2, b, iteration_limit=60, relative_tolerance=1e-10)
    assert A.plan().config().kernel == cmi.CSR_STREAM_C16 and A2.plan().config().kernel == plain_kernel
    assert len(h1.residuals) == len(h2.residuals)
    assert np.allclose(h1.residuals, h2.residuals, rtol=1e-9, atol=0)
    assert torch.allclose(x1, x2, rtol=1e-9, atol=1e-12)


def test_c16_full_size_matches_the_plain_kernel(cmi, torch_cuda):
    """BASELINE configs[1]'s matrix: the compressed plan's y equals the plain plan's bit for bit; 100 MB of extra HBM."""
    torch = torch_cuda
    A = cmi.poisson5pt(3162, 3162, "csr")
    N = A.num_rows
    x = cmi.fill_x(N, device="cuda")
    y0 = torch.full((N,), 10.0, dtype=torch.float64, device="cuda")
    y1 = torch.full((N,), -10.0, dtype=torch.float64, device="cuda")
    cmi.multiply(A, x, y0)
    free0 = torch.cuda.mem_get_info()[0]
    p = cmi.Plan.csr(torch.float64, N, N, A.row_offsets, A.column_indices, cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
    used = free0 - torch.cuda.mem_get_info()[0]
    assert p.config().kernel == cmi.CSR_STREAM_C16
    assert used <= 2 * A.num_entries + (8 << 20)
    cmi.spmv_csr_plan(p, A.row_offsets, A.column_indices, A.values, x, y1)
    assert torch.equal(y0, y1)
