"""Property-based parity (-m gpu): hypothesis draws the matrix SHAPE -- sizes down to 0 x n, row-length profiles (uniform, heavy tail, stretches of
empty rows, one giant row, FEM-like runs), value type, accumulate -- and every way the library can multiply it is compared with the CPU oracle
(the reference's host loops, oracle/): through the containers' plans (CSR / COO / ELL / HYB), plan-less with the table, and with each explicit CSR
kernel.  Derandomised (the same examples every run), 80 examples per property, sizes up to 200 000 rows / columns.

Bars: a plan or kernel that declares storage-order sums (cmi_plan_info) must be BIT-EXACT; every other path |err| <= TOL * sum_j |a_ij x_j|
(1e-6 f64 / 1e-5 f32) -- the same bars as tests/test_spmv_gpu.py.  What the reference tests with fixed small matrices (testing/multiply.cu:
TestMultiply* over every format) is here asked of random ones.
"""
import numpy as np
import pytest

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, given, settings, strategies as st  # noqa: E402

pytestmark = pytest.mark.gpu

TOL = {np.dtype(np.float64): 1e-6, np.dtype(np.float32): 1e-5}
SETTINGS = dict(max_examples=80, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    return torch


def dev(a, torch):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def row_lengths(profile, rows, cols, rng):
    if rows == 0:
        return np.zeros(0, np.int64)
    if profile == "uniform":
        lens = rng.integers(0, min(cols, 12) + 1, size=rows)
    elif profile == "tail":           # most rows short, a few hundred times longer
        lens = np.minimum((rng.pareto(1.2, size=rows) * 3).astype(np.int64), cols * 4)
    elif profile == "deserts":        # stretches of empty rows between dense stretches (more empty rows than lanes in a wave)
        lens = rng.integers(1, 30, size=rows)
        for _ in range(3):
            a = int(rng.integers(0, rows))
            lens[a:a + int(rng.integers(1, 400))] = 0
    elif profile == "giant":          # one row holds almost everything (the merge-path kernel's case)
        lens = rng.integers(0, 4, size=rows)
        lens[int(rng.integers(0, rows))] = int(rng.integers(600, 6000))
    elif profile == "fem":            # 20..80 per row
        lens = rng.integers(20, 81, size=rows)
    else:                             # "equal": every row the same length (the wave-tile kernel's case)
        lens = np.full(rows, int(rng.integers(1, 10)))
    return lens.astype(np.int64)


def make_csr(profile, rows, cols, seed, dtype):
    rng = np.random.default_rng(seed)
    lens = row_lengths(profile, rows, cols, rng)
    Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
    nnz = int(Ap[-1])
    if profile == "fem" and nnz:      # runs of consecutive columns around the diagonal position
        start = np.repeat((np.arange(rows) * cols // max(rows, 1)).astype(np.int64), lens)
        within = np.arange(nnz) - np.repeat(Ap[:-1].astype(np.int64), lens)
        Aj = np.clip(start + within - 10, 0, cols - 1).astype(np.int32)
    else:
        Aj = rng.integers(0, cols, size=nnz).astype(np.int32)     # unsorted within a row, duplicates allowed: the sums do not care
    Ax = rng.standard_normal(nnz).astype(dtype)
    x = rng.standard_normal(cols).astype(dtype)
    y0 = rng.standard_normal(rows).astype(dtype)
    return Ap, Aj, Ax, x, y0


def check(got, want, bound, dtype, exact, what):
    if exact:
        assert np.array_equal(got, want), f"{what}: declared storage-order sums but not bit-exact"
        return
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    lim = TOL[np.dtype(dtype)] * np.maximum(bound.astype(np.float64), np.finfo(dtype).tiny)
    assert np.all(err <= lim), f"{what}: {int((err > lim).sum())} rows out of tolerance (worst {err.max():.3e})"


shapes = st.tuples(st.sampled_from(["uniform", "tail", "deserts", "giant", "fem", "equal"]),
                   st.one_of(st.integers(0, 5000), st.integers(5000, 200000)), st.one_of(st.integers(1, 5000), st.integers(5000, 200000)), st.integers(0, 2**31 - 1), st.sampled_from(["f64", "f32"]), st.booleans())


@settings(**SETTINGS)
@given(shapes)
def test_csr_every_path_against_the_oracle(cmi, torch_cuda, orc, shape):
    torch = torch_cuda
    profile, rows, cols, seed, tag, accumulate = shape
    dtype = np.float64 if tag == "f64" else np.float32
    Ap, Aj, Ax, x, y0 = make_csr(profile, rows, cols, seed, dtype)
    nnz = len(Aj)
    want = orc.spmv_csr(Ap, Aj, Ax, x, y0 if accumulate else None)
    bound = orc.spmv_csr(Ap, Aj, np.abs(Ax), np.abs(x)) + (np.abs(y0) if accumulate else 0)
    A = cmi.CsrMatrix(rows, cols, nnz, dev(Ap, torch), dev(Aj, torch), dev(Ax, torch))
    dx = dev(x, torch)

    def fresh():
        return dev(y0, torch) if accumulate else torch.full((rows,), 7.0, dtype=dx.dtype, device="cuda")
    what = f"{profile} {rows}x{cols} nnz {nnz} {tag} acc {accumulate} seed {seed}"
    # the container's plan (what cusp::multiply runs)
    y = fresh()
    cmi.multiply(A, dx, y, accumulate=accumulate)
    exact = bool(A.plan().info()["storage_order_sums"]) if nnz > 0 and rows > 0 else True
    check(y.cpu().numpy(), want, bound, dtype, exact, "plan: " + what)
    if nnz > 0 and rows > 0:
        assert A.plan().validate(A.row_offsets, A.column_indices)
    # plan-less with the table
    y = fresh()
    cmi.spmv_csr(rows, cols, A.row_offsets, A.column_indices, A.values, dx, y, accumulate=accumulate)
    check(y.cpu().numpy(), want, bound, dtype, False, "table: " + what)
    # explicit kernels: scalar (bit-exact by construction), vector, stream with one lane per row, balanced; wave-private tiles through plans
    for name, cfg, ex in (("scalar", cmi.Config(kernel=cmi.CSR_SCALAR), True),
                          ("vector8", cmi.Config(kernel=cmi.CSR_VECTOR, threads_per_row=8), False),
                          ("stream", cmi.Config(kernel=cmi.CSR_STREAM, threads_per_row=1), False),
                          ("balanced", cmi.Config(kernel=cmi.CSR_BALANCED), False)):
        y = fresh()
        cmi.spmv_csr(rows, cols, A.row_offsets, A.column_indices, A.values, dx, y, accumulate=accumulate, cfg=cfg)
        check(y.cpu().numpy(), want, bound, dtype, ex, name + ": " + what)
    if nnz > 0 and rows > 0:
        for name, cfg in (("wavev4", cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=4)),
                          ("wavev1", cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=1)),
                          ("wavex", cmi.Config(kernel=cmi.CSR_STREAM_WAVEX, items_per_thread=4, rows_per_block=2048))):
            try:
                plan = cmi.Plan.csr(dx.dtype, rows, cols, A.row_offsets, A.column_indices, cfg=cfg)
            except Exception:  # noqa: BLE001  (a row longer than a wave tile: the plan refuses the kernel -- its right)
                continue
            y = fresh()
            cmi.spmv_csr_plan(plan, A.row_offsets, A.column_indices, A.values, dx, y, accumulate=accumulate)
            check(y.cpu().numpy(), want, bound, dtype, bool(plan.info()["storage_order_sums"]), name + ": " + what)
        # round 4: the run-compressed column copy (pieces of consecutive columns: "fem" has long runs, every other profile runs of ~1 --
        # more pieces than the unrolled pass holds) and its packed twin, which also copies the values
        if cols >= 4:
            for name, make in (("waver4", lambda: cmi.Plan.csr(dx.dtype, rows, cols, A.row_offsets, A.column_indices, cfg=cmi.Config(kernel=cmi.CSR_STREAM_WAVER, items_per_thread=4))),
                               ("waver1/cap3", lambda: cmi.Plan.csr(dx.dtype, rows, cols, A.row_offsets, A.column_indices, cfg=cmi.Config(kernel=cmi.CSR_STREAM_WAVER, items_per_thread=1, threads_per_row=3))),
                               ("packed2", lambda: cmi.Plan.csr_values(rows, cols, A.row_offsets, A.column_indices, A.values, cfg=cmi.Config(kernel=cmi.CSR_STREAM_PACKED, items_per_thread=2)))):
                try:
                    plan = make()
                except Exception:  # noqa: BLE001  (a row longer than half a wave tile, or a row of 512+: refused by name)
                    continue
                y = fresh()
                cmi.spmv_csr_plan(plan, A.row_offsets, A.column_indices, A.values, dx, y, accumulate=accumulate)
                check(y.cpu().numpy(), want, bound, dtype, bool(plan.info()["storage_order_sums"]), name + ": " + what)
                assert plan.info()["storage_order_sums"] or plan.config().kernel not in (cmi.CSR_STREAM_WAVER,), name


@settings(**SETTINGS)
@given(shapes)
def test_other_formats_through_their_plans_against_the_oracle(cmi, torch_cuda, orc, shape):
    """CSR -> COO / ELL / HYB on the device (the reference's conversions), then cusp::multiply's path for each; COO also in a random entry
    order, sorted on the device first (stable: the oracle's chain on the entries as given)."""
    torch = torch_cuda
    profile, rows, cols, seed, tag, accumulate = shape
    if profile in ("giant", "tail"):
        rows, cols = min(rows, 1500), min(cols, 5000)   # (ELL of a giant row: rows x its length in slots)
    dtype = np.float64 if tag == "f64" else np.float32
    Ap, Aj, Ax, x, y0 = make_csr(profile, rows, cols, seed, dtype)
    nnz = len(Aj)
    if rows == 0:
        return
    want = orc.spmv_csr(Ap, Aj, Ax, x, y0 if accumulate else None)
    bound = orc.spmv_csr(Ap, Aj, np.abs(Ax), np.abs(x)) + (np.abs(y0) if accumulate else 0)
    A = cmi.CsrMatrix(rows, cols, nnz, dev(Ap, torch), dev(Aj, torch), dev(Ax, torch))
    dx = dev(x, torch)
    what = f"{profile} {rows}x{cols} nnz {nnz} {tag} acc {accumulate} seed {seed}"
    for fmt in ("coo", "ell", "hyb"):
        M = cmi.convert(A, fmt)
        y = dev(y0, torch) if accumulate else torch.full((rows,), 7.0, dtype=dx.dtype, device="cuda")
        cmi.multiply(M, dx, y, accumulate=accumulate)
        check(y.cpu().numpy(), want, bound, dtype, False, fmt + ": " + what)
        back = cmi.convert(M, "csr")                       # and the way back reproduces the matrix (zero VALUES may be dropped by ELL / HYB padding rules: compare products)
        y2 = torch.full((rows,), 7.0, dtype=dx.dtype, device="cuda")
        cmi.multiply(back, dx, y2, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
        check(y2.cpu().numpy(), orc.spmv_csr(Ap, Aj, Ax, x), orc.spmv_csr(Ap, Aj, np.abs(Ax), np.abs(x)), dtype, False, fmt + " -> csr: " + what)
    if nnz >= 2:
        perm = np.random.default_rng(seed ^ 0x5EED).permutation(nnz)
        Ai = np.repeat(np.arange(rows, dtype=np.int32), np.diff(Ap))
        U = cmi.CooMatrix(rows, cols, nnz, dev(Ai[perm], torch), dev(Aj[perm], torch), dev(Ax[perm], torch))
        y = dev(y0, torch) if accumulate else torch.full((rows,), 7.0, dtype=dx.dtype, device="cuda")
        cmi.multiply(U, dx, y, accumulate=accumulate)       # any order: the atomics kernels
        check(y.cpu().numpy(), want, bound, dtype, False, "coo as given: " + what)
        U.sort_by_row()
        order = np.argsort(Ai[perm], kind="stable")
        assert np.array_equal(U.column_indices.cpu().numpy(), Aj[perm][order]) and np.array_equal(U.values.cpu().numpy(), Ax[perm][order])
        y = dev(y0, torch) if accumulate else torch.full((rows,), 7.0, dtype=dx.dtype, device="cuda")
        cmi.multiply(U, dx, y, accumulate=accumulate)
        chain = orc.spmv_coo(rows, Ai[perm], Aj[perm], Ax[perm], x, y0 if accumulate else None)  # the host loop on the entries as given
        exact = nnz >= 4 and bool(U.plan().info()["storage_order_sums"])
        check(y.cpu().numpy(), chain, bound, dtype, exact, "coo sorted on the device: " + what)
