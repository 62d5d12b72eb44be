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


# ------------------------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3] at FULL SIZE: nlpkkt120 / ldoor / thermal2 -- the real files when CMI_SUITESPARSE_DIR
# has them (dimensions from the file), else the seeded stand-ins at scale 1.0 with the collection's published row-length range.
# The sweep being replaced: performance/csr_vector/csr_vector.cu:41-62,86-110 (THREADS_PER_VECTOR over the testing/UF downloads).
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["thermal2", "ldoor", "nlpkkt120"])
def test_configs3_full_size_every_csr_variant(cmi, torch_cuda, orc, name):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import suitesparse_like as ssl
    torch = torch_cuda
    Ap, Aj, Ax, source = ssl.load(name, 1.0)
    st, pub = ssl.stats(Ap, Aj), ssl.PUBLISHED[name]
    print(f"{name}: {source}: {st}")
    if source.startswith("seeded"):
        print(f"{name}: CMI_SUITESPARSE_DIR has no {name}.mtx -- real-file branch SKIPPED, stand-in used (published: {pub})")
        assert (st["min"], st["max"]) == (pub["min"], pub["max"]), (st, pub)  # the published row-length extremes, exactly
        assert abs(st["mean"] - pub["mean"]) <= 0.08 * pub["mean"] and abs(st["rows"] - pub["rows"]) <= 0.03 * pub["rows"], (st, pub)
    rows = cols = st["rows"]
    nnz = st["entries"]
    x = orc.fill_x(cols)
    want = orc.spmv_csr(Ap, Aj, Ax, x, omp=True)                       # the OpenMP oracle: per-row arithmetic = the sequential loop's
    bound = np.maximum(orc.spmv_csr(Ap, Aj, np.abs(Ax), np.abs(x), omp=True), 1e-300)
    dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
    y = torch.empty(rows, dtype=torch.float64, device="cuda")

    def check(label, exact):
        got = y.cpu().numpy()
        if exact:
            assert np.array_equal(got, want), f"{name} {label}: not bit-identical to the host loop"
        else:
            assert np.all(np.abs(got - want) <= 1e-6 * bound), f"{name} {label}: beyond 1e-6"

    def run(label, cfg, exact):
        y.fill_(10.0)
        cmi.spmv_csr(rows, cols, dAp, dAj, dAx, dx, y, cfg=cfg)
        check(label, exact)

    run("csr_scalar", cmi.Config(kernel=cmi.CSR_SCALAR), True)
    for tpr in (2, 4, 8, 16, 32, 64):                                  # the reference's sweep
        run(f"csr_vector T={tpr}", cmi.Config(kernel=cmi.CSR_VECTOR, threads_per_row=tpr), False)
    for tpr, ipt in ((1, 1), (1, 2), (1, 4), (4, 2), (16, 1)):
        run(f"csr_stream lanes/row={tpr} vectors/lane={ipt}", cmi.Config(kernel=cmi.CSR_STREAM, threads_per_row=tpr, items_per_thread=ipt), tpr == 1)
    run("csr_stream lane-strided", cmi.Config(kernel=cmi.CSR_STREAM, threads_per_row=1, nontemporal=4 | 2), True)
    run("csr_balanced", cmi.Config(kernel=cmi.CSR_BALANCED), False)
    run("table (NULL config)", None, True)
    # plans: the default, the wave-private vector kernel, the 16-bit column copy where it is granted, sorted COO through its plan
    plan = cmi.Plan(cmi.FORMAT_CSR, torch.float64, rows, cols, nnz, dAp)
    assert plan.info()["max_row_length"] == st["max"] and plan.info()["storage_order_sums"]
    y.fill_(10.0)
    cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y)
    check("plan", True)
    for v in (0, 2, 4) if name != "thermal2" else (0, 1, 2):
        pv = cmi.Plan(cmi.FORMAT_CSR, torch.float64, rows, cols, nnz, dAp, cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=v))
        assert pv.config().kernel == cmi.CSR_STREAM_WAVEV
        y.fill_(10.0)
        cmi.spmv_csr_plan(pv, dAp, dAj, dAx, dx, y)
        check(f"csr_wavev V={v}", True)
    if name == "thermal2":
        pw = cmi.Plan(cmi.FORMAT_CSR, torch.float64, rows, cols, nnz, dAp, cmi.Config(kernel=cmi.CSR_STREAM_WAVE, rows_per_block=-1))
        y.fill_(10.0)
        cmi.spmv_csr_plan(pw, dAp, dAj, dAx, dx, y)
        check("csr_wave on a row partition", True)
    p16 = cmi.Plan.csr(torch.float64, rows, cols, dAp, dAj, cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
    y.fill_(10.0)
    cmi.spmv_csr_plan(p16, dAp, dAj, dAx, dx, y)
    check(f"16-bit column plan (granted: {p16.config().kernel == cmi.CSR_STREAM_C16})", True)
    dAi = torch.empty(nnz, dtype=torch.int32, device="cuda")
    cmi.check(cmi.lib().cmi_csr_row_indices(rows, ctypes.c_void_p(dAp.data_ptr()), ctypes.c_void_p(dAi.data_ptr()), None))
    cplan = cmi.Plan(cmi.FORMAT_COO, torch.float64, rows, cols, nnz, dAi)
    y.fill_(10.0)
    cmi.spmv_coo_plan(cplan, dAi, dAj, dAx, dx, y)
    check("sorted COO through its plan", True)
    # the containers' own path (cusp::multiply's mirror): plan made at the first multiply
    A = cmi.CsrMatrix(rows, cols, nnz, dAp, dAj, dAx)
    y.fill_(10.0)
    cmi.multiply(A, dx, y)
    check("multiply(A, x, y)", True)


def test_python_comm_one_rank_through_rccl(cmi, torch_cuda, orc):
    """binding.Comm (cmi_comm: RCCL behind the C-ABI) with ONE rank -- the calls the sharded SpMV / CG make on an 8-GPU node:
    in-place all-gather, all-gather of unequal pieces (both algorithms), all-reduce, barrier, host records; then a ShardedCsr and
    krylov.cg carried by that communicator instead of torch.distributed, against the single-GPU results."""
    torch = torch_cuda
    comm = cmi.binding.Comm(0, 1)
    assert comm.library_version() > 20000
    v = torch.arange(1000, dtype=torch.float64, device="cuda")
    comm.allgather(v, v, 1000)
    w = torch.zeros(1000, dtype=torch.float64, device="cuda")
    comm.allgatherv(v, w, [1000], [0], algorithm=0)
    comm.allgatherv(v[:500], w[:], [500], [0], algorithm=1)
    s = torch.tensor([2.5, -1.0], dtype=torch.float64, device="cuda")
    comm.allreduce(s)
    comm.allreduce(s, op=cmi.binding.OP_MAX)
    comm.barrier()
    assert torch.equal(w, v) and s.tolist() == [2.5, -1.0]
    assert comm.allgather_host(b"abcdefgh") == [b"abcdefgh"]
    f = torch.arange(64, dtype=torch.float32, device="cuda")
    comm.allgather(f, f, 64)
    comm.halo_exchange(v, [], [], [], [], [])
    # the sharded operator + CG on that communicator (world 1: the exchange is RCCL's one-rank all-gather)
    m, n = 97, 83
    N = m * n
    A = cmi.poisson5pt(m, n, "csr")
    sh = cmi.distributed.ShardedCsr(A, N, 0, 1, mode="allgather", comm=comm)
    assert sh.vec.comm is comm
    x = cmi.fill_x(N).cuda()
    sh.x_local.copy_(x)
    y = torch.empty(N, dtype=torch.float64, device="cuda")
    sh.multiply(y)
    Ap, Aj, Ax = orc.poisson5pt_csr(m, n)
    assert np.array_equal(y.cpu().numpy(), orc.spmv_csr(Ap, Aj, Ax, orc.fill_x(N)))
    b = torch.ones(N, dtype=torch.float64, device="cuda")
    x1, x2 = torch.zeros_like(b), torch.zeros_like(b)
    mon1 = cmi.krylov.cg(sh, x1, b, iteration_limit=400, relative_tolerance=1e-8)
    mon2 = cmi.krylov.cg(A, x2, b, iteration_limit=400, relative_tolerance=1e-8)
    assert mon1.iteration_count == mon2.iteration_count and mon1.converged()
    assert torch.allclose(x1, x2, rtol=0, atol=1e-10)
    comm.close()


def test_every_entry_of_the_shipped_tuning_table_against_the_oracle(cmi, torch_cuda, orc):
    """the shipped table (cusp-autotuned_amd/tuned/gfx950.json) is loaded as data and EVERY entry -- format x
    value type x bucket -- is run, as an explicit config, on a matrix of its bucket's width and compared with the oracle (the
    validation pattern of the reference's testing/ktt.cu:142-202: reference y first, then every configuration against it).
    One-lane-per-row shapes must have the host loop's bits; lane groups 1e-6 (f64) / 1e-5 (f32) of the row's |a||x| sum."""
    import json
    torch = torch_cuda
    table = json.load(open(os.path.join(os.path.dirname(cmi.lib_path()), "..", "tuned", "gfx950.json")))
    entries = table["entries"]
    assert len(entries) >= 64  # (round 4: the 16 coo_sorted keys are retired -- sorted COO multiplies through its plan)
    seen = set()
    for e in entries:
        fmt, tag, bucket = e["format"], e["dtype"], e["bucket"]
        seen.add((fmt, tag, bucket))
        dtype, tdt = (np.float64, torch.float64) if tag == "f64" else (np.float32, torch.float32)
        width = max(1, int(round(e["mean"])))
        rows = 6000 if width <= 40 else 2500
        rng = np.random.default_rng(1000 * bucket + len(fmt))
        # banded: near diagonals and a few far ones (every format can hold it; DIA needs few distinct offsets)
        offs = sorted({0} | {(-1) ** k * ((k + 1) // 2) * (1 if k < 4 else 37) for k in range(1, width)})
        while len(offs) < width:
            offs.append(max(offs) + 37)
        r = np.arange(rows, dtype=np.int64)
        cols2 = r[:, None] + np.array(offs, np.int64)[None, :]
        mask = (cols2 >= 0) & (cols2 < rows)
        Ap = np.r_[0, np.cumsum(mask.sum(axis=1))].astype(np.int32)
        Aj = cols2[mask].astype(np.int32)
        Ax = rng.standard_normal(len(Aj)).astype(dtype)
        x = rng.standard_normal(rows).astype(dtype)
        want = orc.spmv_csr(Ap, Aj, Ax, x)
        bound = np.maximum(orc.spmv_csr(Ap, Aj, np.abs(Ax), np.abs(x)), 1e-30)
        cfg = cmi.Config(**{k: e[k] for k in ("kernel", "block_size", "threads_per_row", "rows_per_block", "items_per_thread", "nontemporal", "xcd_swizzle", "blocks_per_cu")})
        A = cmi.CsrMatrix(rows, rows, len(Aj), dev(Ap, torch), dev(Aj, torch), dev(Ax, torch))
        M = A if fmt == "csr" else cmi.convert(A, {"coo_sorted": "coo"}.get(fmt, fmt))
        y = torch.full((rows,), 10.0, dtype=tdt, device="cuda")
        cmi.multiply(M, dev(x, torch), y, cfg=cfg)
        got = y.cpu().numpy()
        exact = (fmt == "csr" and e["kernel"] in (cmi.CSR_SCALAR, cmi.CSR_STREAM, cmi.CSR_STREAM_PIPE) and e["threads_per_row"] <= 1) or \
                (fmt == "ell" and e["threads_per_row"] == 1) or fmt == "dia" or e["kernel"] == cmi.COO_TILE
        if exact:
            assert np.array_equal(got, want), (fmt, tag, bucket, e)
        else:
            tol = 1e-6 if tag == "f64" else 1e-5
            assert np.all(np.abs(got - want) <= tol * bound), (fmt, tag, bucket, e)
        # and what a NULL config selects for this shape is this entry (the table is what the library consults)
        sel = cmi.tuning_select({"csr": 0, "ell": 1, "dia": 2, "coo": 3, "coo_sorted": cmi.TABLE_COO_SORTED}[fmt], cmi.F64 if tag == "f64" else cmi.F32,
                                rows, rows, len(Aj) if fmt in ("csr", "coo", "coo_sorted") else rows * width)
        actual_mean = len(Aj) / rows if fmt in ("csr", "coo", "coo_sorted") else float(width)
        actual_bucket = 0 if actual_mean <= 1.0 else min(7, int(np.floor(np.log2(actual_mean))))
        if actual_bucket == bucket and fmt != "csr":  # (boundary rows can put a width-2^b matrix one bucket lower: then another entry is consulted)
            assert sel.kernel == e["kernel"], (fmt, tag, bucket, sel, e)
    assert len(seen) == len(entries), "duplicate (format, dtype, bucket) keys in the shipped table"
    for fmt in ("csr", "ell", "dia", "coo"):
        for tag in ("f64", "f32"):
            assert {b for f, t, b in seen if f == fmt and t == tag} == set(range(8)), (fmt, tag)
    # round 4: the table's coo_sorted keys are retired -- sorted COO multiplies through its plan's row offsets + the CSR
    # kernels; CMI_COO_TILE stays an explicit-config kernel with its built-in shape
    assert not any(f == "coo_sorted" for f, _, _ in seen)
    assert cmi.tuning_select(cmi.TABLE_COO_SORTED, cmi.F64, 10000, 10000, 50000).kernel == cmi.COO_TILE


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_wavex_lds_window_bit_exact(cmi, torch_cuda, orc, tag):
    """CMI_CSR_STREAM_WAVEX = csr_wavev + an x window in LDS per workgroup: columns inside the window are gathered from LDS, columns
    outside from memory -- the same products in the same order either way, so every row must keep the host loop's bits
    (cusp/system/detail/sequential/multiply/csr_spmv.h:42-74): band matrices (all columns inside), columns far outside the window,
    rectangular matrices, windows of every allowed length, accumulate and the fused <y, w>."""
    torch = torch_cuda
    dtype, tdt = (np.float64, torch.float64) if tag == "f64" else (np.float32, torch.float32)
    for seed, rows, cols, lo, hi, band in ((1, 9000, 9000, 2, 9, 700), (2, 5000, 5000, 10, 30, 1500), (3, 7001, 9100, 0, 12, 9100), (4, 4099, 3000, 20, 60, 300)):
        rng = np.random.default_rng(seed)
        lens = rng.integers(lo, hi + 1, size=rows)
        lens[100:400] = 0
        if int(lens.sum()) % 4 == 0:
            lens[0] += 1
        Ap = np.r_[0, np.cumsum(lens)].astype(np.int32)
        nnz = int(Ap[-1])
        ri = np.repeat(np.arange(rows, dtype=np.int64), lens)
        centre = ri * cols // rows
        Aj = np.clip(centre + rng.integers(-band, band + 1, size=nnz), 0, cols - 1).astype(np.int32)
        Ax = rng.standard_normal(nnz).astype(dtype)
        x = rng.standard_normal(cols).astype(dtype)
        y0 = rng.standard_normal(rows).astype(dtype)
        want, want_acc = orc.spmv_csr(Ap, Aj, Ax, x), orc.spmv_csr(Ap, Aj, Ax, x, y0)
        dAp, dAj, dAx, dx = dev(Ap, torch), dev(Aj, torch), dev(Ax, torch), dev(x, torch)
        for v, window in ((2, 0), (4, 0), (2, 512 if tag == "f64" else 1024), (4, 2048), (4, 100000)):
            plan = cmi.Plan(cmi.FORMAT_CSR, tdt, rows, cols, nnz, dAp, cmi.Config(kernel=cmi.CSR_STREAM_WAVEX, items_per_thread=v, rows_per_block=window))
            c = plan.config()
            assert c.kernel == cmi.CSR_STREAM_WAVEX and c.items_per_thread == v and plan.info()["storage_order_sums"]
            y = torch.full((rows,), 9.0, dtype=tdt, device="cuda")
            cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y)
            assert np.array_equal(y.cpu().numpy(), want), (seed, v, window)
            y = dev(y0, torch)
            cmi.spmv_csr_plan(plan, dAp, dAj, dAx, dx, y, accumulate=True)
            assert np.array_equal(y.cpu().numpy(), want_acc), (seed, v, window, "accumulate")
        w = rng.standard_normal(rows).astype(dtype)
        res = torch.zeros(1, dtype=torch.float64, device="cuda")
        y = torch.zeros(rows, dtype=tdt, device="cuda")
        cmi.spmv_csr_dot(rows, cols, dAp, dAj, dAx, dx, y, dev(w, torch), res, cmi.blas_workspace(), plan=plan)
        assert np.array_equal(y.cpu().numpy(), want), (seed, "dot")
        ref = float(np.dot(want.astype(np.float64), w.astype(np.float64)))
        assert abs(float(res) - ref) <= 1e-12 * float(np.abs(want.astype(np.float64) * w).sum()) + (0 if tag == "f64" else 1e-6 * abs(ref)), seed


def test_auto_plans_pick_wavev_and_wavex_by_the_measured_rule(cmi, torch_cuda, orc):
    """The plan's auto rule (plan.hip wavev_auto / column_profile, measured in profiles/r03_wavev_wavex_ab.txt): large irregular
    matrices of 2.5-44 entries per row run csr_wavev; made WITH the column indices, a plan whose columns lie anywhere inside a band
    adds the LDS x window (csr_wavex); small matrices, FEM-like column runs and the plan-less path stay what they were.  Every choice
    keeps the host loop's bits."""
    torch = torch_cuda

    def irregular(rows, lo, hi, band, seed, dt, runs=1):
        g = torch.Generator(device="cuda").manual_seed(seed)
        lens = torch.randint(lo, hi + 1, (rows,), device="cuda", generator=g) * runs
        Ap = torch.zeros(rows + 1, dtype=torch.int32, device="cuda")
        Ap[1:] = lens.cumsum(0).to(torch.int32)
        nnz = int(Ap[-1])
        row = torch.repeat_interleave(torch.arange(rows, device="cuda"), lens)
        if runs == 1:
            Aj = (row + torch.randint(-band, band + 1, (nnz,), device="cuda", generator=g)) % rows
        else:  # runs of `runs` consecutive columns (FEM blocks): neighbours share x lines
            base = torch.randint(-band, band + 1, (nnz // runs,), device="cuda", generator=g).repeat_interleave(runs)
            Aj = (row + base + torch.arange(nnz, device="cuda") % runs) % rows
        return Ap, Aj.to(torch.int32), torch.randn(nnz, dtype=dt, device="cuda", generator=g)

    cases = [("band f32", irregular(4_000_000, 5, 12, 2000, 4, torch.float32), cmi.CSR_STREAM_WAVEX, cmi.CSR_STREAM_WAVEV),
             ("band f64", irregular(3_000_000, 5, 12, 2000, 5, torch.float64), cmi.CSR_STREAM_WAVEX, cmi.CSR_STREAM_WAVEV),
             # (round 4: columns in runs of 3 -> a plan made WITH the columns multiplies from the run-compressed copy, csr_waver)
             ("blocks of 3 f64", irregular(1_500_000, 3, 8, 2000, 6, torch.float64, runs=3), cmi.CSR_STREAM_WAVER, cmi.CSR_STREAM_WAVEV),
             ("small band f64", irregular(200_000, 5, 12, 2000, 7, torch.float64), cmi.CSR_STREAM, cmi.CSR_STREAM)]
    for name, (Ap, Aj, Ax), want_with_columns, want_offsets_only in cases:
        N, nnz, dt = Ap.numel() - 1, Aj.numel(), Ax.dtype
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(N, dtype=dt, device="cuda", generator=g)
        want = orc.spmv_csr(Ap.cpu().numpy(), Aj.cpu().numpy(), Ax.cpu().numpy(), x.cpu().numpy())
        p_cols = cmi.Plan.csr(dt, N, N, Ap, Aj)
        p_offs = cmi.Plan(cmi.FORMAT_CSR, dt, N, N, nnz, Ap)
        assert p_cols.config().kernel == want_with_columns, (name, p_cols.config())
        assert p_offs.config().kernel == want_offsets_only, (name, p_offs.config())
        for p in (p_cols, p_offs):
            assert p.info()["storage_order_sums"]
            y = torch.full((N,), 3.0, dtype=dt, device="cuda")
            cmi.spmv_csr_plan(p, Ap, Aj, Ax, x, y)
            assert np.array_equal(y.cpu().numpy(), want), (name, p.config())
        # a caller who ASKS for csr_wave's lane-strided body on a row partition gets that kernel, not the rule's (found in session 30)
        p_part = cmi.Plan(cmi.FORMAT_CSR, dt, N, N, nnz, Ap, cmi.Config(kernel=cmi.CSR_STREAM_WAVE, rows_per_block=-1, items_per_thread=8))
        assert p_part.config().kernel == cmi.CSR_STREAM_WAVE and p_part.config().items_per_thread == 8, (name, p_part.config())
        y = torch.full((N,), 3.0, dtype=dt, device="cuda")
        cmi.spmv_csr_plan(p_part, Ap, Aj, Ax, x, y)
        assert np.array_equal(y.cpu().numpy(), want), (name, p_part.config())
        y = torch.full((N,), 3.0, dtype=dt, device="cuda")
        cmi.spmv_csr(N, N, Ap, Aj, Ax, x, y)  # plan-less: the table's csr_stream, whatever the matrix
        assert np.array_equal(y.cpu().numpy(), want), name
        # the container path (cusp::multiply's mirror) makes its plan with the columns
        A = cmi.CsrMatrix(N, N, nnz, Ap, Aj, Ax)
        y.fill_(1.0)
        cmi.multiply(A, x, y)
        assert A.plan().config().kernel == want_with_columns and np.array_equal(y.cpu().numpy(), want), name
        del p_cols, p_offs, A


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_device_coo_sort_by_row_is_stable_and_feeds_the_plan_path(cmi, torch_cuda, orc, tag):
    """coo_matrix::sort_by_row / sort_by_row_and_column / is_sorted_* on the device (reference cusp/coo_matrix.h:208-224,
    cusp/sort.h:231,302).  Integer work: the sorted arrays must be EXACTLY what a stable host sort gives.  And the point of it:
    an any-order COO matrix sorted once multiplies through its plan (the CSR kernels on plan-built offsets) with the bits of the
    reference's host COO loop (sequential/multiply/coo_spmv.h) run on the UNSORTED entries -- the stable sort keeps every row's
    chain."""
    torch = torch_cuda
    dtype = np.float64 if tag == "f64" else np.float32
    rng = np.random.default_rng(11)
    for rows, cols, n in ((1, 1, 1), (5, 7, 2), (1000, 900, 37), (70000, 70000, 700001), (30, 50, 4096), (1 << 20, 1 << 20, 3_000_000)):
        Ai = rng.integers(0, rows, size=n).astype(np.int32)
        Aj = rng.integers(0, cols, size=n).astype(np.int32)
        Ax = rng.standard_normal(n).astype(dtype)
        x = rng.standard_normal(cols).astype(dtype)
        C = cmi.CooMatrix(rows, cols, n, dev(Ai, torch), dev(Aj, torch), dev(Ax, torch))
        assert C.is_sorted_by_row() == bool(np.all(Ai[1:] >= Ai[:-1]))
        C.sort_by_row()
        order = np.argsort(Ai, kind="stable")
        assert np.array_equal(C.row_indices.cpu().numpy(), Ai[order]), (rows, n)
        assert np.array_equal(C.column_indices.cpu().numpy(), Aj[order]) and np.array_equal(C.values.cpu().numpy(), Ax[order])
        assert C.is_sorted_by_row()
        # the multiply through the plan on the sorted matrix == the host loop on the entries as they were given
        want = orc.spmv_coo(rows, Ai, Aj, Ax, x)
        y = torch.full((rows,), 3.0, dtype=C.values.dtype, device="cuda")
        cmi.multiply(C, dev(x, torch), y)
        info = C.plan().info()
        if n >= 4:
            assert info["coo_sorted"] == 1
        if info["storage_order_sums"] or n < 4:        # one lane per row: the host chain, bit for bit
            assert np.array_equal(y.cpu().numpy(), want), (rows, n)
        else:                                          # rows of 100+ entries: the table gives a row several lanes (re-associated sums)
            bound = (1e-6 if tag == "f64" else 1e-5) * orc.spmv_coo(rows, Ai, Aj, np.abs(Ax), np.abs(x))
            assert np.all(np.abs(y.cpu().numpy() - want) <= bound), (rows, n)
        # by (row, column): a second call on a fresh copy
        D = cmi.CooMatrix(rows, cols, n, dev(Ai, torch), dev(Aj, torch), dev(Ax, torch))
        D.sort_by_row_and_column()
        order2 = np.lexsort((np.arange(n), Aj, Ai))   # stable in (row, column)
        assert np.array_equal(D.row_indices.cpu().numpy(), Ai[order2]) and np.array_equal(D.column_indices.cpu().numpy(), Aj[order2])
        assert np.array_equal(D.values.cpu().numpy(), Ax[order2])
        assert D.is_sorted_by_row_and_column() and D.is_sorted_by_row()
        before = D.values.clone()
        D.sort_by_row_and_column()                    # already in order: nothing moves
        assert torch.equal(before, D.values)
    # a row index outside the matrix: refused, arrays untouched
    Ai = np.array([3, 1, 9, 0], dtype=np.int32)
    C = cmi.CooMatrix(9, 9, 4, dev(Ai, torch), dev(Ai % 9, torch), dev(np.ones(4, dtype=dtype), torch))
    with pytest.raises(Exception):
        C.sort_by_row()
    assert np.array_equal(C.row_indices.cpu().numpy(), Ai)
    # empty
    E = cmi.CooMatrix(4, 4, 0, dev(np.zeros(0, np.int32), torch), dev(np.zeros(0, np.int32), torch), dev(np.zeros(0, dtype), torch))
    E.sort_by_row()
    assert E.is_sorted_by_row() and E.is_sorted_by_row_and_column()


def test_interior_rows_of_a_row_block(cmi, torch_cuda):
    """cmi_csr_interior_rows (the split of a sharded multiply into rows that need only the rank's own slice of x and the boundary rows at
    the block's two ends) against a numpy restatement: poisson5pt row blocks, a block with no outside column, one with
    outside columns in the middle (no usable range), empty rows."""
    torch = torch_cuda
    m, n = 300, 200
    for r0, r1 in ((0, 20000), (20000, 41000), (41000, 60000)):
        A = cmi.poisson5pt(m, n, fmt="csr", dtype=torch.float64, row_begin=r0, row_end=r1)
        Ap, Aj = A.row_offsets.cpu().numpy(), A.column_indices.cpu().numpy()
        rows = r1 - r0
        outside = np.array([np.any((Aj[Ap[i]:Ap[i + 1]] < r0) | (Aj[Ap[i]:Ap[i + 1]] >= r1)) for i in range(rows)])
        below = np.nonzero(outside[:rows // 2])[0]
        above = np.nonzero(outside[rows // 2:])[0]
        want = (int(below[-1]) + 1 if below.size else 0, int(above[0]) + rows // 2 if above.size else rows)
        assert cmi.csr_interior_rows(rows, A.row_offsets, A.column_indices, r0, r1) == want, (r0, r1)
        if r0 == 20000:
            assert want == (m, rows - m)      # one grid line at each end reaches into the neighbours
    assert cmi.csr_interior_rows(rows, A.row_offsets, A.column_indices, 0, m * n) == (0, rows)          # nothing outside
    # empty rows and an outside column in the middle of each half
    Ap = torch.tensor([0, 0, 2, 2, 3, 5, 5, 6], dtype=torch.int32, device="cuda")
    Aj = torch.tensor([3, 4, 9, 3, 4, 0], dtype=torch.int32, device="cuda")
    assert cmi.csr_interior_rows(7, Ap, Aj, 1, 8) == (0, 3)   # rows 3 (column 9) and 6 (column 0) reach outside; mid = 3: nothing in the first half, row 3 is the first of the second
