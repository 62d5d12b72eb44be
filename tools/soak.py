"""Repeat every kernel thousands of times on the headline matrix and check the results never change (tools)."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, cusp_autotuned_amd as cmi
A = cmi.poisson5pt(3162, 3162, "csr")
n = A.num_rows
x = cmi.fill_x(n).cuda()
y0 = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty_like(y0)
ws = cmi.blas_workspace()
res = torch.zeros(1, dtype=torch.float64, device="cuda")
cmi.multiply(A, x, y0)
cmi.spmv_csr_dot(n, n, A.row_offsets, A.column_indices, A.values, x, y, x, res, ws)
d0 = float(res)
t0 = time.time()
bad = 0
for it in range(200):
    for _ in range(100):
        cmi.multiply(A, x, y)
    cmi.spmv_csr_dot(n, n, A.row_offsets, A.column_indices, A.values, x, y, x, res, ws)
    if not torch.equal(y, y0) or float(res) != d0:
        bad += 1
print("20000 SpMV + 200 fused dots in %.1f s, mismatching checks: %d" % (time.time() - t0, bad))
# balanced kernel + ELL/DIA/COO loops
for fmt in ("ell", "dia", "coo", "hyb"):
    M = cmi.poisson5pt(3162, 3162, "dia") if fmt == "dia" else cmi.convert(A, fmt, num_entries_per_row=5 if fmt == "hyb" else None)
    cmi.multiply(M, x, y0)
    ok = True
    for _ in range(2000):
        cmi.multiply(M, x, y)
    ok = torch.allclose(y, y0, rtol=1e-12, atol=1e-12)
    print(fmt, "2000 multiplies consistent:", bool(ok))
bal = cmi.Config(kernel=cmi.CSR_BALANCED)
cmi.multiply(A, x, y0, cfg=bal)
for _ in range(2000):
    cmi.multiply(A, x, y, cfg=bal)
print("balanced 2000 multiplies consistent:", bool(torch.allclose(y, y0, rtol=1e-12, atol=1e-12)))

# the long-row path (LONG instances), the merge-path kernel on a skewed matrix, and the fused CG in both value types
import numpy as np
rng = np.random.default_rng(5)
rows = 400000
lens = rng.integers(2, 9, size=rows)
lens[rng.integers(0, rows, size=40)] = 3000
lens[123] = 150000
Ap = torch.from_numpy(np.r_[0, np.cumsum(lens)].astype(np.int32)).cuda()
nnz = int(Ap[-1])
Aj = torch.randint(0, rows, (nnz,), dtype=torch.int32, device="cuda")
Ax = torch.rand(nnz, dtype=torch.float64, device="cuda")
xs = torch.rand(rows, dtype=torch.float64, device="cuda")
for name, cfg in (("csr_stream long rows", cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, rows, rows, nnz)), ("csr_balanced", cmi.Config(kernel=cmi.CSR_BALANCED)), ("table", None)):
    ya = torch.empty(rows, dtype=torch.float64, device="cuda")
    yb = torch.empty_like(ya)
    cmi.spmv_csr(rows, rows, Ap, Aj, Ax, xs, ya, cfg=cfg)
    same = True
    for _ in range(3000):
        cmi.spmv_csr(rows, rows, Ap, Aj, Ax, xs, yb, cfg=cfg)
    torch.cuda.synchronize()
    if name == "csr_stream long rows":  # fixed fold order: bit-identical; the merge-path kernel (also what the table picks for
        same = torch.equal(ya, yb)      # this matrix: a 150000-entry row) completes split rows with atomics: rounding may differ
    else:
        same = bool(torch.allclose(ya, yb, rtol=1e-12, atol=1e-12))
    print(f"{name}: 3000 multiplies of a skewed matrix consistent: {same}")
for dt in (torch.float64, torch.float32):
    P = cmi.poisson5pt(1000, 1000, "csr", dtype=dt)
    b = cmi.fill_x(P.num_rows, dt, "cuda")
    hist = None
    for _ in range(5):
        xk = torch.zeros(P.num_rows, dtype=dt, device="cuda")
        h = cmi.krylov.cg(P, xk, b, iteration_limit=400, relative_tolerance=0.0).residuals
        hist = hist or h
        assert h == hist, "CG history changed between runs"
    print(f"fused CG {dt}: 5 x 400 iterations, identical residual histories: True")

# round 2's kernels: the 16-bit column plan, HYB in one launch (a real split), sorted COO through its plan's row offsets, ELL with
# several lanes per row, the fold-ahead CG steps -- thousands of repeats each, results must never change
p16 = cmi.Plan.csr(torch.float64, n, n, A.row_offsets, A.column_indices, cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
assert p16.config().kernel == cmi.CSR_STREAM_C16
cmi.multiply(A, x, y0)
bad = 0
for _ in range(20):
    for _ in range(100):
        cmi.spmv_csr_plan(p16, A.row_offsets, A.column_indices, A.values, x, y)
    bad += not torch.equal(y, y0)
print("16-bit column plan: 2000 multiplies, mismatching checks:", bad)
for name, M in (("hyb K=4 one launch", cmi.convert(A, "hyb", num_entries_per_row=4)), ("hyb K=1 two launches", cmi.convert(A, "hyb", num_entries_per_row=1)),
                ("coo through its plan", cmi.convert(A, "coo"))):
    bad = 0
    for _ in range(20):
        for _ in range(100):
            cmi.multiply(M, x, y)
        bad += not torch.equal(y, y0)
    print(name + ": 2000 multiplies, mismatching checks:", bad)
rows_w, width_w = 20000, 256
g = torch.Generator(device="cuda").manual_seed(1)
pitch_w = rows_w
Ajw = torch.randint(0, rows_w, (width_w * pitch_w,), dtype=torch.int32, device="cuda", generator=g)
Axw = torch.rand(width_w * pitch_w, dtype=torch.float64, device="cuda", generator=g)
xw = torch.rand(rows_w, dtype=torch.float64, device="cuda", generator=g)
ya, yb = torch.empty(rows_w, dtype=torch.float64, device="cuda"), torch.empty(rows_w, dtype=torch.float64, device="cuda")
cmi.spmv_ell(rows_w, rows_w, width_w, pitch_w, Ajw, Axw, xw, ya)
for _ in range(3000):
    cmi.spmv_ell(rows_w, rows_w, width_w, pitch_w, Ajw, Axw, xw, yb)
print("ell lanes per row (auto): 3000 multiplies identical:", bool(torch.equal(ya, yb)))
import os
os.environ["CMI_CG_FOLD_AHEAD"] = "1"
b = cmi.fill_x(n).cuda()
ref = None
for rep in range(5):
    xs0 = torch.zeros(n, dtype=torch.float64, device="cuda")
    mon = cmi.krylov.cg(A, xs0, b, iteration_limit=200, relative_tolerance=0.0)
    if ref is None:
        ref = (mon.residuals, xs0.clone())
    else:
        assert mon.residuals == ref[0] and torch.equal(xs0, ref[1]), "fold-ahead CG is not repeatable"
print("fold-ahead CG: 5 x 200 iterations, identical histories and solutions")

# round 3's kernels: wave-private vector-body tiles (csr_wavev V = 1 / 2 / 4) and the LDS x window (csr_wavex) on an irregular band matrix, the
# fenced fold of the reductions under the fused CG (default path, 5 x 300 iterations), the device COO sort
os.environ.pop("CMI_CG_FOLD_AHEAD", None)
g = torch.Generator(device="cuda").manual_seed(3)
rows_b = 2_000_000
lens_b = torch.randint(5, 40, (rows_b,), device="cuda", generator=g)
Apb = torch.zeros(rows_b + 1, dtype=torch.int32, device="cuda")
Apb[1:] = lens_b.cumsum(0).to(torch.int32)
nnz_b = int(Apb[-1])
row_b = torch.repeat_interleave(torch.arange(rows_b, device="cuda"), lens_b)
Ajb = ((row_b + torch.randint(-2000, 2001, (nnz_b,), device="cuda", generator=g)) % rows_b).to(torch.int32)
Axb = torch.randn(nnz_b, dtype=torch.float64, device="cuda", generator=g)
xb = torch.randn(rows_b, dtype=torch.float64, device="cuda", generator=g)
yref = torch.empty(rows_b, dtype=torch.float64, device="cuda")
cmi.spmv_csr(rows_b, rows_b, Apb, Ajb, Axb, xb, yref, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
yb2 = torch.empty_like(yref)
for name, cfg in (("csr_wavev V=1", cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=1)), ("csr_wavev V=2", cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=2)),
                  ("csr_wavev V=4", cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=4)),
                  ("csr_wavex V=4 window 2048", cmi.Config(kernel=cmi.CSR_STREAM_WAVEX, items_per_thread=4, rows_per_block=2048)), ("auto plan", None)):
    pl = cmi.Plan.csr(torch.float64, rows_b, rows_b, Apb, Ajb, cfg=cfg)
    bad = 0
    for _ in range(20):
        for _ in range(100):
            cmi.spmv_csr_plan(pl, Apb, Ajb, Axb, xb, yb2)
        bad += not torch.equal(yb2, yref)
    print(f"{name} (plan kernel {pl.config().kernel}): 2000 multiplies of a {rows_b}-row band matrix, checks that differ from csr_scalar's bits: {bad}")
ref = None
for rep in range(5):
    xs0 = torch.zeros(n, dtype=torch.float64, device="cuda")
    mon = cmi.krylov.cg(A, xs0, b, iteration_limit=300, relative_tolerance=0.0)
    if ref is None:
        ref = (mon.residuals, xs0.clone())
    else:
        assert mon.residuals == ref[0] and torch.equal(xs0, ref[1]), "fused CG (fenced folds) is not repeatable"
print("fused CG, fenced folds: 5 x 300 iterations on the headline matrix, identical histories and solutions")
C = cmi.convert(A, "coo")
perm = torch.randperm(C.num_entries, device="cuda", generator=g)
first = None
for rep in range(5):
    U = cmi.CooMatrix(C.num_rows, C.num_cols, C.num_entries, C.row_indices[perm].contiguous(), C.column_indices[perm].contiguous(), C.values[perm].contiguous())
    U.sort_by_row()
    cur = (U.row_indices.clone(), U.column_indices.clone(), U.values.clone())
    if first is None:
        first = cur
    else:
        assert all(torch.equal(a_, b_) for a_, b_ in zip(first, cur)), "device COO sort is not repeatable"
assert torch.equal(first[0], C.row_indices)
print("device COO sort: 5 x 50 M shuffled entries, identical results, rows as the sorted original's")

# round 4: the run-compressed copy (f64 and f32) and the packed tiles -- thousands of launches, the result never changes; plans made and
# destroyed in a loop (device memory owned by plans is returned)
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import suitesparse_like as ssl
Ap, Aj, Ax = ssl.ldoor_like(0.5)
rows_ = len(Ap) - 1
dAp, dAj = torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda()
for dt in (torch.float64, torch.float32):
    dAx = torch.from_numpy(Ax).cuda().to(dt)
    xx = cmi.fill_x(rows_, dt, "cuda")
    yy0, yy = torch.empty(rows_, dtype=dt, device="cuda"), torch.empty(rows_, dtype=dt, device="cuda")
    cmi.spmv_csr(rows_, rows_, dAp, dAj, dAx, xx, yy0, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
    for label, make in (("csr_waver (AUTO)", lambda: cmi.Plan.csr(dt, rows_, rows_, dAp, dAj)),
                        ("packed tiles", lambda: cmi.Plan.csr_values(rows_, rows_, dAp, dAj, dAx, cmi.Config(kernel=cmi.CSR_STREAM_PACKED)))):
        p = make()
        bad = 0
        for it in range(20):
            for _ in range(200):
                cmi.spmv_csr_plan(p, dAp, dAj, dAx, xx, yy)
            bad += 0 if torch.equal(yy, yy0) else 1
        free0 = torch.cuda.mem_get_info()[0]
        for _ in range(30):
            q = make()
            cmi.spmv_csr_plan(q, dAp, dAj, dAx, xx, yy)
            del q
        torch.cuda.synchronize()
        leaked = free0 - torch.cuda.mem_get_info()[0]
        print(f"{label} {str(dt)[6:]}: kernel {p.config().kernel}, 4000 multiplies, mismatching checks {bad}; 30 plans made and destroyed: {leaked / 1e6:.1f} MB not returned")
        del p
