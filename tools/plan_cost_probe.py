#!/usr/bin/env python3
"""What a plan costs to make (one synchronising call per matrix) against the multiply it steers: headline matrix, every plan kind.

    python tools/plan_cost_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402

A = cmi.poisson5pt(3162, 3162, "csr")
N, nnz = A.num_rows, A.num_entries
C = cmi.convert(A, "coo")
H = cmi.convert(A, "hyb", num_entries_per_row=4)
H1 = cmi.convert(A, "hyb", num_entries_per_row=1)
x = cmi.fill_x(N, device="cuda")
y = torch.empty(N, dtype=torch.float64, device="cuda")


def wall_us(make, reps=10):
    make()  # first call: lazy module / allocator effects
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        p = make()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e6)
        del p
    return sorted(out)[len(out) // 2]


def mult_us(fn, iters=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


rows = [
    ("CSR (row-length profile)", lambda: cmi.Plan(cmi.FORMAT_CSR, torch.float64, N, N, nnz, A.row_offsets), lambda: cmi.multiply(A, x, y)),
    ("CSR with the columns (what the containers make: + partition, + a look at the columns)", lambda: cmi.Plan.csr(torch.float64, N, N, A.row_offsets, A.column_indices), lambda: cmi.multiply(A, x, y)),
    ("CSR + 16-bit column copy", lambda: cmi.Plan.csr(torch.float64, N, N, A.row_offsets, A.column_indices, cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16)), None),
    ("COO sorted (row offsets + CSR sub-plan)", lambda: cmi.Plan(cmi.FORMAT_COO, torch.float64, N, N, nnz, C.row_indices), lambda: cmi.multiply(C, x, y)),
    ("HYB K=4 (order check + tile ranges)", lambda: cmi.Plan.hyb(torch.float64, N, N, 4, H.coo.row_indices), lambda: cmi.multiply(H, x, y)),
    ("HYB K=1 (order check + COO sub-plan)", lambda: cmi.Plan.hyb(torch.float64, N, N, 1, H1.coo.row_indices), lambda: cmi.multiply(H1, x, y)),
]
print(f"poisson5pt 3162x3162 fp64: plan creation (wall, incl. its synchronisation, median of 10) vs one multiply through it")
for name, make, mult in rows:
    t = wall_us(make)
    if mult is None:
        p = make()
        mult = lambda p=p: cmi.spmv_csr_plan(p, A.row_offsets, A.column_indices, A.values, x, y)  # noqa: E731
    m = mult_us(mult)
    print(f"  {name:42s} {t:9.0f} us to create   {m:7.1f} us per multiply   = {t / m:5.1f} multiplies")

# round 4: the plans that own a copy derived from the columns / the values, on the matrices they are made for
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import numpy as np  # noqa: E402
import suitesparse_like as ssl  # noqa: E402

p = cmi.Plan.csr_values(N, N, A.row_offsets, A.column_indices, A.values, cmi.Config(kernel=cmi.CSR_STREAM_PACKED))
t = wall_us(lambda: cmi.Plan.csr_values(N, N, A.row_offsets, A.column_indices, A.values, cmi.Config(kernel=cmi.CSR_STREAM_PACKED)))
m = mult_us(lambda: cmi.spmv_csr_plan(p, A.row_offsets, A.column_indices, A.values, x, y))
print(f"  {'CSR packed wave tiles (16-bit copy + pack)':42s} {t:9.0f} us to create   {m:7.1f} us per multiply   = {t / m:5.1f} multiplies   (owns {p.device_bytes() / 1e6:.0f} MB)")
del p, A, C, H, H1
torch.cuda.empty_cache()
for name in ("ldoor", "nlpkkt120"):
    Ap, Aj, Ax = ssl.GENERATORS[name](1.0)
    rows_, nnz_ = len(Ap) - 1, len(Aj)
    dAp, dAj, dAx = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (Ap, Aj, Ax))
    xx = cmi.fill_x(rows_, device="cuda")
    yy = torch.empty(rows_, dtype=torch.float64, device="cuda")
    for label, make in (("AUTO with the columns (csr_waver)", lambda: cmi.Plan.csr(torch.float64, rows_, rows_, dAp, dAj)),
                        ("row offsets only (csr_stream / csr_wavev)", lambda: cmi.Plan(cmi.FORMAT_CSR, torch.float64, rows_, rows_, nnz_, dAp)),
                        ("packed run-compressed tiles", lambda: cmi.Plan.csr_values(rows_, rows_, dAp, dAj, dAx, cmi.Config(kernel=cmi.CSR_STREAM_PACKED)))):
        p = make()
        t = wall_us(make, reps=5)
        m = mult_us(lambda: cmi.spmv_csr_plan(p, dAp, dAj, dAx, xx, yy), iters=100)
        print(f"  {name + '-like: ' + label:58s} {t:9.0f} us to create   {m:7.1f} us per multiply   = {t / m:5.1f} multiplies   (kernel {p.config().kernel}, owns {p.device_bytes() / 1e6:.1f} MB)")
        del p
    del dAp, dAj, dAx, xx, yy
    torch.cuda.empty_cache()
