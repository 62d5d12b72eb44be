#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (run under the profiler, one counter set per pass):
launches the dominant SpMV kernel of bench.py (same matrix, same tuning-table config) N times, and a
CALIBRATION kernel with an exactly known byte count in a comparable access pattern (axpby: 16-byte
per-lane streaming reads of 2 vectors, 16-byte streaming writes of 1), so that FETCH_SIZE /
WRITE_SIZE can be corrected as MI355X_MICROARCH.md (HBM section) prescribes before they are compared
with algorithmic bytes.  Prints the known byte counts as JSON on the last line.

    rocprofv3 --pmc FETCH_SIZE  --kernel-trace --output-format csv -d <dir> -o fetch -- python3 tools/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE  --kernel-trace --output-format csv -d <dir> -o write -- python3 tools/pmc_probe.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402

fmts = (sys.argv[1] if len(sys.argv) > 1 else "csr").split(",")  # several formats in one process: one profiler start-up
M = 3162
N = M * M
A = cmi.poisson5pt(M, M, "csr")
x = cmi.fill_x(N, device="cuda")
y = torch.empty(N, dtype=torch.float64, device="cuda")
alg = {}
for fmt in fmts:
    # hyb: width 4 leaves the fifth entry of the interior rows to the COO part (a width of 5 would time ELL alone)
    if fmt == "csr16":  # the opt-in plan with the 16-bit column copy (CMI_CSR_STREAM_C16): same matrix, same arrays
        Afmt = cmi.CsrMatrix(N, N, A.num_entries, A.row_offsets, A.column_indices, A.values)
        assert Afmt.plan(compress=True).config().kernel == cmi.CSR_STREAM_C16
    elif fmt == "csr16p":  # round 4: the opt-in PACKED wave tiles (CMI_CSR_STREAM_PACKED): one span per tile of 64 rows
        pk = cmi.Plan.csr_values(N, N, A.row_offsets, A.column_indices, A.values, cmi.Config(kernel=cmi.CSR_STREAM_PACKED))
        assert pk.config().kernel == cmi.CSR_STREAM_PACKED
        packed_bytes = pk.device_bytes()
        Afmt = None  # (multiplied through the plan below: the containers do not make packed plans by themselves)
    else:
        Afmt = A if fmt == "csr" else (cmi.poisson5pt(M, M, "dia") if fmt == "dia" else
                                       cmi.convert(A, fmt, num_entries_per_row=4 if fmt == "hyb" else None))
    torch.cuda.synchronize()
    for _ in range(10):
        if fmt == "csr16p":
            cmi.spmv_csr_plan(pk, A.row_offsets, A.column_indices, A.values, x, y)
        else:
            cmi.multiply(Afmt, x, y)
    torch.cuda.synchronize()
    alg[fmt] = {"csr": cmi.csr_bytes(N, A.num_entries), "csr16p": (packed_bytes + 16 * N) if fmt == "csr16p" else 0, "csr16": cmi.csr_bytes(N, A.num_entries) - 2 * A.num_entries, "ell": cmi.ell_bytes(N, 5, 9998272), "dia": cmi.dia_bytes(N, 5, N),
                "coo": cmi.coo_bytes(N, A.num_entries),
                "hyb": cmi.ell_bytes(N, 4, 9998272) + 16 * (A.num_entries - 4 * N if fmt != "hyb" else Afmt.coo.num_entries)}[fmt]
    del Afmt
# calibration: z = 2x + 3y over 2^25 doubles (256 MiB per vector: far beyond the 256 MiB Infinity Cache in total)
n_cal = 1 << 25
a = torch.ones(n_cal, dtype=torch.float64, device="cuda")
b = torch.ones(n_cal, dtype=torch.float64, device="cuda")
c = torch.empty(n_cal, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for _ in range(10):
    cmi.blas_axpby(2.0, a, 3.0, b, c)
torch.cuda.synchronize()
print(json.dumps({"formats": fmts, "spmv_algorithmic_bytes": alg, "spmv_write_bytes": 8 * N,
                  "calibration_kernel": "axpby_kernel", "calibration_read_bytes": 16 * n_cal,
                  "calibration_write_bytes": 8 * n_cal}))
