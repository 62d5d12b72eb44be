#!/usr/bin/env python3
"""CG iterations/s on the headline matrix (poisson5pt 3162x3162, CSR, fp64), plain vs fused driver.
SURVEY.md 8(d): "(d) CG iterations/s"; 8(f).1: fused vector updates.  One SpMV per iteration."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 3162
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
A = cmi.poisson5pt(m, m, "csr")
N = m * m
b = cmi.fill_x(N, device="cuda")
for fused in (False, True, False, True):
    x = torch.zeros(N, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mon = cmi.krylov.cg(A, x, b, iteration_limit=iters, relative_tolerance=0.0, fused=fused)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    spmv_bytes = cmi.csr_bytes(N, A.num_entries)
    vec = 8 * N
    passes = (spmv_bytes + (2 + 4 + 2 + 3) * vec * 1.0 + vec) if fused else (spmv_bytes + (2 + 3 + 3 + 2 + 2 + 3 + 1) * vec)
    print(f"fused={fused!s:5}  {mon.iteration_count} iterations in {dt * 1e3:8.1f} ms  = {mon.iteration_count / dt:8.0f} it/s, "
          f"{dt / mon.iteration_count * 1e6:7.1f} us/it, {passes / (dt / mon.iteration_count) / 1e12:5.2f} TB/s of the bytes each "
          f"driver touches; final ||r|| = {mon.residuals[-1]:.6e}")
