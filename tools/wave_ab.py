#!/usr/bin/env python3
"""The wave-tile kernel a CSR plan selects for stencil-like rows (CMI_CSR_STREAM_WAVE) against the table's csr_stream entry it
replaces, same process, interleaved rounds: 5-point (headline), 3-per-row, 7-point 3-D, 9-point 2-D stencils, f64 and f32.  Every
result is checked bit for bit against the other kernel's.

    python tools/wave_ab.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
from stream_shape_ab import banded, time_us  # noqa: E402


def main():
    cases = []
    for dt in (torch.float64, torch.float32):
        tag = str(dt)[6:]
        A = cmi.poisson5pt(3162, 3162, "csr", dtype=dt)
        cases.append((f"5-point 3162^2 {tag}", A.row_offsets, A.column_indices, A.values))
        cases.append((f"3 per row 1e7 rows {tag}",) + banded(10_000_000, [-1, 0, 1], dt))
        g = 180
        cases.append((f"7-point 180^3 {tag}",) + banded(g ** 3, [-g * g, -g, -1, 0, 1, g, g * g], dt))
        m = 2000
        cases.append((f"9-point 2000^2 {tag}",) + banded(m * m, [-m - 1, -m, -m + 1, -1, 0, 1, m - 1, m, m + 1], dt))
        cases.append((f"5-point 1000^2 {tag} (fits the Infinity Cache)",) + banded(1_000_000, [-1000, -1, 0, 1, 1000], dt))
    for name, Ap, Aj, Ax in cases:
        N, nnz = Ap.numel() - 1, Aj.numel()
        dt = Ax.dtype
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(N, dtype=dt, device="cuda", generator=g)
        auto = cmi.Plan.csr(dt, N, N, Ap, Aj)
        table = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64 if dt == torch.float64 else cmi.F32, N, N, nnz)
        stream = cmi.Plan.csr(dt, N, N, Ap, Aj, cfg=table)
        y1 = torch.full((N,), 7.0, dtype=dt, device="cuda")
        y2 = torch.full((N,), 9.0, dtype=dt, device="cuda")
        cmi.spmv_csr_plan(auto, Ap, Aj, Ax, x, y1)
        cmi.spmv_csr_plan(stream, Ap, Aj, Ax, x, y2)
        same = torch.equal(y1, y2)
        t = time_us([lambda: cmi.spmv_csr_plan(auto, Ap, Aj, Ax, x, y1), lambda: cmi.spmv_csr_plan(stream, Ap, Aj, Ax, x, y2)], 30, 5)
        ca, cs = auto.config(), stream.config()
        print(f"{name}: {N} rows, {nnz / N:.2f} per row | plan: kernel {ca.kernel} k {ca.items_per_thread} policy {ca.nontemporal} swz {ca.xcd_swizzle}: {t[0]:7.1f} us | "
              f"table csr_stream block {cs.block_size} rpb {cs.rows_per_block} policy {cs.nontemporal} swz {cs.xcd_swizzle}: {t[1]:7.1f} us | ratio {t[0] / t[1]:.3f} | "
              f"{'same bits' if same else 'DIFFERENT'}", flush=True)


if __name__ == "__main__":
    main()
