#!/usr/bin/env python3
"""The wave-tile kernel on a plan-built partition (irregular short rows; opt-in: kernel CMI_CSR_STREAM_WAVE, rows_per_block < 0) against the table's csr_stream entry it replaces, same process,
interleaved rounds: thermal2-like, seeded matrices of 2..9 entries per row on average (large and cache-resident), f64 and f32.

    python tools/wavep_ab.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
from stream_shape_ab import time_us  # noqa: E402


def irregular(rows, lo, hi, band, seed, dt):
    g = torch.Generator(device="cuda").manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (rows,), device="cuda", generator=g)
    Ap = torch.zeros(rows + 1, dtype=torch.int32, device="cuda")
    Ap[1:] = lens.cumsum(0).to(torch.int32)
    nnz = int(Ap[-1])
    row = torch.repeat_interleave(torch.arange(rows, device="cuda"), lens)
    Aj = ((row + torch.randint(-band, band + 1, (nnz,), device="cuda", generator=g)) % rows).to(torch.int32)
    Ax = torch.randn(nnz, dtype=dt, device="cuda", generator=g)
    return Ap, Aj, Ax


def main():
    import suitesparse_like as ssl
    cases = []
    Ap, Aj, Ax, _ = ssl.load("thermal2")
    cases.append(("thermal2-like f64", torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda(), torch.from_numpy(Ax).cuda()))
    cases.append(("thermal2-like f32", torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda(), torch.from_numpy(Ax.astype(np.float32)).cuda()))
    for dt in (torch.float64, torch.float32):
        tag = str(dt)[6:]
        cases.append((f"1..5 per row, 8e6 rows {tag}",) + irregular(8_000_000, 1, 5, 2000, 1, dt))
        cases.append((f"2..8 per row, 6e6 rows {tag}",) + irregular(6_000_000, 2, 8, 2000, 2, dt))
        cases.append((f"0..12 per row, 5e6 rows {tag}",) + irregular(5_000_000, 0, 12, 2000, 3, dt))
        cases.append((f"5..12 per row, 4e6 rows {tag}",) + irregular(4_000_000, 5, 12, 2000, 4, dt))
        cases.append((f"2..8 per row, 1e6 rows {tag} (fits the Infinity Cache)",) + irregular(1_000_000, 2, 8, 2000, 5, dt))
    for name, Ap, Aj, Ax in cases:
        N, nnz = Ap.numel() - 1, Aj.numel()
        dt = Ax.dtype
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(N, dtype=dt, device="cuda", generator=g)
        auto = cmi.Plan.csr(dt, N, N, Ap, Aj, cfg=cmi.Config(kernel=cmi.CSR_STREAM_WAVE, rows_per_block=-1))  # asked for: no plan selects it by itself
        table = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64 if dt == torch.float64 else cmi.F32, N, N, nnz)
        stream = cmi.Plan.csr(dt, N, N, Ap, Aj, cfg=table)
        y1 = torch.full((N,), 7.0, dtype=dt, device="cuda")
        y2 = torch.full((N,), 9.0, dtype=dt, device="cuda")
        cmi.spmv_csr_plan(auto, Ap, Aj, Ax, x, y1)
        cmi.spmv_csr_plan(stream, Ap, Aj, Ax, x, y2)
        same = torch.equal(y1, y2)
        t = time_us([lambda: cmi.spmv_csr_plan(auto, Ap, Aj, Ax, x, y1), lambda: cmi.spmv_csr_plan(stream, Ap, Aj, Ax, x, y2)], 30, 5)
        ca, cs = auto.config(), stream.config()
        print(f"{name}: {N} rows, {nnz / N:.2f} per row | plan: kernel {ca.kernel} k {ca.items_per_thread} rpb {ca.rows_per_block} policy {ca.nontemporal} swz {ca.xcd_swizzle}: {t[0]:7.1f} us | "
              f"table csr_stream block {cs.block_size} rpb {cs.rows_per_block} policy {cs.nontemporal} swz {cs.xcd_swizzle}: {t[1]:7.1f} us | ratio {t[0] / t[1]:.3f} | "
              f"{'same bits' if same else 'DIFFERENT'}", flush=True)
        del auto, stream


if __name__ == "__main__":
    main()
