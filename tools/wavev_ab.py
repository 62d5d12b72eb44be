#!/usr/bin/env python3
"""csr_wavev (CMI_CSR_STREAM_WAVEV: wave-private tiles, one-line-per-instruction vector body) against what an AUTO plan runs today,
same process, interleaved rounds, every variant checked bit for bit against the plan's result first -- the measurement behind the
plan's auto rule (plan.hip wavev_auto).  Matrix zoo: the three configs[3] stand-ins at full size, f64 and f32; seeded irregular
matrices of 3..60 entries per row with banded and with SCATTERED columns (where round 2's lane-strided wave partition lost), large
and cache-resident; 9- / 27-point banded stencil-like matrices.

    python tools/wavev_ab.py [--quick]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
from stream_shape_ab import banded, time_us  # noqa: E402


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
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    import suitesparse_like as ssl
    cases = []
    for name in ("thermal2", "ldoor", "nlpkkt120"):
        Ap, Aj, Ax, _ = ssl.load(name, 0.05 if args.quick else 1.0)
        for dt, nd in ((torch.float64, np.float64), (torch.float32, np.float32)):
            cases.append((f"{name}-like {str(dt)[6:]}", torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda(), torch.from_numpy(Ax.astype(nd)).cuda()))
    # an unstructured FEM matrix BEYOND the Infinity Cache: sorted columns, neighbours rarely adjacent (few shared x lines inside a row) but
    # shared between neighbouring rows -- the case an LDS x window must not be switched on for by a careless rule
    Ap, Aj, Ax, _ = ssl.load("thermal2", 0.3 if args.quick else 6.0)
    for dt, nd in ((torch.float64, np.float64), (torch.float32, np.float32)):
        cases.append((f"thermal2-like x6 {str(dt)[6:]}", torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda(), torch.from_numpy(Ax.astype(nd)).cuda()))
    s = 0.1 if args.quick else 1.0
    for dt in (torch.float64, torch.float32):
        tag = str(dt)[6:]
        cases.append((f"2..8 per row, 6e6 rows, band 2000 {tag}",) + irregular(int(6e6 * s), 2, 8, 2000, 2, dt))
        cases.append((f"5..12 per row, 4e6 rows, band 2000 {tag}",) + irregular(int(4e6 * s), 5, 12, 2000, 4, dt))
        cases.append((f"10..30 per row, 2e6 rows, band 2000 {tag}",) + irregular(int(2e6 * s), 10, 30, 2000, 6, dt))
        cases.append((f"20..60 per row, 1e6 rows, band 5000 {tag}",) + irregular(int(1e6 * s), 20, 60, 5000, 7, dt))
        cases.append((f"10..30 per row, 2e6 rows, SCATTERED columns {tag}",) + irregular(int(2e6 * s), 10, 30, int(1e6 * s) - 1, 8, dt))
        cases.append((f"2..8 per row, 6e6 rows, SCATTERED columns {tag}",) + irregular(int(6e6 * s), 2, 8, int(3e6 * s) - 1, 9, dt))
        cases.append((f"10..30 per row, 3e5 rows (fits the Infinity Cache) {tag}",) + irregular(int(3e5 * s) + 5000, 10, 30, 2000, 10, dt))
        cases.append((f"27-point-like banded, 3e6 rows {tag}",) + banded(int(3e6 * s), [d + e for d in (-90000, -300, 0, 300, 90000) for e in (-301, -300, -299, -1, 0, 1)][:27], dt))
    for name, Ap, Aj, Ax in cases:
        N, nnz = Ap.numel() - 1, Aj.numel()
        dt = Ax.dtype
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(N, dtype=dt, device="cuda", generator=g)
        auto = cmi.Plan.csr(dt, N, N, Ap, Aj)
        y0 = torch.full((N,), 7.0, dtype=dt, device="cuda")
        cmi.spmv_csr_plan(auto, Ap, Aj, Ax, x, y0)
        fns, labels = [lambda: cmi.spmv_csr_plan(auto, Ap, Aj, Ax, x, y0)], ["auto plan"]
        plans, ys, same = [], [], []
        yt = torch.full((N,), 5.0, dtype=dt, device="cuda")
        fns.append(lambda: cmi.spmv_csr(N, N, Ap, Aj, Ax, x, yt))   # no plan: the table's csr_stream entry
        labels.append("table csr_stream")
        for v in (1, 2, 4):
            try:
                p = cmi.Plan.csr(dt, N, N, Ap, Aj, cfg=cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=v))
            except cmi.CmiError:
                continue
            y = torch.full((N,), 9.0, dtype=dt, device="cuda")
            cmi.spmv_csr_plan(p, Ap, Aj, Ax, x, y)
            plans.append(p); ys.append(y); same.append(bool(torch.equal(y, y0)))
            fns.append(lambda p=p, y=y: cmi.spmv_csr_plan(p, Ap, Aj, Ax, x, y))
            labels.append(f"wavev V={v} policy {p.config().nontemporal}")
        if hasattr(cmi, "CSR_STREAM_WAVEX"):
            for v, win in ((2, 0), (4, 0), (4, 2048)):
                try:
                    p = cmi.Plan.csr(dt, N, N, Ap, Aj, cfg=cmi.Config(kernel=cmi.CSR_STREAM_WAVEX, items_per_thread=v, rows_per_block=win))
                except cmi.CmiError:
                    continue
                y = torch.full((N,), 9.0, dtype=dt, device="cuda")
                cmi.spmv_csr_plan(p, Ap, Aj, Ax, x, y)
                plans.append(p); ys.append(y); same.append(bool(torch.equal(y, y0)))
                fns.append(lambda p=p, y=y: cmi.spmv_csr_plan(p, Ap, Aj, Ax, x, y))
                labels.append(f"wavex V={v} window {win or 4096}")
        t = time_us(fns, 30, 5)
        ca = auto.config()
        alg = cmi.csr_bytes(N, nnz, 8 if dt == torch.float64 else 4)
        best = min(range(2, len(t)), key=lambda k: t[k]) if len(t) > 2 else 0
        lens = (Ap[1:] - Ap[:-1])
        # the two column statistics a plan made with the columns can measure (plan.hip): entries within 1536 columns of their row's
        # diagonal position, and entries 16+ columns away from their predecessor in the row
        rowid = torch.repeat_interleave(torch.arange(N, device="cuda"), lens.long())
        inside = float(((Aj.long() - rowid).abs() < 1536).double().mean())
        first = torch.zeros(nnz, dtype=torch.bool, device="cuda")
        first[Ap[:-1][lens > 0].long()] = True
        gap = (Aj[1:].long() - Aj[:-1].long()).abs() >= 16
        jumps = float((gap & ~first[1:]).double().sum()) / max(nnz, 1)
        del rowid, first, gap
        print(f"[inside {inside:.2f} jumps {jumps:.2f}] " + f"{name}: {N} rows, {nnz / N:.2f} per row (max {int(lens.max())}), {alg / 1e6:.0f} MB | auto plan kernel {ca.kernel} block {ca.block_size} rpb {ca.rows_per_block} "
              f"ipt {ca.items_per_thread} policy {ca.nontemporal}: {t[0]:7.1f} us ({alg / t[0] / 8e6:.3f}) | " +
              " | ".join(f"{labels[k]}: {t[k]:7.1f} us" for k in range(1, len(t))) +
              (f" | best wavev / auto = {t[best] / t[0]:.3f}" if best else "") + (" | same bits" if all(same) else " | DIFFERENT BITS"), flush=True)
        del auto, plans, ys


if __name__ == "__main__":
    main()
