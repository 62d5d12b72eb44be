#!/usr/bin/env python3
"""configs[4]'s per-rank block (1.25e7 rows of poisson5pt(10000, 10000), global columns, x buffer of 1e8 entries): does the XCD dealing of
the wave tiles matter THERE?  On the headline matrix (grid lines of 3162 rows) it does not (profiles/r04_headline_wave_tiles_sweep.txt:
+-0.1 %); this block's rows reach 10000 columns to either side, so a chunk of C workgroups (C x ~205 rows) gathers from a window of
C x 205 + 20000 x entries -- with the table's C = 64 that is 2.5 x the chunk's own rows, fetched into that XCD's L2 again by the next chunk
on another XCD.  One process, plans interleaved: AUTO, wave tiles V = 1 with chunks of 64 ... 2048 workgroups and in launch order, csr_wave.

    python3 tools/rank_block_dealing_probe.py [--grid 10000] [--ranks 8] [--rank 4]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=10000)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--launches", type=int, default=30)
    args = ap.parse_args()
    g, N = args.grid, args.grid * args.grid
    lines = g // args.ranks
    lo, hi = args.rank * lines * g, (args.rank + 1) * lines * g
    A = cmi.poisson5pt(g, g, "csr", dtype=torch.float64, device="cuda", row_begin=lo, row_end=hi)
    rows, nnz = A.num_rows, A.num_entries
    x = cmi.fill_x(N, torch.float64, "cuda")
    y = torch.empty(rows, dtype=torch.float64, device="cuda")
    cmi.spmv_csr(rows, N, A.row_offsets, A.column_indices, A.values, x, y, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
    want = y.clone()
    alg = cmi.csr_bytes(rows, nnz)
    print(f"# rank {args.rank} of {args.ranks}: rows [{lo}, {hi}) of poisson5pt({g}, {g}): {rows} rows, {nnz} entries, x buffer {N * 8 / 1e6:.0f} MB; algorithmic bytes {alg / 1e6:.1f} MB", flush=True)
    variants = [("AUTO plan", None)]
    for swz in (-1, 16, 64, 128, 256, 512, 1024, 2048, 4096):
        variants.append((f"wave tiles V=1, chunks of {swz}" if swz > 0 else "wave tiles V=1, launch order", cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=1, nontemporal=3, xcd_swizzle=swz)))
    variants.append(("csr_wave (plan-less rule)", "planless"))
    plans = {}
    for label, cfg in variants:
        if cfg == "planless":
            plans[label] = None
            continue
        p = cmi.Plan.csr(torch.float64, rows, N, A.row_offsets, A.column_indices, cfg=cfg)
        y.fill_(float("nan"))
        cmi.spmv_csr_plan(p, A.row_offsets, A.column_indices, A.values, x, y)
        assert torch.equal(y, want), label
        plans[label] = p
    out = {k: [] for k in plans}
    for _ in range(args.rounds):
        for label, p in plans.items():
            if p is None:
                go = lambda: cmi.spmv_csr(rows, N, A.row_offsets, A.column_indices, A.values, x, y)  # noqa: E731
            else:
                go = lambda p=p: cmi.spmv_csr_plan(p, A.row_offsets, A.column_indices, A.values, x, y)  # noqa: E731
            t0 = time.time()
            n = 0
            while time.time() - t0 < 0.05:
                go()
                n += 1
                if n % 32 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.launches):
                go()
            b.record()
            b.synchronize()
            out[label].append(a.elapsed_time(b) * 1e3 / args.launches)
    for label, v in out.items():
        m = sorted(v)[len(v) // 2]
        c = plans[label].config() if plans[label] is not None else None
        print(f"  {label:36s} {m:7.1f} us = {alg / m / 8e6:.3f} of peak   rounds {' '.join(f'{t:.1f}' for t in v)}" + (f"   kernel {c.kernel} V {c.items_per_thread} pol {c.nontemporal} swz {c.xcd_swizzle}" if c else ""), flush=True)


if __name__ == "__main__":
    main()
