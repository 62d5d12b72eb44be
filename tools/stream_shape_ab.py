#!/usr/bin/env python3
"""csr_stream's two request shapes for the entry streams (16-byte vectors per lane / lane-strided, policy bit 4 of
cmi_config.nontemporal) across matrix kinds: the headline matrix in f64 and f32, the SuiteSparse-like stand-ins, 9- and 27-point
stencil-like banded matrices.  For each matrix the plan's resolved shape is kept and only the policy bits change; every variant is
checked bit for bit against the plan's own result before it is timed (interleaved rounds, median).

    python tools/stream_shape_ab.py [--quick]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402


def banded(n, offsets, dtype):
    """rows of len(offsets) entries at i + off (clipped): a stencil-like banded matrix as device CSR arrays"""
    i = torch.arange(n, device="cuda").view(n, 1)
    cols = i + torch.tensor(offsets, device="cuda").view(1, -1)
    ok = (cols >= 0) & (cols < n)
    lens = ok.sum(1)
    Ap = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
    Ap[1:] = lens.cumsum(0).to(torch.int32)
    Aj = cols[ok].to(torch.int32)
    g = torch.Generator(device="cuda").manual_seed(3)
    Ax = torch.randn(Aj.numel(), dtype=dtype, device="cuda", generator=g)
    return Ap, Aj, Ax


def time_us(fns, iters, rounds):
    for f in fns:
        for _ in range(3):
            f()
    out = [[] for _ in fns]
    for _ in range(rounds):
        for k, f in enumerate(fns):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            e1.synchronize()
            out[k].append(e0.elapsed_time(e1) * 1e3 / iters)
    return [sorted(o)[len(o) // 2] for o in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    import suitesparse_like as ssl
    cases = []
    for dt in (torch.float64, torch.float32):
        A = cmi.poisson5pt(3162, 3162, "csr", dtype=dt)
        cases.append((f"poisson5pt 3162^2 {str(dt)[6:]}", A.row_offsets, A.column_indices, A.values))
    if not args.quick:
        for name in ("thermal2", "ldoor", "nlpkkt120"):
            Ap, Aj, Ax, _ = ssl.load(name)
            cases.append((f"{name}-like f64", torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda(), torch.from_numpy(Ax).cuda()))
        m = 1500
        cases.append(("9-point banded 2.25e6 rows f64",) + banded(m * m, [-m - 1, -m, -m + 1, -1, 0, 1, m - 1, m, m + 1], torch.float64))
        g = 110
        off27 = [a * g * g + b * g + c for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]
        cases.append(("27-point banded 1.33e6 rows f64",) + banded(g ** 3, off27, torch.float64))
        cases.append(("27-point banded 1.33e6 rows f32",) + banded(g ** 3, off27, torch.float32))
        cases.append(("3 per row 1e7 rows f64",) + banded(10_000_000, [-1, 0, 1], torch.float64))
    for name, Ap, Aj, Ax in cases:
        N, nnz = Ap.numel() - 1, Aj.numel()
        dt = Ax.dtype
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(N, dtype=dt, device="cuda", generator=g)
        base = cmi.Plan.csr(dt, N, N, Ap, Aj)
        cfg = base.config()
        want = torch.empty(N, dtype=dt, device="cuda")
        cmi.spmv_csr_plan(base, Ap, Aj, Ax, x, want)
        print(f"{name}: {N} rows, {nnz / N:.2f} per row; plan {cfg}", flush=True)
        if cfg.kernel != cmi.CSR_STREAM:
            print("   (not csr_stream: skipped)")
            continue
        plans, labels = [], []
        for pol in (cfg.nontemporal & 3, 2, 3, 6, 7):
            c = cmi.Config(kernel=cfg.kernel, block_size=cfg.block_size, threads_per_row=cfg.threads_per_row, rows_per_block=cfg.rows_per_block,
                           items_per_thread=cfg.items_per_thread, nontemporal=pol, xcd_swizzle=cfg.xcd_swizzle, blocks_per_cu=cfg.blocks_per_cu)
            p = cmi.Plan.csr(dt, N, N, Ap, Aj, cfg=c)
            y = torch.full((N,), 7.0, dtype=dt, device="cuda")
            cmi.spmv_csr_plan(p, Ap, Aj, Ax, x, y)
            tag = "bit-exact" if torch.equal(y, want) else f"max diff {float((y - want).abs().max()):.3e}"
            plans.append((p, y))
            labels.append(f"policy {pol} ({tag})")
        fns = [(lambda p=p, y=y: cmi.spmv_csr_plan(p, Ap, Aj, Ax, x, y)) for p, y in plans]
        t = time_us(fns, args.iters, args.rounds)
        for lab, us in zip(labels, t):
            print(f"   {lab:34s} {us:8.1f} us   {us / t[0]:.3f} of the plan's")
        del plans, fns


if __name__ == "__main__":
    main()
