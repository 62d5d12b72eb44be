#!/usr/bin/env python3
"""Launch-shape sweep for the opt-in 16-bit-column plan (CMI_CSR_STREAM_C16, csrc/spmv_csr16.hip) on the headline matrix
(and, with --matrix, the SuiteSparse-like stand-ins that qualify): block x vectors per lane x rows per tile x cache policy x
XCD dealing, each validated bit for bit against the plain plan's result before it is timed.  Prints the ranking and the best
shape next to the plain kernel's time.

    python tools/c16_sweep.py [--matrix poisson|ldoor] [--quick]
"""
import argparse
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402


def time_us(fn, iters, rounds):
    for _ in range(5):
        fn()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / iters)
    return sorted(out)[len(out) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="poisson")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    if args.matrix == "poisson":
        A = cmi.poisson5pt(3162, 3162, "csr")
    else:
        import suitesparse_like as ssl
        Ap, Aj, Ax, src = ssl.load(args.matrix)
        print(src)
        A = cmi.CsrMatrix(len(Ap) - 1, len(Ap) - 1, len(Aj), torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda(), torch.from_numpy(Ax).cuda())
    N, nnz = A.num_rows, A.num_entries
    mean = nnz / N
    x = cmi.fill_x(N, device="cuda")
    y = torch.empty(N, dtype=torch.float64, device="cuda")
    want = torch.empty_like(y)
    cmi.multiply(A, x, want)
    plain = time_us(lambda: cmi.multiply(A, x, y), args.iters, 5)
    print(f"{args.matrix}: {N} rows, {nnz} entries ({mean:.2f}/row); plain plan {A.plan().config()}: {plain:.1f} us")
    moved = 10 * nnz + 20 * N
    results = []
    blocks = (128, 256, 512) if not args.quick else (256,)
    for blk, ipt in itertools.product(blocks, (1, 2, 4)):
        fit = int((blk * ipt * 4 - 3) / mean)
        cands = sorted({r for r in (fit // 16 * 16, fit // 16 * 16 - 16, fit // 64 * 64, blk, blk // 2, (blk * 3) // 4) if 1 <= r <= min(blk, fit)})
        for rpb in cands:
            for nt, swz in itertools.product((2, 3), (0, 16, 32, 64, 128) if not args.quick else (32, 64)):
                cfg = cmi.Config(kernel=cmi.CSR_STREAM_C16, block_size=blk, items_per_thread=ipt, rows_per_block=rpb, nontemporal=nt, xcd_swizzle=swz)
                p = cmi.Plan.csr(torch.float64, N, N, A.row_offsets, A.column_indices, cfg=cfg)
                if p.config().kernel != cmi.CSR_STREAM_C16:
                    break
                y.fill_(7.0)
                cmi.spmv_csr_plan(p, A.row_offsets, A.column_indices, A.values, x, y)
                if not torch.equal(y, want):
                    print("VALIDATION FAILED", cfg)
                    continue
                t = time_us(lambda: cmi.spmv_csr_plan(p, A.row_offsets, A.column_indices, A.values, x, y), args.iters, args.rounds)
                results.append((t, blk, ipt, rpb, nt, swz))
                del p
    results.sort()
    for t, blk, ipt, rpb, nt, swz in results[:15]:
        print(f"  {t:7.1f} us  {moved / t / 1e3:6.0f} GB/s moved  block {blk} vectors/lane {ipt} rows/tile {rpb} nt {nt} swizzle {swz}")
    print(f"  ... {len(results)} shapes; slowest {results[-1][0]:.1f} us")
    auto = cmi.Plan.csr(torch.float64, N, N, A.row_offsets, A.column_indices, cfg=cmi.Config(kernel=cmi.CSR_STREAM_C16))
    t = time_us(lambda: cmi.spmv_csr_plan(auto, A.row_offsets, A.column_indices, A.values, x, y), args.iters, 5)
    print(f"  table-derived shape {auto.config()}: {t:.1f} us; best/plain = {results[0][0] / plain:.3f}")


if __name__ == "__main__":
    main()
