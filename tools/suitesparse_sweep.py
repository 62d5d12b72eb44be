#!/usr/bin/env python3
"""BASELINE.json configs[3]: the CSR threads-per-row sweep on the SuiteSparse irregular set (nlpkkt120, ldoor, thermal2),
replacing the reference's performance/csr_vector/csr_vector.cu:41-62,86-110 (THREADS_PER_VECTOR 2..32 of spmv_csr_vector
over the testing/UF downloads).  Per matrix -- the real file when CMI_SUITESPARSE_DIR has it, else the seeded stand-in of
tools/suitesparse_like.py, and the output says which -- every candidate is validated against csr_scalar (pinned bit for
bit to the reference host loop by tests/) and timed with HIP events in interleaved rounds:

  csr_scalar | csr_vector T = 2..64 (the reference's sweep; its own selector's T marked) | csr_stream with 1 / T lanes per row |
  the tuning table's choice (NULL config) | the plan's choice

    python tools/suitesparse_sweep.py [--scale 1.0] [--dtype f64] > archive/profiles/r02_suitesparse_like_sweep.txt
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--shapes", action="store_true", help="also time explicit csr_stream tile shapes (block, vectors per lane, rows per tile)")
    ap.add_argument("--only", default="", help="comma-separated subset of thermal2,ldoor,nlpkkt120")
    ap.add_argument("--policies", action="store_true", help="cache policy (nt loads / nt stores) x XCD dealing around the table's csr_stream shape")
    args = ap.parse_args()
    import torch
    import cusp_autotuned_amd as cmi
    import suitesparse_like as ssl

    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    vb = 8 if args.dtype == "f64" else 4
    lib = cmi.lib()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    cmi.check(lib.cmi_event_create(ctypes.byref(e0)))
    cmi.check(lib.cmi_event_create(ctypes.byref(e1)))

    def time_us(fn):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        cmi.check(lib.cmi_event_record(e0, s))
        for _ in range(args.iters):
            fn()
        cmi.check(lib.cmi_event_record(e1, s))
        ms = ctypes.c_float()
        cmi.check(lib.cmi_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        return ms.value / args.iters * 1e3

    for name in ("thermal2", "ldoor", "nlpkkt120"):
        if args.only and name not in args.only.split(","):
            continue
        Ap, Aj, Ax, source = ssl.load(name, args.scale)
        st = ssl.stats(Ap, Aj)
        rows, nnz = st["rows"], st["entries"]
        A = cmi.CsrMatrix(rows, rows, nnz, torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda(), torch.from_numpy(Ax).to(tdt).cuda())
        x = cmi.fill_x(rows, tdt, "cuda")
        y = torch.empty(rows, dtype=tdt, device="cuda")
        cmi.multiply(A, x, y, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
        want = y.clone()
        Aabs = cmi.CsrMatrix(rows, rows, nnz, A.row_offsets, A.column_indices, A.values.abs())
        cmi.multiply(Aabs, x.abs(), y, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
        bound = y.clone().clamp_(min=1e-30)
        alg = cmi.csr_bytes(rows, nnz, vb)
        mean = nnz / rows
        ref_T = 2 if mean <= 2 else 4 if mean <= 4 else 8 if mean <= 8 else 16 if mean <= 16 else 32  # csr_vector_spmv.h:241-256
        print(f"\n== {name}: {source}\n   {st}  published {ssl.PUBLISHED[name]}  dtype {args.dtype}  compulsory bytes {alg}")
        cands = [("csr_scalar", cmi.Config(kernel=cmi.CSR_SCALAR), True)]
        for T in (2, 4, 8, 16, 32, 64):
            cands.append((f"csr_vector T={T}" + ("  <- the reference selector's T" if T == ref_T else ""),
                          cmi.Config(kernel=cmi.CSR_VECTOR, threads_per_row=T), False))
        for T in (1, 2, 4, 8, 16, 32):
            for ipt in (1, 2):
                cands.append((f"csr_stream lanes/row={T} vectors/lane={ipt}", cmi.Config(kernel=cmi.CSR_STREAM, threads_per_row=T, items_per_thread=ipt,
                                                                                       nontemporal=2), T == 1))
        if args.shapes:  # explicit csr_stream tile shapes around the table's: block x vectors/lane, rows per tile by fill of the LDS pass
            for blk, ipt in ((256, 1), (256, 2), (512, 2), (256, 4)):
                fit = int((blk * ipt * 4 - 3) / mean)
                for r in sorted({max(16, fit // 16 * 16), max(16, fit // 16 * 16 - 16), max(16, fit // 8 * 8), min(fit, blk)}):
                    cands.append((f"csr_stream block {blk} vectors/lane={ipt} rows/tile={r} (fit {fit})",
                                  cmi.Config(kernel=cmi.CSR_STREAM, block_size=blk, threads_per_row=1, items_per_thread=ipt, rows_per_block=r, nontemporal=2), True))
        if args.policies:  # the table's shape under every cache policy and a few XCD dealings (the table's entry was tuned on other matrices)
            t = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64 if vb == 8 else cmi.F32, rows, rows, nnz)
            for pol in (0, 1, 2, 3):
                for swz in (0, 1, 4, 16, 64):
                    cands.append((f"csr_stream table shape, policy {pol} (1 = nt loads, 2 = nt stores), XCD dealing {swz}",
                                  cmi.Config(kernel=cmi.CSR_STREAM, block_size=t.block_size, threads_per_row=1, items_per_thread=t.items_per_thread,
                                             rows_per_block=t.rows_per_block, nontemporal=pol, xcd_swizzle=swz), True))
        # round 3: the wave-private vector-body kernels run through plans made for them (a plan refuses a shape whose tile cannot hold the longest row)
        planned = {}
        for label, c in ([(f"csr_wavev V={v} (wave-private tiles, one line per load instruction; through a plan)", cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=v)) for v in (1, 2, 4)]
                         + [(f"csr_wavex V=4, x window of {w} entries in LDS (through a plan)", cmi.Config(kernel=cmi.CSR_STREAM_WAVEX, items_per_thread=4, rows_per_block=w)) for w in (2048, 4096)]):
            try:
                planned[label] = cmi.Plan.csr(tdt, rows, rows, A.row_offsets, A.column_indices, cfg=c)
                cands.append((label, ("planned", label), True))
            except Exception as e:  # noqa: BLE001
                print(f"   (no plan for {label}: {e})")
        plan = A.plan()
        cands.append((f"table (NULL config): {cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64 if vb == 8 else cmi.F32, rows, rows, nnz)}", None, False))
        runs = []
        for label, cfg, exact in cands:
            y.fill_(10.0)
            if cfg is None:
                cmi.spmv_csr(rows, rows, A.row_offsets, A.column_indices, A.values, x, y)
            elif isinstance(cfg, tuple):
                cmi.spmv_csr_plan(planned[cfg[1]], A.row_offsets, A.column_indices, A.values, x, y)
            else:
                cmi.multiply(A, x, y, cfg=cfg)
            ok = bool(torch.equal(y, want)) if exact else bool(((y - want).abs() <= (1e-6 if vb == 8 else 1e-5) * bound).all().item())
            runs.append((label, cfg, ok))
        runs.append((f"plan: {plan.config()} {plan.info()}", "plan", True))
        times = {l: [] for l, _, _ in runs}
        for _ in range(args.rounds):
            for label, cfg, ok in runs:
                if not ok:
                    continue
                if cfg == "plan":
                    fn = lambda: cmi.multiply(A, x, y)  # noqa: E731
                elif cfg is None:
                    fn = lambda: cmi.spmv_csr(rows, rows, A.row_offsets, A.column_indices, A.values, x, y)  # noqa: E731
                elif isinstance(cfg, tuple):
                    fn = lambda q=planned[cfg[1]]: cmi.spmv_csr_plan(q, A.row_offsets, A.column_indices, A.values, x, y)  # noqa: E731
                else:
                    fn = lambda c=cfg: cmi.multiply(A, x, y, cfg=c)  # noqa: E731
                times[label].append(time_us(fn))
        best = min((np.median(t) for t in times.values() if t), default=float("nan"))
        for label, cfg, ok in runs:
            if not ok:
                print(f"   {'VALIDATION FAILED':>9s}            {label}")
                continue
            t = float(np.median(times[label]))
            print(f"   {t:9.1f} us  {alg / t / 1e3:7.0f} GB/s  {2 * nnz / t / 1e3:7.1f} GFLOP/s  {'*' if t == best else ' '} {label}")
        del A, Aabs, x, y, want, bound
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
