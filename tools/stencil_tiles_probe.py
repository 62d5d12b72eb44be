#!/usr/bin/env python3
"""Stencil rows: csr_wave (64 rows per wave, K = longest row entries per lane, no plan memory) against the wave tiles on a plan-built
partition with the 16-byte-vector body (csr_wavev, V = 1 / 2) and -- where the stencil's columns come in runs -- the run-compressed copy.
tools/auto_regret.py (profiles/r04_auto_regret.txt) had csr_wavev V = 1 ahead of the AUTO plan's csr_wave by 4 % on the 5- and 7-point
matrices in f64 and the run-compressed copy ahead by 23-33 % on the 9-point one; this probe is the A/B that a rule change needs:
one process, plans interleaved round-robin, per variant

    replay   median of R rounds x L launches of the multiply on one (A, x, y)
    cold     the same over 4 copies of (A, x, y) in rotation (3+ GB between reuses: nothing comes from the Infinity Cache)
    dot      the multiply with the fused <y, x> (the CG instance), replayed
    cg       microseconds per iteration of cusp::krylov::cg's fused loop (python mirror, 150 iterations, tolerance 0) with the plan injected

    python3 tools/stencil_tiles_probe.py [--matrices 5pt,7pt,9pt,3pt,5pt32] [--rounds 5]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
import autotune as at  # noqa: E402
from cusp_autotuned_amd import krylov  # noqa: E402

P5 = [(0, -1, 0, -1.0), (-1, 0, 0, -1.0), (0, 0, 0, 4.0), (1, 0, 0, -1.0), (0, 1, 0, -1.0)]
MATS = {
    "5pt": ("poisson5pt 3162^2 f64 (headline)", lambda: at.stencil_csr(3162, 3162, 1, P5, np.float64), torch.float64),
    "5pt32": ("poisson5pt 3162^2 f32", lambda: at.stencil_csr(3162, 3162, 1, P5, np.float64), torch.float32),
    "7pt": ("7-point 215^3 f64", lambda: at.stencil_csr(215, 215, 215, at.stencil_points(7), np.float64), torch.float64),
    **{f"5pt_{m}": (f"poisson5pt {m}^2 f64", (lambda m=m: at.stencil_csr(m, m, 1, P5, np.float64)), torch.float64) for m in (1500, 2000, 2400, 2800)},  # the rule's size gate
    "5pt_10000x1250": ("poisson5pt 10000 x 1250 f64 (configs[4]'s rank block as a square matrix: same rows, same band, x of 1.25e7)", lambda: at.stencil_csr(10000, 1250, 1, P5, np.float64), torch.float64),
    "5pt_1250x10000": ("poisson5pt 1250 x 10000 f64 (the same rows with a band of 1250)", lambda: at.stencil_csr(1250, 10000, 1, P5, np.float64), torch.float64),
    "7pt32": ("7-point 215^3 f32", lambda: at.stencil_csr(215, 215, 215, at.stencil_points(7), np.float64), torch.float32),
    "9pt": ("9-point 3000^2 f64", lambda: at.stencil_csr(3000, 3000, 1, at.stencil_points(9), np.float64), torch.float64),
    "9pt32": ("9-point 3000^2 f32", lambda: at.stencil_csr(3000, 3000, 1, at.stencil_points(9), np.float64), torch.float32),
    "3pt": ("tridiagonal 10^7 f64", lambda: at.stencil_csr(10_000_000, 1, 1, [(-1, 0, 0, -1.0), (0, 0, 0, 2.0), (1, 0, 0, -1.0)], np.float64), torch.float64),
}


def group_us(go, launches):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(launches):
        go(i)
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / launches


def settle(go, seconds=0.05):
    t0 = time.time()
    i = 0
    while time.time() - t0 < seconds:
        go(i)
        i += 1
        if i % 64 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrices", default="5pt,7pt,9pt,3pt,5pt32,9pt32")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--launches", type=int, default=40)
    ap.add_argument("--cg-iterations", type=int, default=150)
    ap.add_argument("--sweep", action="store_true")
    args = ap.parse_args()
    for key in args.matrices.split(","):
        name, build, dt = MATS[key]
        Ap, Aj, Ax = build()
        rows, nnz = len(Ap) - 1, int(Ap[-1])
        vb = 8 if dt == torch.float64 else 4
        alg = cmi.csr_bytes(rows, nnz, vb)
        sets = []
        for _ in range(4):
            sets.append((torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda(), torch.from_numpy(Ax).cuda().to(dt), cmi.fill_x(rows, dt, "cuda"), torch.empty(rows, dtype=dt, device="cuda")))
        dAp, dAj, dAx, x, y = sets[0]
        cmi.spmv_csr(rows, rows, dAp, dAj, dAx, x, y, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
        want = y.clone()
        print(f"# {name}: rows {rows} entries {nnz}; algorithmic bytes {alg / 1e6:.1f} MB", flush=True)
        variants = [("AUTO plan", None), ("wave tiles V=1", cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=1)),
                    ("wave tiles V=2", cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=2))]
        if args.sweep:  # the wave tiles' cache policy (1 = nt loads, 2 = nt stores) x XCD dealing (chunks of that many workgroups; -1: launch order)
            variants = [("AUTO plan", None)] + [(f"wave tiles V=1 pol {pol} swz {swz}", cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=1, nontemporal=pol, xcd_swizzle=swz))
                                                 for pol in (3, 1, 2) for swz in (-1, 8, 16, 32, 64, 128, 256)]
        if key.startswith("9pt"):
            variants += [("run-compressed copy V=2", cmi.Config(kernel=cmi.CSR_STREAM_WAVER, items_per_thread=2)),
                         ("run-compressed copy V=4", cmi.Config(kernel=cmi.CSR_STREAM_WAVER, items_per_thread=4))]
        plans = {}
        for label, cfg in variants:
            p = cmi.Plan.csr(dt, rows, rows, dAp, dAj, cfg=cfg)
            y.fill_(float("nan"))
            cmi.spmv_csr_plan(p, dAp, dAj, dAx, x, y)
            assert torch.equal(y, want), label
            plans[label] = p
        res = torch.zeros(1, dtype=torch.float64, device="cuda")
        ws = cmi.blas_workspace()
        out = {label: {"replay": [], "cold": [], "dot": []} for label in plans}
        for _ in range(args.rounds):
            for label, p in plans.items():
                go = lambda i, p=p: cmi.spmv_csr_plan(p, dAp, dAj, dAx, x, y)  # noqa: E731
                settle(go)
                out[label]["replay"].append(group_us(go, args.launches))
                gc = lambda i, p=p: cmi.spmv_csr_plan(p, sets[i % 4][0], sets[i % 4][1], sets[i % 4][2], sets[i % 4][3], sets[i % 4][4])  # noqa: E731
                settle(gc)
                out[label]["cold"].append(group_us(gc, args.launches))
                gd = lambda i, p=p: cmi.spmv_csr_dot(rows, rows, dAp, dAj, dAx, x, y, x, res, ws, plan=p)  # noqa: E731
                settle(gd)
                out[label]["dot"].append(group_us(gd, args.launches))
        # the CG loop with the plan injected (the containers' plan slot)
        cg_us = {}
        for label, p in plans.items():
            A = cmi.CsrMatrix(rows, rows, nnz, dAp, dAj, dAx)
            A.plan()      # (sets the container's plan key for these arrays) ...
            A._plan = p   # ... and this is the plan its multiplies then run
            b = torch.ones(rows, dtype=dt, device="cuda")
            best = None
            for _ in range(3):
                xs = torch.zeros(rows, dtype=dt, device="cuda")
                torch.cuda.synchronize()
                t0 = time.time()
                mon = krylov.cg(A, xs, b, iteration_limit=args.cg_iterations, relative_tolerance=0.0)
                torch.cuda.synchronize()
                us = (time.time() - t0) * 1e6 / max(1, mon.iteration_count)
                best = us if best is None else min(best, us)
            cg_us[label] = best
        med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
        base = med(out["AUTO plan"]["replay"])
        for label in plans:
            r, c, d = med(out[label]["replay"]), med(out[label]["cold"]), med(out[label]["dot"])
            kc = plans[label].config()
            print(f"  {label:26s} kernel {kc.kernel:2d} V/K {kc.items_per_thread}: replay {r:7.1f} us ({alg / r / 8e6:.3f} of peak; x{r / base:.3f} of AUTO; rounds {' '.join(f'{v:.1f}' for v in out[label]['replay'])})  "
                  f"cold {c:7.1f} us ({alg / c / 8e6:.3f})  with the fused dot {d:7.1f} us  CG {cg_us[label]:7.1f} us/iteration  plan owns {plans[label].device_bytes() / 1e6:.1f} MB", flush=True)
        del sets, plans, dAp, dAj, dAx, x, y, want
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
