#!/usr/bin/env python3
"""HYB in one launch or in two?  For the row-length distributions of tools/autotune_hyb.py and a few widths K each, times the
multiply with CMI_HYB_ONE_LAUNCH=1 (hyb_tile_kernel: ELL slots, then the tile's COO entries 256 at a time) and =0 (ELL kernel,
then the entry-tiled COO kernel accumulating) and prints them beside the COO part's weight (entries per row, most entries in
one 256-row tile) and what the plan's own rule picks (csrc/common.h kHybFusedMax*).  Both results are validated against
csr_scalar's.

    python tools/hyb_fuse_probe.py [--quick]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
import autotune_hyb as ah  # noqa: E402


def time_us(fn, iters=20, rounds=3):
    for _ in range(3):
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
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    scalar = cmi.Config(kernel=cmi.CSR_SCALAR)
    print(f"{'matrix':28s} {'K':>4s} {'coo/row':>8s} {'max/tile':>9s} | {'one launch':>11s} {'two':>9s}  ratio | plan picks")
    mats = [("poisson5pt_3162", None)] + ah.distributions(args.quick)
    for mi, (name, lens) in enumerate(mats):
        if lens is None:
            A = cmi.poisson5pt(3162, 3162, "csr")
            lens = (A.row_offsets[1:] - A.row_offsets[:-1]).cpu().numpy().astype(np.int64)
        else:
            A = ah.make_csr(cmi, torch, lens, torch.float64, seed=100 + mi)
        rows = A.num_rows
        x = cmi.fill_x(rows, torch.float64, "cuda")
        y = torch.empty(rows, dtype=torch.float64, device="cuda")
        cmi.multiply(A, x, y, cfg=scalar)
        want = y.clone()
        Aabs = cmi.CsrMatrix(rows, rows, A.num_entries, A.row_offsets, A.column_indices, A.values.abs())
        cmi.multiply(Aabs, x.abs(), y, cfg=scalar)
        bound = y.clone().clamp_(min=1e-30)
        widths = sorted({k for k in ah.candidate_widths(lens) if k < lens.max()})
        if len(widths) > 6:
            widths = widths[:: max(1, len(widths) // 6)]
        for K in widths:
            H = cmi.convert(A, "hyb", num_entries_per_row=K)
            if H.coo.num_entries == 0:
                continue
            ri = H.coo.row_indices
            per_tile = torch.bincount(ri // 256, minlength=(rows + 255) // 256)
            t = {}
            for force in ("1", "0", None):
                os.environ.pop("CMI_HYB_ONE_LAUNCH", None)
                if force is not None:
                    os.environ["CMI_HYB_ONE_LAUNCH"] = force
                H.invalidate()
                y.fill_(10.0)
                cmi.multiply(H, x, y)
                assert bool(((y - want).abs() <= 1e-6 * bound).all().item()), (name, K, force)
                if force is None:
                    picks = "one" if H.plan().hyb_launches() == 1 else "two"
                else:
                    t[force] = time_us(lambda: cmi.multiply(H, x, y))
            os.environ.pop("CMI_HYB_ONE_LAUNCH", None)
            print(f"{name:28s} {K:4d} {H.coo.num_entries / rows:8.2f} {int(per_tile.max()):9d} | {t['1']:9.1f} us {t['0']:7.1f} us  {t['1'] / t['0']:5.2f} | {picks}", flush=True)
            del H


if __name__ == "__main__":
    main()
