#!/usr/bin/env python3
"""Offline autotuner: replaces the reference's run-time KTT tuning layer
(cusp/ktt/ktt.h:35-101 `tune`, cusp/system/cuda/ktt/multiply.h:107-154; parameter spaces in
cusp/system/cuda/ktt/{csr,ell,dia,coo}_multiply.h) with a sweep that runs ONCE on an MI355X and
persists its choices to cusp-autotuned_amd/tuned/gfx950.json -- the table the library consults
when cmi_spmv_* is called with a NULL config.

For every (format, dtype, matrix) it enumerates the kernel-variant x launch-shape space below,
VALIDATES each configuration before timing it against the result of the simplest kernel of the
library (csr_scalar, one lane per row, itself pinned bit-for-bit to the reference host loop by the
test-suite) -- exactly what testing/ktt.cu:142-202 does: reference y from the stock multiply with
the tuner disabled (:176-178), then every configuration compared with it; a configuration that
fails validation is reported and never selected, times it with HIP events in interleaved rounds (guide rule 24), and records the fastest.

    python tools/autotune.py [--quick] [--formats csr,ell,dia,coo] [--dtypes f64,f32]
                             [--out cusp-autotuned_amd/tuned/gfx950.json] [--log gpurun_out/autotune.jsonl]

Matrices: poisson5pt 3162x3162 (the headline workload; mean 5 entries/row) plus seeded synthetic CSR
matrices with mean row lengths 2..96 so every bucket of the table gets a measured entry
(the reference's performance/csr_vector/csr_vector.cu:41-62 sweeps D = 1..64 the same way).
"""
import argparse
import ctypes
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def csr_space(cmi, mean, quick, stream_only=False):
    out = []
    blocks = (256,) if quick else (128, 256, 512)
    for b in blocks:
        out.append(cmi.Config(kernel=cmi.CSR_SCALAR, block_size=b))
    tprs = [t for t in (2, 4, 8, 16, 32, 64) if t <= max(2, 4 * mean) and 4 * t >= mean / 4]
    for t, b, nt in itertools.product(tprs, blocks, (0, 1)):
        out.append(cmi.Config(kernel=cmi.CSR_VECTOR, block_size=b, threads_per_row=t, nontemporal=nt))
    if stream_only:  # --csr-stream-only: a re-tune of the row-tile kernel alone (the other variants lost by 10 %+ in the full runs)
        out = []
    if mean <= 100:  # one lane per row (storage order, bit-exact): its LDS reads are batched, so it holds up to ~80/row
        # policy bits: 1 = nt loads of the streams, 2 = nt stores of y, 4 = entry streams requested lane-strided instead of as
        # 16-byte vectors (short rows: round 2)
        pols = (0, 2) if quick else (0, 2, 6, 7) if mean > 12 else (0, 1, 2, 3, 6, 7)
        for b, ipt, nt, swz in itertools.product(blocks, (1, 2, 4), pols, (0, 8) if quick else (0, 8, 16, 32, 64, 128)):
            if stream_only and ipt == 4 and mean <= 12:
                continue
            tile = b * ipt * 4
            base = max(1, int((tile - 3) / max(mean, 0.25)))
            aligned = max(1, base // 16 * 16)
            rpbs = {min(aligned, 4 * b)}
            if not quick:
                rpbs |= {min(base, 4 * b), min(max(1, aligned - 16), 4 * b), min(b, base)}
            for rpb in sorted(rpbs):
                out.append(cmi.Config(kernel=cmi.CSR_STREAM, block_size=b, items_per_thread=ipt, rows_per_block=rpb,
                                      nontemporal=nt, xcd_swizzle=swz))
    if stream_only:
        return out
    if mean >= 6:
        # longer rows: the same LDS-staged tile, but a power-of-two group of lanes sums each row
        for b, ipt, tpr, nt, swz in itertools.product(blocks, (1, 2, 4), (2, 4, 8, 16, 32, 64), (0, 2), (0, 8)):
            if tpr > 8 * mean or tpr * 16 < mean:
                continue
            tile = b * ipt * 4
            base = max(1, int((tile - 3) / mean))
            base = min(base, 4 * (b // tpr))
            for rpb in sorted({base, max(1, base // 16 * 16)}):
                out.append(cmi.Config(kernel=cmi.CSR_STREAM, block_size=b, items_per_thread=ipt, rows_per_block=rpb,
                                      threads_per_row=tpr, nontemporal=nt, xcd_swizzle=swz))
    if mean <= 40:
        for b, nt, chunked, bpc in itertools.product(blocks, (0, 2) if quick else (0, 1, 2, 3), (0, 1),
                                                     (4, 8) if quick else (2, 3, 4, 6, 8, 12)):
            base = max(1, min(int((b * 4 - 3) / max(mean, 0.25)), b - 1))
            aligned = max(1, base // 16 * 16)
            for rpb in sorted({aligned} if quick else {base, aligned}):
                out.append(cmi.Config(kernel=cmi.CSR_STREAM_PIPE, block_size=b, rows_per_block=rpb, nontemporal=nt,
                                      xcd_swizzle=chunked, blocks_per_cu=bpc))
    return out


def ell_space(cmi, quick, width=0):
    # xcd_swizzle: tiles (one workgroup's rows) dealt to the XCDs in chunks, so that a chunk's x window lands in one L2
    out = [cmi.Config(kernel=cmi.ELL_ROW, block_size=b, items_per_thread=r, threads_per_row=1 if width else 0, nontemporal=nt, xcd_swizzle=swz)
           for b, r, nt, swz in itertools.product((256,) if quick else (128, 256, 512, 1024), (1, 2), (0, 1, 2, 3),
                                                  (0, 32) if quick else (0, 8, 16, 32, 64, 128))]
    # per-bucket tuning (round 3): the lanes-per-row axis of the reference's THREADS_PER_ROW (ktt/kernels/ell_kernel.h:102-109) for
    # wide rows -- 2..16 lanes per row (re-associated sums, <= 1e-6), each lane keeping at least 4 slots
    for tpr in (2, 4, 8, 16):
        if width and width >= 4 * tpr:
            out += [cmi.Config(kernel=cmi.ELL_ROW, block_size=b, items_per_thread=1, threads_per_row=tpr, nontemporal=nt, xcd_swizzle=swz)
                    for b, nt, swz in itertools.product((256,) if quick else (256, 512), (1, 3), (0, 32))]
    return out


def dia_space(cmi, quick):
    return [cmi.Config(kernel=cmi.DIA_ROW, block_size=b, items_per_thread=r, nontemporal=nt, xcd_swizzle=swz)
            for b, r, nt, swz in itertools.product((256,) if quick else (128, 256, 512, 1024), (1, 2), (0, 1, 2, 3),
                                                   (0, 32) if quick else (0, 8, 16, 32, 64, 128))]


def coo_space(cmi, quick):
    out = [cmi.Config(kernel=k, block_size=b, items_per_thread=i, nontemporal=nt)
           for k, b, i, nt in itertools.product((cmi.COO_SEGMENTED, cmi.COO_LANE4), (256,) if quick else (128, 256, 512),
                                                (1, 2, 4, 8, 16, 32), (0, 1))]
    # the tile kernel (row-sorted entries -- the tuning matrices are: what a plan selects; its launch shape is fixed, the
    # cache policy and the XCD dealing of its 1024-entry tiles are tuned)
    out += [cmi.Config(kernel=cmi.COO_TILE, block_size=256, nontemporal=nt, xcd_swizzle=swz)
            for nt, swz in itertools.product((0, 1, 2, 3), (0, 32) if quick else (0, 8, 16, 32, 64, 128))]
    return out


class Timer:
    def __init__(self, cmi, torch):
        self.cmi, self.lib, self.torch = cmi, cmi.lib(), torch
        self.e0, self.e1 = ctypes.c_void_p(), ctypes.c_void_p()
        cmi.check(self.lib.cmi_event_create(ctypes.byref(self.e0)))
        cmi.check(self.lib.cmi_event_create(ctypes.byref(self.e1)))

    def time(self, fn, iters):
        s = ctypes.c_void_p(self.torch.cuda.current_stream().cuda_stream)
        self.cmi.check(self.lib.cmi_event_record(self.e0, s))
        for _ in range(iters):
            fn()
        self.cmi.check(self.lib.cmi_event_record(self.e1, s))
        ms = ctypes.c_float()
        self.cmi.check(self.lib.cmi_event_elapsed_ms(self.e0, self.e1, ctypes.byref(ms)))
        return ms.value / iters


def synthetic_csr(rows, cols, mean, seed, dtype):
    rng = np.random.default_rng(seed)
    lens = rng.poisson(mean, size=rows).astype(np.int64)
    lens = np.minimum(lens, cols)
    Ap = np.zeros(rows + 1, np.int32)
    Ap[1:] = np.cumsum(lens)
    nnz = int(Ap[-1])
    # columns clustered around the diagonal (FEM-like locality), sorted within the row
    centre = np.repeat(np.arange(rows, dtype=np.int64) * cols // rows, lens)
    Aj = np.clip(centre + rng.integers(-2000, 2001, size=nnz), 0, cols - 1).astype(np.int32)
    row_id = np.repeat(np.arange(rows, dtype=np.int64), lens)
    order = np.lexsort((Aj, row_id))
    Aj = Aj[order]
    Ax = rng.standard_normal(nnz).astype(dtype)
    return Ap, Aj, Ax


def stencil_csr(nx, ny, nz, points, dtype):
    """CSR of a constant-coefficient stencil on an nx x ny x nz grid, rows in grid order, entries in
    stencil order (the layout cusp::gallery::generate_matrix_from_stencil + DIA->CSR produces)."""
    N = nx * ny * nz
    r = np.arange(N, dtype=np.int64)
    ix, iy, iz = r % nx, (r // nx) % ny, r // (nx * ny)
    cols = np.empty((N, len(points)), np.int64)
    vals = np.empty((N, len(points)), dtype)
    mask = np.empty((N, len(points)), bool)
    for k, (dx, dy, dz, v) in enumerate(points):
        jx, jy, jz = ix + dx, iy + dy, iz + dz
        mask[:, k] = (jx >= 0) & (jx < nx) & (jy >= 0) & (jy < ny) & (jz >= 0) & (jz < nz)
        cols[:, k] = r + dx + dy * nx + dz * nx * ny
        vals[:, k] = v
    Ap = np.zeros(N + 1, np.int32)
    Ap[1:] = np.cumsum(mask.sum(axis=1))
    return Ap, cols[mask].astype(np.int32), vals[mask]


def block_expand(Ap, Aj, Ax, dof, dtype):
    """d degrees of freedom per grid point: every scalar entry becomes a dense d x d block (the row
    structure of 3-D elasticity / ldoor-like FEM matrices: 27-point x 3 dof = 81 entries per row)."""
    N = len(Ap) - 1
    lens = np.diff(Ap).astype(np.int64)
    Ap2 = np.zeros(N * dof + 1, np.int64)
    Ap2[1:] = np.cumsum(np.repeat(lens * dof, dof))
    # row r*dof + a holds, for every scalar entry (r, c), the columns c*dof + 0..dof-1
    cols_block = (Aj.astype(np.int64)[:, None] * dof + np.arange(dof)[None, :]).reshape(-1)  # per scalar entry: dof cols
    # rows of one grid point are identical in structure: tile each scalar row's expanded columns dof times
    starts = Ap[:-1].astype(np.int64) * dof
    out_cols = np.empty(int(Ap2[-1]), np.int32)
    out_vals = np.empty(int(Ap2[-1]), dtype)
    rng = np.random.default_rng(99)
    for a in range(dof):
        idx_rows = np.arange(N) * dof + a
        # destination ranges of rows idx_rows are contiguous runs of length lens*dof
        dst = np.repeat(Ap2[idx_rows], lens * dof) + (np.arange(int((lens * dof).sum())) - np.repeat(np.cumsum(lens * dof) - lens * dof, lens * dof))
        out_cols[dst] = cols_block
        out_vals[dst] = rng.standard_normal(len(cols_block)).astype(dtype)
    del starts
    return Ap2.astype(np.int32), out_cols, out_vals


def stencil_points(kind):
    if kind == 27:
        return [(i, j, k, 26.0 if (i, j, k) == (0, 0, 0) else -1.0) for k in (-1, 0, 1) for j in (-1, 0, 1) for i in (-1, 0, 1)]
    if kind == 9:
        return [(i, j, 0, 8.0 if (i, j) == (0, 0) else -1.0) for j in (-1, 0, 1) for i in (-1, 0, 1)]
    if kind == 7:
        return [(0, 0, -1, -1.0), (0, -1, 0, -1.0), (-1, 0, 0, -1.0), (0, 0, 0, 6.0), (1, 0, 0, -1.0), (0, 1, 0, -1.0), (0, 0, 1, -1.0)]
    raise ValueError(kind)


def tune_one(cmi, torch, timer, label, space, run, check, iters, rounds, log, alg_bytes):
    """validate, then interleaved timing rounds; returns (best_cfg, best_ms, results)."""
    valid = []
    for cfg in space:
        try:
            ok, note = check(cfg)
        except cmi.CmiError as e:  # e.g. a tile that does not fit LDS: not a candidate
            ok, note = False, f"rejected: {e}"
        if ok:
            valid.append(cfg)
        else:
            log({"label": label, "config": cfg.as_dict(), "status": "ValidationFailed", "note": note})
    times = {id(c): [] for c in valid}
    for _ in range(rounds):
        for cfg in valid:
            times[id(cfg)].append(timer.time(lambda: run(cfg), iters))
    results = []
    for cfg in valid:
        t = times[id(cfg)]
        rec = {"label": label, "config": cfg.as_dict(), "status": "Ok", "ms_min": min(t), "ms_median": float(np.median(t)),
               "gbps_algorithmic": alg_bytes / (float(np.median(t)) * 1e-3) / 1e9}
        log(rec)
        results.append((float(np.median(t)), cfg, rec))
    results.sort(key=lambda r: r[0])
    return (results[0][1], results[0][0], results) if results else (None, None, [])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--formats", default="csr,ell,dia,coo")
    ap.add_argument("--dtypes", default="f64,f32")
    ap.add_argument("--out", default=os.path.join(ROOT, "cusp-autotuned_amd", "tuned", "gfx950.json"))
    ap.add_argument("--log", default=os.path.join(ROOT, "gpurun_out", "autotune.jsonl"))
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--grid", type=int, default=3162)
    ap.add_argument("--skip-synthetic", action="store_true")
    ap.add_argument("--csr-stream-only", action="store_true", help="CSR: only the row-tile kernel with one lane per row (re-tune of its shapes)")
    ap.add_argument("--synthetic-rows", type=int, default=1_000_000, help="rows of the seeded synthetic CSR matrices (8e6: their streams no longer fit the 256 MiB Infinity Cache)")
    ap.add_argument("--csr-max-mean", type=float, default=1e9, help="CSR: skip the tuning matrices with more entries per row than this")
    ap.add_argument("--merge", action="store_true", help="start from the table at --out (tune some formats, keep the others)")
    ap.add_argument("--buckets", default="", help="--per-bucket: only these buckets (comma-separated)")
    ap.add_argument("--skip-headline", action="store_true", help="do not re-tune on the headline matrix (per-bucket re-runs)")
    ap.add_argument("--per-bucket", action="store_true", help="ELL / DIA / COO: a tuning matrix per bucket of the table (width 1.2 x 2^b), not the "
                                                             "headline matrix's shape replicated over all eight")
    args = ap.parse_args()

    import torch
    import cusp_autotuned_amd as cmi

    assert torch.cuda.is_available(), "the autotuner needs an MI355X"
    scalar = cmi.Config(kernel=cmi.CSR_SCALAR)  # the validator: simplest kernel, tuner "disabled"
    timer = Timer(cmi, torch)
    os.makedirs(os.path.dirname(args.log), exist_ok=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    logf = open(args.log, "a")

    def log(rec):
        logf.write(json.dumps(rec) + "\n")
        logf.flush()

    cmi.tuning_clear()
    if args.merge and os.path.exists(args.out):
        cmi.tuning_load(args.out)  # explicit configs are timed below, so the loaded entries do not steer the search
    formats = args.formats.split(",")
    t_start = time.time()
    summary = []
    synth_cache = {}
    for tag in args.dtypes.split(","):
        tdt = torch.float64 if tag == "f64" else torch.float32
        ndt = np.float64 if tag == "f64" else np.float32
        vb = 8 if tag == "f64" else 4
        dcode = cmi.F64 if tag == "f64" else cmi.F32
        m = n = args.grid
        N = m * n
        A = cmi.poisson5pt(m, n, "csr", dtype=tdt)
        dx = cmi.fill_x(N, tdt, "cuda")
        y = torch.empty(N, dtype=tdt, device="cuda")
        cmi.multiply(A, dx, y, cfg=scalar)
        want = y.cpu().numpy()
        scale = float(np.abs(want).max())
        tol = 1e-6 if tag == "f64" else 1e-5

        def checker(mat, exact_kernels):
            def check(cfg):
                y.fill_(10.0)
                cmi.multiply(mat, dx, y, cfg=cfg)
                got = y.cpu().numpy()
                if cfg.kernel in exact_kernels and cfg.threads_per_row <= 1:
                    return bool(np.array_equal(got, want)), "bit-exact required"
                return bool(np.max(np.abs(got - want)) <= tol * scale), f"tolerance {tol}"
            return check

        if args.skip_headline:
            formats_h = []
        else:
            formats_h = formats
        if "csr" in formats_h:
            label = f"csr/{tag}/poisson{m}x{n}"
            best, ms, res = tune_one(cmi, torch, timer, label, csr_space(cmi, 5.0, args.quick, args.csr_stream_only),
                                     lambda cfg: cmi.multiply(A, dx, y, cfg=cfg),
                                     checker(A, (cmi.CSR_SCALAR, cmi.CSR_STREAM, cmi.CSR_STREAM_PIPE)), args.iters, args.rounds, log,
                                     cmi.csr_bytes(N, A.num_entries, vb))
            cmi.tuning_set(cmi.FORMAT_CSR, dcode, A.num_entries / N, best)
            summary.append((label, best.as_dict(), ms))
            print(label, best, f"{ms * 1e3:.1f} us", flush=True)
        if "ell" in formats_h:
            E = cmi.convert(A, "ell")
            label = f"ell/{tag}/poisson{m}x{n}"
            best, ms, res = tune_one(cmi, torch, timer, label, ell_space(cmi, args.quick),
                                     lambda cfg: cmi.multiply(E, dx, y, cfg=cfg), checker(E, (cmi.ELL_ROW,)),
                                     args.iters, args.rounds, log, cmi.ell_bytes(N, 5, E.pitch, vb))
            for b in range(0, 8):  # ELL/DIA launch shape does not depend on the width: fill every bucket
                cmi.tuning_set(cmi.FORMAT_ELL, dcode, 2.0 ** b * 1.2, best)
            summary.append((label, best.as_dict(), ms))
            print(label, best, f"{ms * 1e3:.1f} us", flush=True)
            del E
        if "dia" in formats_h:
            D = cmi.poisson5pt(m, n, "dia", dtype=tdt)
            label = f"dia/{tag}/poisson{m}x{n}"
            best, ms, res = tune_one(cmi, torch, timer, label, dia_space(cmi, args.quick),
                                     lambda cfg: cmi.multiply(D, dx, y, cfg=cfg), checker(D, (cmi.DIA_ROW,)),
                                     args.iters, args.rounds, log, cmi.dia_bytes(N, 5, D.pitch, vb))
            for b in range(0, 8):
                cmi.tuning_set(cmi.FORMAT_DIA, dcode, 2.0 ** b * 1.2, best)
            summary.append((label, best.as_dict(), ms))
            print(label, best, f"{ms * 1e3:.1f} us", flush=True)
            del D
        if "coo" in formats_h:
            C = cmi.convert(A, "coo")
            label = f"coo/{tag}/poisson{m}x{n}"
            space = coo_space(cmi, args.quick)
            # two table keys: "coo" = what a plan-less call runs on entries in ANY order (never the tile kernel);
            # "coo_sorted" = what a plan runs once it has found the entries sorted by row (best of everything)
            agnostic = [c for c in space if c.kernel != cmi.COO_TILE]
            best, ms, res = tune_one(cmi, torch, timer, label, agnostic,
                                     lambda cfg: cmi.multiply(C, dx, y, cfg=cfg), checker(C, (cmi.COO_TILE,)),
                                     args.iters, args.rounds, log, cmi.coo_bytes(N, A.num_entries, vb))
            for b in range(0, 8):
                cmi.tuning_set(cmi.FORMAT_COO, dcode, 2.0 ** b * 1.2, best)
            summary.append((label, best.as_dict(), ms))
            print(label, best, f"{ms * 1e3:.1f} us", flush=True)
            label = f"coo_sorted/{tag}/poisson{m}x{n}"
            best2, ms2, res = tune_one(cmi, torch, timer, label, [c for c in space if c.kernel == cmi.COO_TILE] + [best],
                                       lambda cfg: cmi.multiply(C, dx, y, cfg=cfg), checker(C, (cmi.COO_TILE,)),
                                       args.iters, args.rounds, log, cmi.coo_bytes(N, A.num_entries, vb))
            # (round 4: measured and logged, no longer written to the table -- sorted COO multiplies through its plan's row offsets + the
            #  CSR kernels; CMI_COO_TILE stays an explicit-config kernel with its built-in shape)
            summary.append((label, best2.as_dict(), ms2))
            print(label, best2, f"{ms2 * 1e3:.1f} us", flush=True)
            del C
        del A

        # ---- per-bucket tuning of ELL / DIA / COO (the table held ONE shape per format replicated over its eight
        #      buckets).  Bucket b (mean 1.2 x 2^b entries per row) gets a banded matrix of that width -- near diagonals plus far
        #      ones a thousand rows apart, like a stencil's -- sized for ~600 MB of ELL slots (beyond the Infinity Cache) where the
        #      row count allows; bucket 2 keeps the headline matrix.  Parameter spaces being replaced: cusp/system/cuda/ktt/
        #      {ell_multiply.h:20-77, dia_multiply.h:24-55, coo_multiply.h:22-54}; validation as testing/ktt.cu:142-202.
        if args.per_bucket and any(f in formats for f in ("ell", "dia", "coo")):
            import math
            only = {int(t) for t in args.buckets.split(",") if t != ""} if args.buckets else None
            for b in range(0, 8):
                if b == 2 or (only is not None and b not in only):
                    continue  # (2: the headline matrix above)
                width = max(1, int(round(1.2 * 2.0 ** b)))
                while b >= 1 and math.floor(math.log2(width * (1.0 - 1e-3))) < b:
                    width += 1  # boundary rows make the mean a little smaller than the width: stay inside bucket b (width 2 -> mean 1.9999999 is bucket 0)
                rows_b = int(min(1.0e7, max(2.0e5, 5.0e7 / width))) if not args.quick else 100_000
                offs = sorted({0} | {(-1) ** k * ((k + 1) // 2) * (1 if k < 4 else 1000) for k in range(1, width)})
                while len(offs) < width:
                    offs = sorted(set(offs) | {max(offs) + 1000})
                r = np.arange(rows_b, dtype=np.int64)
                cols_b = r[:, None] + np.array(offs, np.int64)[None, :]
                mask = (cols_b >= 0) & (cols_b < rows_b)
                Ap_b = np.zeros(rows_b + 1, np.int32)
                Ap_b[1:] = np.cumsum(mask.sum(axis=1))
                Aj_b = cols_b[mask].astype(np.int32)
                Ax_b = np.random.default_rng(100 + b).standard_normal(len(Aj_b)).astype(ndt)
                S = cmi.CsrMatrix(rows_b, rows_b, len(Aj_b), torch.from_numpy(Ap_b).cuda(), torch.from_numpy(Aj_b).cuda(), torch.from_numpy(Ax_b).cuda())
                dxs = cmi.fill_x(rows_b, tdt, "cuda")
                ys = torch.empty(rows_b, dtype=tdt, device="cuda")
                cmi.multiply(S, dxs, ys, cfg=scalar)
                wants = ys.cpu().numpy()
                Sabs = cmi.CsrMatrix(rows_b, rows_b, len(Aj_b), S.row_offsets, S.column_indices, S.values.abs())
                cmi.multiply(Sabs, dxs.abs(), ys, cfg=scalar)
                bound_b = np.maximum(ys.cpu().numpy(), 1e-30)
                del Sabs

                def checker_b(mat, exact_kernels):
                    def check(cfg):
                        ys.fill_(10.0)
                        cmi.multiply(mat, dxs, ys, cfg=cfg)
                        got = ys.cpu().numpy()
                        if cfg.kernel in exact_kernels and cfg.threads_per_row <= 1:
                            return bool(np.array_equal(got, wants)), "bit-exact required"
                        return bool(np.all(np.abs(got - wants) <= tol * bound_b)), f"tolerance {tol}"
                    return check
                mean_b = len(Aj_b) / rows_b
                name_b = f"banded{width}_rows{rows_b}"
                print(f"bucket {b}: {name_b}: {len(Aj_b)} entries ({mean_b:.2f}/row)", flush=True)
                if "ell" in formats:
                    E = cmi.convert(S, "ell")
                    label = f"ell/{tag}/{name_b}"
                    best, ms, res = tune_one(cmi, torch, timer, label, ell_space(cmi, args.quick, width), lambda cfg: cmi.multiply(E, dxs, ys, cfg=cfg),
                                             checker_b(E, (cmi.ELL_ROW,)), args.iters, args.rounds, log, cmi.ell_bytes(rows_b, E.num_entries_per_row, E.pitch, vb))
                    if best is not None:
                        if best.threads_per_row == 1:
                            best.threads_per_row = 0  # one lane per row won on this (many-row) matrix: the few-row rule stays in force
                        cmi.tuning_set(cmi.FORMAT_ELL, dcode, mean_b, best)
                        summary.append((label, best.as_dict(), ms))
                        print(label, best, f"{ms * 1e3:.1f} us", flush=True)
                    del E
                if "dia" in formats:
                    D = cmi.convert(S, "dia")
                    label = f"dia/{tag}/{name_b}"
                    best, ms, res = tune_one(cmi, torch, timer, label, dia_space(cmi, args.quick), lambda cfg: cmi.multiply(D, dxs, ys, cfg=cfg),
                                             checker_b(D, (cmi.DIA_ROW,)), args.iters, args.rounds, log, cmi.dia_bytes(rows_b, width, D.pitch, vb))
                    if best is not None:
                        cmi.tuning_set(cmi.FORMAT_DIA, dcode, mean_b, best)
                        summary.append((label, best.as_dict(), ms))
                        print(label, best, f"{ms * 1e3:.1f} us", flush=True)
                    del D
                if "coo" in formats:
                    C = cmi.convert(S, "coo")
                    space = coo_space(cmi, args.quick)
                    label = f"coo/{tag}/{name_b}"
                    best, ms, res = tune_one(cmi, torch, timer, label, [c for c in space if c.kernel != cmi.COO_TILE], lambda cfg: cmi.multiply(C, dxs, ys, cfg=cfg),
                                             checker_b(C, (cmi.COO_TILE,)), args.iters, args.rounds, log, cmi.coo_bytes(rows_b, len(Aj_b), vb))
                    if best is not None:
                        cmi.tuning_set(cmi.FORMAT_COO, dcode, mean_b, best)
                        summary.append((label, best.as_dict(), ms))
                        print(label, best, f"{ms * 1e3:.1f} us", flush=True)
                    label = f"coo_sorted/{tag}/{name_b}"
                    best2, ms2, res = tune_one(cmi, torch, timer, label, [c for c in space if c.kernel == cmi.COO_TILE] + ([best] if best is not None else []),
                                               lambda cfg: cmi.multiply(C, dxs, ys, cfg=cfg), checker_b(C, (cmi.COO_TILE,)), args.iters, args.rounds, log,
                                               cmi.coo_bytes(rows_b, len(Aj_b), vb))
                    if best2 is not None:
                        pass  # (round 4: coo_sorted keys retired from the table, see above)
                        summary.append((label, best2.as_dict(), ms2))
                        print(label, best2, f"{ms2 * 1e3:.1f} us", flush=True)
                    del C
                del S, dxs, ys

        # synthetic CSR matrices for the other mean-row-length buckets
        if "csr" in formats and not args.skip_synthetic:
            rows = cols = args.synthetic_rows if not args.quick else 200_000
            # FEM-like stencils where a realistic shape exists for the bucket (9-point 2-D: 9/row,
            # 27-point 3-D: ~27/row, an nlpkkt120-like shape), seeded synthetic matrices elsewhere
            cases = [("synthetic", 1.5), ("synthetic", 3.0), ("stencil9", 2000 if not args.quick else 600),
                     ("stencil27", 150 if not args.quick else 60), ("block27x2", 90 if not args.quick else 40),
                     ("block27x3", 70 if not args.quick else 30), ("block27x8", 36 if not args.quick else 20)]
            for kind, param in cases:
                nominal = param if kind == "synthetic" else {"stencil9": 9, "stencil27": 27, "block27x2": 54, "block27x3": 81, "block27x8": 216}[kind]
                if nominal > args.csr_max_mean:
                    continue
                if kind == "synthetic":
                    mean = param
                    r = rows if mean < 100 else rows // 4
                    key = (r, mean)
                    if key not in synth_cache:  # same structure for f64 and f32
                        synth_cache[key] = synthetic_csr(r, r, mean, int(mean * 10), np.float64)
                    name = f"synthetic_mean{mean}"
                else:
                    g = param
                    key = (kind, g)
                    if key not in synth_cache:
                        if kind == "stencil9":
                            synth_cache[key] = stencil_csr(g, g, 1, stencil_points(9), np.float64)
                        else:
                            base = stencil_csr(g, g, g, stencil_points(27), np.float64)
                            dof = {"stencil27": 1, "block27x2": 2, "block27x3": 3, "block27x8": 8}[kind]
                            synth_cache[key] = base if dof == 1 else block_expand(*base, dof, np.float64)
                    name = f"{kind}_{g}"
                Ap, Aj, Ax = synth_cache[key]
                r = len(Ap) - 1
                mean = len(Ax) / r
                Ax = Ax.astype(ndt)
                print(f"{name}: {r} rows, {len(Ax)} entries ({mean:.2f}/row)", flush=True)
                S = cmi.CsrMatrix(r, r, len(Ax), torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda(),
                                  torch.from_numpy(Ax).cuda())
                dxs = cmi.fill_x(r, tdt, "cuda")
                ys = torch.empty(r, dtype=tdt, device="cuda")
                cmi.multiply(S, dxs, ys, cfg=scalar)
                wants = ys.cpu().numpy()
                Sabs = cmi.CsrMatrix(r, r, len(Ax), S.row_offsets, S.column_indices, S.values.abs())
                cmi.multiply(Sabs, dxs.abs(), ys, cfg=scalar)
                bound = ys.cpu().numpy()
                del Sabs

                def check(cfg):
                    ys.fill_(10.0)
                    cmi.multiply(S, dxs, ys, cfg=cfg)
                    got = ys.cpu().numpy()
                    if cfg.kernel in (cmi.CSR_SCALAR, cmi.CSR_STREAM, cmi.CSR_STREAM_PIPE) and cfg.threads_per_row <= 1:
                        return bool(np.array_equal(got, wants)), "bit-exact required"
                    return bool(np.all(np.abs(got - wants) <= tol * np.maximum(bound, 1e-30))), f"tolerance {tol}"

                label = f"csr/{tag}/{name}"
                best, ms, res = tune_one(cmi, torch, timer, label, csr_space(cmi, mean, args.quick, args.csr_stream_only),
                                         lambda cfg: cmi.multiply(S, dxs, ys, cfg=cfg), check, args.iters, args.rounds,
                                         log, cmi.csr_bytes(r, len(Ax), vb))
                if best is not None:
                    cmi.tuning_set(cmi.FORMAT_CSR, dcode, len(Ax) / r, best)
                    summary.append((label, best.as_dict(), ms))
                    print(label, best, f"{ms * 1e3:.1f} us", flush=True)
                del S, dxs, ys

    keep = {}
    if os.path.exists(args.out):  # keys the library's writer does not know about (provenance notes) survive a re-tune
        try:
            keep = {k: v for k, v in json.load(open(args.out)).items() if k not in ("arch", "version", "entries", "hyb_rule")}
        except Exception:
            keep = {}
    cmi.tuning_save(args.out)
    if keep:
        doc = json.load(open(args.out))
        doc.update(keep)
        with open(args.out, "w") as f:
            f.write("{\n")
            for k, v in doc.items():
                if k != "entries":
                    f.write(f"  {json.dumps(k)}: {json.dumps(v)},\n")
            f.write('  "entries": [\n' + ",\n".join("    " + json.dumps(e) for e in doc["entries"]) + "\n  ]\n}\n")
    print(f"wrote {args.out} in {time.time() - t_start:.0f} s")
    log({"summary": [{"label": l, "config": c, "ms": t} for l, c, t in summary]})


if __name__ == "__main__":
    main()
