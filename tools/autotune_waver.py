#!/usr/bin/env python3
"""Offline autotune of csr_waver's rule (CMI_CSR_STREAM_WAVER, the run-compressed column copy of round 4) -- what the reference's KTT
tuner does per kernel (parameter space cuda/ktt/csr_multiply.h:239-247, one validated launch per configuration,
cusp/system/cuda/ktt/kernel.h:37-62), done once per architecture and persisted in the table file as "waver_rule".

Per value type, on the matrices of BASELINE.json configs[3] that the kernel serves (ldoor, nlpkkt120: real files when
CMI_SUITESPARSE_DIR holds them, else the seeded stand-ins of tools/suitesparse_like.py):

  * SHAPE  items_per_thread {1, 2, 4} x cap {0 = 3-where-cheap, 3, 4} x xcd_swizzle {0, 8, 16, 32}: every combination is a plan of its
    own, validated against csr_scalar's bits BEFORE it is timed (a mismatch is logged and the shape is out); the score of a shape is the
    geometric mean over the matrices of its time over that matrix's best time; the winner must beat the incumbent rule by more than the
    timing's spread (--margin, 2 %, on the median of three interleaved re-timings of the finalists) or the incumbent stays.
  * GATE   min_entries: the same matrices at scales 0.05 .. 0.3: csr_waver at the winning shape against AUTO's other candidates for the
    class (the wave tiles on the arrays, V = 2 / 4, and the table's csr_stream); min_entries is the smallest measured entry count from
    which the copy wins by 2 % on every matrix at and above it (rounded down to two digits), the incumbent when it wins nowhere.

    python3 tools/autotune_waver.py [--dtypes f64,f32] [--log gpurun_out/autotune_waver.jsonl] [--patch cusp-autotuned_amd/tuned/gfx950.json]
"""
import argparse
import itertools
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cusp_autotuned_amd as cmi  # noqa: E402
import suitesparse_like as ssl  # noqa: E402

SHAPES = list(itertools.product((1, 2, 4), (0, 3, 4), (0, 8, 16, 32)))
SCALES = (0.05, 0.075, 0.1, 0.15, 0.2, 0.3)
WIN = 0.98  # the copy "wins" a size when it takes at most this share of the best other candidate's time (2 %: the timing's spread)


def on_device(name, scale, dt):
    Ap, Aj, Ax, src = ssl.load(name, scale)
    rows, nnz = len(Ap) - 1, int(Ap[-1])
    A = cmi.CsrMatrix(rows, rows, nnz, torch.from_numpy(Ap).cuda(), torch.from_numpy(Aj).cuda(), torch.from_numpy(Ax).cuda().to(dt))
    x = cmi.fill_x(rows, dt, "cuda")
    y = torch.empty(rows, dtype=dt, device="cuda")
    cmi.multiply(A, x, y, cfg=cmi.Config(kernel=cmi.CSR_SCALAR))
    return A, x, y, y.clone(), src


def time_us(go, launches, settle_s=0.05):
    """median of 5 groups of `launches` back-to-back launches, after settling the clocks BY TIME (a fixed count of
    warm-up launches under-settles the short kernels)"""
    t0 = time.time()
    while time.time() - t0 < settle_s:
        go()
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(launches):
            go()
        b.record()
        b.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / launches)
    return sorted(out)[2]


def run_plan(A, x, y, want, dt, cfg, launches):
    """(us, kernel) of the plan for cfg, validated first; None when the library refuses the shape or its bits differ"""
    rows = A.num_rows
    try:
        plan = cmi.Plan.csr(dt, rows, rows, A.row_offsets, A.column_indices, cfg=cfg)
    except Exception as e:  # noqa: BLE001
        return None, f"refused: {e}"
    y.fill_(float("nan"))
    cmi.spmv_csr_plan(plan, A.row_offsets, A.column_indices, A.values, x, y)
    if not torch.equal(y, want):
        return None, "bits differ from csr_scalar's"
    us = time_us(lambda: cmi.spmv_csr_plan(plan, A.row_offsets, A.column_indices, A.values, x, y), launches)
    return us, plan.config().kernel


def gm(v):
    return math.exp(sum(math.log(t) for t in v) / len(v))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtypes", default="f64,f32")
    ap.add_argument("--matrices", default="ldoor,nlpkkt120")
    ap.add_argument("--log", default="")
    ap.add_argument("--patch", default="")
    ap.add_argument("--launches", type=int, default=40)
    ap.add_argument("--margin", type=float, default=0.02)
    args = ap.parse_args()
    log = open(args.log, "w") if args.log else None

    def emit(rec):
        if log:
            log.write(json.dumps(rec) + "\n")
            log.flush()

    names = args.matrices.split(",")
    for tag in args.dtypes.split(","):
        dt, code = (torch.float64, cmi.F64) if tag == "f64" else (torch.float32, cmi.F32)
        incumbent = cmi.tuning_waver_rule(code)
        inc_shape = (incumbent.items_per_thread, incumbent.cap, incumbent.xcd_swizzle)
        print(f"== {tag}: incumbent {incumbent}", flush=True)
        times = {}  # shape -> {matrix: us}
        for name in names:
            A, x, y, want, src = on_device(name, 1.0, dt)
            print(f"# {name}: {src}; rows {A.num_rows} entries {A.num_entries}", flush=True)
            for v, cap, swz in SHAPES:
                # (an asked plan keeps the caller's xcd_swizzle; 0 there means launch order, so the table's dealing is passed explicitly)
                cfg = cmi.Config(kernel=cmi.CSR_STREAM_WAVER, items_per_thread=v, threads_per_row=cap, xcd_swizzle=swz if swz else -1)
                us, note = run_plan(A, x, y, want, dt, cfg, args.launches)
                emit({"dtype": tag, "matrix": name, "items_per_thread": v, "cap": cap, "xcd_swizzle": swz, "us": us, "note": note})
                if us is None:
                    print(f"  V={v} cap={cap} swz={swz}: {note}", flush=True)
                    continue
                times.setdefault((v, cap, swz), {})[name] = us
            del A, x, y, want
            torch.cuda.empty_cache()
        full = {s: t for s, t in times.items() if len(t) == len(names)}

        def scores(table):
            best_of = {n: min(t[n] for t in table.values()) for n in names}
            return {s: gm([t[n] / best_of[n] for n in names]) for s, t in table.items()}

        score = scores(full)
        ranked = sorted(score, key=score.get)
        print("  first pass (one timing per shape):")
        for s in ranked[:8]:
            print(f"  V={s[0]} cap={s[1]} swz={s[2]:2d}: score {score[s]:.4f}  " + "  ".join(f"{n} {full[s][n]:.1f} us" for n in names))
        # finalists: the first pass's best five and the incumbent, timed again three times each, interleaved (one plan per shape kept
        # alive, round-robin), the median kept -- single timings of equal plans differ by up to 2.4 % (cap 0 and cap 3 on ldoor are the same plan)
        finalists = list(dict.fromkeys(ranked[:5] + ([inc_shape] if inc_shape in score else [])))
        again = {s: {} for s in finalists}
        for name in names:
            A, x, y, want, _ = on_device(name, 1.0, dt)
            plans = {s: cmi.Plan.csr(dt, A.num_rows, A.num_rows, A.row_offsets, A.column_indices,
                                     cfg=cmi.Config(kernel=cmi.CSR_STREAM_WAVER, items_per_thread=s[0], threads_per_row=s[1], xcd_swizzle=s[2] if s[2] else -1)) for s in finalists}
            seen = {s: [] for s in finalists}
            for _ in range(3):
                for s in finalists:
                    seen[s].append(time_us(lambda: cmi.spmv_csr_plan(plans[s], A.row_offsets, A.column_indices, A.values, x, y), args.launches * 2))
            for s in finalists:
                again[s][name] = sorted(seen[s])[1]
                emit({"dtype": tag, "matrix": name, "finalist": list(s), "us": seen[s]})
            del A, x, y, want, plans
            torch.cuda.empty_cache()
        score = scores(again)
        ranked = sorted(score, key=score.get)
        print("  finalists (median of three interleaved timings):")
        for s in ranked:
            print(f"  V={s[0]} cap={s[1]} swz={s[2]:2d}: score {score[s]:.4f}  " + "  ".join(f"{n} {again[s][n]:.1f} us" for n in names) + ("   <- incumbent" if s == inc_shape else ""))
        full = again
        win = ranked[0]
        if inc_shape in score and score[inc_shape] <= score[win] * (1.0 + args.margin):
            print(f"  -> the incumbent stays (within {args.margin:.0%} of the best shape)")
            win = inc_shape
        else:
            print(f"  -> new shape V={win[0]} cap={win[1]} swz={win[2]}")

        # the size gate: the copy against what AUTO runs without it, down the scales
        gate_rows = []
        for name in names:
            for sc in SCALES:
                A, x, y, want, _ = on_device(name, sc, dt)
                cfg = cmi.Config(kernel=cmi.CSR_STREAM_WAVER, items_per_thread=win[0], threads_per_row=win[1], xcd_swizzle=win[2] if win[2] else -1)
                us_w, note = run_plan(A, x, y, want, dt, cfg, args.launches * 2)
                # without the copy: the best of the wave tiles on the arrays (V = 2, 4) and the table's csr_stream -- AUTO's other candidates
                others = []
                for v in (2, 4):
                    us_o, _ = run_plan(A, x, y, want, dt, cmi.Config(kernel=cmi.CSR_STREAM_WAVEV, items_per_thread=v), args.launches * 2)
                    if us_o is not None:
                        others.append(us_o)
                others.append(time_us(lambda: cmi.spmv_csr(A.num_rows, A.num_rows, A.row_offsets, A.column_indices, A.values, x, y), args.launches * 2))
                rec = {"dtype": tag, "matrix": name, "scale": sc, "entries": A.num_entries, "waver_us": us_w, "other_us": min(others), "note": note}
                emit(rec)
                gate_rows.append(rec)
                print(f"  gate {name} x{sc}: {A.num_entries} entries, copy {us_w if us_w is None else round(us_w, 2)} us, best other {min(others):.2f} us", flush=True)
                del A, x, y, want
                torch.cuda.empty_cache()
        sized = sorted(gate_rows, key=lambda r: r["entries"])
        min_entries = incumbent.min_entries
        losing = [r["entries"] for r in sized if r["waver_us"] is None or r["waver_us"] > WIN * r["other_us"]]
        winning = [r["entries"] for r in sized if r["waver_us"] is not None and r["waver_us"] <= WIN * r["other_us"]]
        if winning:
            above = [e for e in winning if not losing or e > max(losing)]
            if above:
                # round down to 2 significant digits: the gate sits at the smallest measured size from which the copy wins on every larger one
                e = min(above)
                mag = 10 ** (len(str(e)) - 2)
                min_entries = (e // mag) * mag
        print(f"  -> min_entries {min_entries} (incumbent {incumbent.min_entries}; measured sizes where the copy does not win by 2 %: {losing or 'none'})")
        emit({"dtype": tag, "rule": {"items_per_thread": win[0], "cap": win[1], "xcd_swizzle": win[2], "min_piece": incumbent.min_piece, "min_entries": min_entries}})
        cmi.tuning_set_waver_rule(code, win[0], win[1], win[2], incumbent.min_piece, min_entries)
        print(f"== {tag}: rule {cmi.tuning_waver_rule(code)}", flush=True)
    if args.patch:  # the table file keeps its own key order and source notes: only "waver_rule" (+ its source line) changes
        doc = json.load(open(args.patch))
        doc["waver_rule"] = {tag: cmi.tuning_waver_rule(cmi.F64 if tag == "f64" else cmi.F32).as_dict() for tag in ("f64", "f32")}
        doc["waver_rule_source"] = ("tools/autotune_waver.py on MI355X: items_per_thread x cap x xcd_swizzle swept on the configs[3] FEM / KKT matrices, "
                                    "every shape validated against csr_scalar's bits before it is timed; the size gate from the same matrices down the scales")
        entries = doc.pop("entries")
        with open(args.patch, "w") as f:
            f.write("{\n")
            for k, v in doc.items():
                f.write(f"  {json.dumps(k)}: {json.dumps(v)},\n")
            f.write('  "entries": [\n')
            f.write(",\n".join("    " + json.dumps(e) for e in entries))
            f.write("\n  ]\n}\n")
        print(f"wrote {args.patch}")


if __name__ == "__main__":
    main()
