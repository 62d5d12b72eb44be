#!/usr/bin/env python3
"""Offline autotune of the HYB split: the ELL width cutoff (BASELINE.json north_star: "format/launch-shape selection
(threads-per-row, ELL width cutoff) redone as an offline autotune").

The reference picks the width with cusp::compute_optimal_entries_per_row(relative_speed = 3.0, breakeven_threshold = 4096)
-- constants "chosen empirically for a GTX280" (cusp/system/detail/generic/format_utils.inl:270-325,
cusp/detail/functional.inl:114-132, call site generic/conversions/csr_to_other.h:248-254).  Here the same functional form
is kept (so (3.0, 4096) still reproduces the reference's widths) and the pair is MEASURED:

  1. for a set of row-length distributions (seeded, built on the device) every candidate width K is converted
     (cmi_csr_to_ell + cmi_csr_to_hyb_coo), VALIDATED against the library's csr_scalar result (pinned bit for bit to the
     reference host loop by tests/) and timed (ELL launch + COO launch through its plan, HIP events, interleaved rounds);
  2. two rule kinds are fitted on a grid of (relative_speed, threshold) pairs, each pair scored by the time of the width IT
     would choose for each matrix relative to that matrix's best width (geometric mean of the regrets): the reference's
     functional form, and the launch-cost model CMI_HYB_RULE_COST (include/cusp_mi355x.h) that adds the fixed cost of the
     second launch -- what decides small matrices on this machine;
  3. the better kind with its best pair is written into the tuning table (`hyb_rule`), the raw sweep to --log.

    python tools/autotune_hyb.py [--quick] [--dtypes f64,f32] [--out cusp-autotuned_amd/tuned/gfx950.json] [--log gpurun_out/autotune_hyb.jsonl]
"""
import argparse
import ctypes
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rule_width(hist, num_rows, relative_speed, breakeven):
    """cmi_hyb_entries_per_row's rule on a host histogram (hist[k] = rows of length k), float arithmetic as the reference."""
    max_len = len(hist) - 1
    cum = 0
    for k in range(max_len):
        cum += int(hist[k])
        longer = num_rows - cum
        if np.float32(relative_speed) * np.float32(longer) < np.float32(num_rows) or longer < breakeven:
            return k
    return max_len


def cost_rule_width(hist, num_rows, relative_speed, threshold):
    """CMI_HYB_RULE_COST on a host histogram: argmin_k num_rows*k + [coo(k) > 0] * (threshold + relative_speed * coo(k));
    ties go to the wider ELL part (the library walks k downwards with a strict comparison)."""
    max_len = len(hist) - 1
    best, K = float(num_rows) * max_len, max_len
    longer, coo = 0, 0.0
    for k in range(max_len - 1, -1, -1):
        longer += int(hist[k + 1])
        coo += float(longer)
        cost = float(num_rows) * k + threshold + relative_speed * coo
        if cost < best:
            best, K = cost, k
    return K


def make_csr(cmi, torch, lens, tdt, seed):
    """CSR on the device from row lengths: banded, strictly increasing columns per row (stride 3 around the diagonal)."""
    dev = "cuda"
    lens = torch.as_tensor(lens, dtype=torch.int64, device=dev)
    rows = lens.numel()
    Ap = torch.zeros(rows + 1, dtype=torch.int64, device=dev)
    Ap[1:] = torch.cumsum(lens, 0)
    nnz = int(Ap[-1].item())
    row = torch.repeat_interleave(torch.arange(rows, device=dev), lens)
    j = torch.arange(nnz, device=dev) - Ap[:-1][row]
    col = row + (j - lens[row] // 2) * 3
    col = col.clamp_(0, rows - 1)
    # keep columns strictly increasing inside a row after the clamp (boundary rows): shift by the within-row index there
    col = torch.where((col == 0) | (col == rows - 1), (row + j).clamp_(0, rows - 1), col)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    Ax = torch.randn(nnz, dtype=tdt, device=dev, generator=g)
    return cmi.CsrMatrix(rows, rows, nnz, Ap.to(torch.int32), col.to(torch.int32), Ax)


def distributions(quick):
    """(name, row lengths) -- seeded.  Shapes after BASELINE.json configs[2] (the headline matrix) and configs[3] (the
    SuiteSparse set's published row-length statistics: thermal2 1..11 mean 7, ldoor 28..77 mean 45, nlpkkt120 5..28 mean 27)."""
    rng = np.random.default_rng(2024)
    s = 4 if quick else 1
    out = []
    n = 2_000_000 // s
    out.append(("uniform_1_16", rng.integers(1, 17, size=n)))
    b = np.full(n, 4)
    b[rng.random(n) < 0.10] = 40
    out.append(("bimodal_4_40", b))
    b = np.full(n, 6)
    b[rng.random(n) < 0.30] = 12
    out.append(("bimodal_6_12_30pct", b))
    n = 1_000_000 // s
    p = np.minimum(4 + np.floor(rng.pareto(1.3, size=n) * 3), 3000).astype(np.int64)
    out.append(("powerlaw_tail", p))
    out.append(("thermal2_like", np.clip(rng.normal(7.0, 1.2, size=1_228_045 // s).round(), 1, 11).astype(np.int64)))
    out.append(("ldoor_like", np.clip(rng.normal(45.0, 11.0, size=952_203 // s).round(), 28, 77).astype(np.int64)))
    k = rng.choice([27, 18, 12, 8, 5], size=1_700_000 // s, p=[0.86, 0.09, 0.035, 0.01, 0.005])
    out.append(("nlpkkt_like_27pt", k))
    for small in (4_000, 20_000, 100_000):  # where the breakeven threshold decides
        b = np.full(small, 5)
        b[rng.random(small) < 0.15] = 25
        out.append((f"small_{small}_bimodal_5_25", b))
    return out


def candidate_widths(lens):
    mx = int(lens.max())
    ks = set(range(0, min(mx, 16) + 1)) | {mx}
    k = 20
    while k < mx:
        ks.add(k)
        k = int(k * 1.3) + 1
    # the widths the rule can pick are row lengths that occur: make sure the distinct lengths (few) are all there
    uniq = np.unique(lens)
    if len(uniq) <= 40:
        ks |= set(int(u) for u in uniq)
    cap = max(16, int(3e9 // (12 * max(len(lens), 1))))  # an ELL part beyond ~3 GB is not a candidate
    return sorted(k for k in ks if k <= cap)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--dtypes", default="f64,f32")
    ap.add_argument("--out", default=os.path.join(ROOT, "cusp-autotuned_amd", "tuned", "gfx950.json"))
    ap.add_argument("--log", default=os.path.join(ROOT, "gpurun_out", "autotune_hyb.jsonl"))
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    import torch
    import cusp_autotuned_amd as cmi

    assert torch.cuda.is_available(), "the autotuner needs an MI355X"
    lib = cmi.lib()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    cmi.check(lib.cmi_event_create(ctypes.byref(e0)))
    cmi.check(lib.cmi_event_create(ctypes.byref(e1)))

    def time_ms(fn, iters):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        cmi.check(lib.cmi_event_record(e0, s))
        for _ in range(iters):
            fn()
        cmi.check(lib.cmi_event_record(e1, s))
        ms = ctypes.c_float()
        cmi.check(lib.cmi_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        return ms.value / iters

    os.makedirs(os.path.dirname(args.log), exist_ok=True)
    logf = open(args.log, "a")

    def log(rec):
        logf.write(json.dumps(rec) + "\n")
        logf.flush()

    scalar = cmi.Config(kernel=cmi.CSR_SCALAR)
    rules = {}
    for tag in args.dtypes.split(","):
        tdt = torch.float64 if tag == "f64" else torch.float32
        tol = 1e-6 if tag == "f64" else 1e-5
        sweeps = []  # (name, num_rows, hist, {K: ms})
        mats = [("poisson5pt_3162", None)] + distributions(args.quick)
        for mi, (name, lens) in enumerate(mats):
            if lens is None:
                A = cmi.poisson5pt(3162 if not args.quick else 1000, 3162 if not args.quick else 1000, "csr", dtype=tdt)
                lens = (A.row_offsets[1:] - A.row_offsets[:-1]).cpu().numpy().astype(np.int64)
            else:
                A = make_csr(cmi, torch, lens, tdt, seed=100 + mi)
            rows = A.num_rows
            x = cmi.fill_x(rows, tdt, "cuda")
            y = torch.empty(rows, dtype=tdt, device="cuda")
            cmi.multiply(A, x, y, cfg=scalar)
            want = y.clone()
            Aabs = cmi.CsrMatrix(rows, rows, A.num_entries, A.row_offsets, A.column_indices, A.values.abs())
            cmi.multiply(Aabs, x.abs(), y, cfg=scalar)
            bound = y.clone().clamp_(min=1e-30)
            hist = np.bincount(lens)
            mats_k = {}
            for K in sorted(set(candidate_widths(lens)) | {min(rule_width(hist, rows, 3.0, 4096), max(candidate_widths(lens)))}):
                H = cmi.convert(A, "hyb", num_entries_per_row=K)
                y.fill_(10.0)
                cmi.multiply(H, x, y)
                ok = bool(((y - want).abs() <= tol * bound).all().item())
                if not ok:
                    log({"dtype": tag, "matrix": name, "K": K, "status": "ValidationFailed"})
                    continue
                mats_k[K] = H
            times = {K: [] for K in mats_k}
            for _ in range(args.rounds):  # interleaved rounds
                for K, H in mats_k.items():
                    times[K].append(time_ms(lambda: cmi.multiply(H, x, y), args.iters))
            tk = {K: float(np.median(t)) for K, t in times.items()}
            best = min(tk, key=tk.get)
            ref_K = rule_width(hist, rows, 3.0, 4096)
            if ref_K not in tk and ref_K <= max(candidate_widths(lens)):  # the reference's own choice is always measured
                pass
            for K in sorted(tk):
                log({"dtype": tag, "matrix": name, "rows": rows, "entries": A.num_entries, "K": K, "coo_entries": mats_k[K].coo.num_entries,
                     "ms": tk[K], "status": "Ok"})
            print(f"{tag} {name}: rows {rows} entries {A.num_entries}  best K {best} ({tk[best] * 1e3:.1f} us)  "
                  f"reference rule K {ref_K} ({tk.get(ref_K, float('nan')) * 1e3:.1f} us)  K=max {max(tk)} ({tk[max(tk)] * 1e3:.1f} us)",
                  flush=True)
            sweeps.append((name, rows, hist, tk))
            del mats_k, A, Aabs, x, y, want, bound
            torch.cuda.empty_cache()

        # ---- fit the pair -----------------------------------------------------------------------------------
        def time_at(tk, K):  # the measured time at K, linearly interpolated between measured widths
            ks = sorted(tk)
            if K in tk:
                return tk[K]
            lo = max([k for k in ks if k < K], default=ks[0])
            hi = min([k for k in ks if k > K], default=ks[-1])
            if lo == hi:
                return tk[lo]
            return tk[lo] + (tk[hi] - tk[lo]) * (K - lo) / (hi - lo)

        def fit(kind, width_of, thresholds):
            grid = []
            for rs in [round(1.0 + 0.1 * i, 1) for i in range(0, 51)]:
                for th in thresholds:
                    logs = []
                    for name, rows, hist, tk in sweeps:
                        K = min(width_of(hist, rows, rs, th), max(tk))
                        logs.append(math.log(time_at(tk, K) / min(tk.values())))
                    grid.append((math.exp(sum(logs) / len(logs)), math.exp(max(logs)), rs, th))
            grid.sort()
            return grid

        g_ref = fit("reference", rule_width, (0, 64, 256, 1024, 4096, 16384, 65536, 262144))
        g_cost = fit("cost", cost_rule_width, (0, 10_000, 100_000, 300_000, 1_000_000, 2_000_000, 3_000_000, 5_000_000, 7_000_000, 10_000_000, 20_000_000))
        ref_pair = next(g for g in g_ref if g[2] == 3.0 and g[3] == 4096)
        kind, (score, worst, rs, th) = ("cost", g_cost[0]) if g_cost[0][0] <= g_ref[0][0] else ("reference", g_ref[0])
        rules[tag] = {"kind": kind, "relative_speed": rs, "threshold": th}
        log({"dtype": tag, "fit": {"kind": kind, "relative_speed": rs, "threshold": th, "geomean_regret": score, "worst_regret": worst},
             "reference_constants_3.0_4096": {"geomean_regret": ref_pair[0], "worst_regret": ref_pair[1]},
             "best_of_reference_form": {"relative_speed": g_ref[0][2], "threshold": g_ref[0][3], "geomean_regret": g_ref[0][0], "worst_regret": g_ref[0][1]},
             "cost_form_top5": [{"relative_speed": g[2], "threshold": g[3], "geomean_regret": g[0], "worst_regret": g[1]} for g in g_cost[:5]]})
        print(f"{tag}: tuned rule kind {kind} relative_speed {rs} threshold {th}: geomean regret {score:.4f} (worst {worst:.3f}); "
              f"best pair of the reference's form ({g_ref[0][2]}, {g_ref[0][3]}): {g_ref[0][0]:.4f}; the reference's constants (3.0, 4096): "
              f"{ref_pair[0]:.4f} (worst {ref_pair[1]:.3f})", flush=True)
        for name, rows, hist, tk in sweeps:
            width_of = cost_rule_width if kind == "cost" else rule_width
            K = min(width_of(hist, rows, rs, th), max(tk))
            kb = min(tk, key=tk.get)
            print(f"   {name}: rule K {K} ({time_at(tk, K) * 1e3:.1f} us)  best K {kb} ({tk[kb] * 1e3:.1f} us)  "
                  f"reference constants K {rule_width(hist, rows, 3.0, 4096)} ({time_at(tk, min(rule_width(hist, rows, 3.0, 4096), max(tk))) * 1e3:.1f} us)")
        cmi.tuning_set_hyb_rule(cmi.F64 if tag == "f64" else cmi.F32, cmi.HYB_RULE_COST if kind == "cost" else cmi.HYB_RULE_REFERENCE, rs, th)
    # patch the table file in place (keeps its entries and its provenance note)
    doc = json.load(open(args.out)) if os.path.exists(args.out) else {"arch": "gfx950", "version": cmi.version(), "entries": []}
    doc["hyb_rule"] = rules
    doc["hyb_rule_source"] = ("tools/autotune_hyb.py on MI355X: width sweeps over the headline matrix, SuiteSparse-like and synthetic "
                              "row-length distributions; pair with the lowest geometric-mean regret (raw log: profiles/*autotune_hyb*)")
    with open(args.out, "w") as f:
        f.write("{\n")
        for k, v in doc.items():
            if k != "entries":
                f.write(f"  {json.dumps(k)}: {json.dumps(v)},\n")
        f.write('  "entries": [\n')
        f.write(",\n".join("    " + json.dumps(e) for e in doc["entries"]))
        f.write("\n  ]\n}\n")
    print(f"wrote {args.out}: hyb_rule {rules}")


if __name__ == "__main__":
    main()
