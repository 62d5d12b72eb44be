#!/usr/bin/env python3
"""Re-fit the HYB width rule OFFLINE from a sweep log of tools/autotune_hyb.py (every (matrix, width K) -> time): regenerates the
tuning matrices' row-length histograms (seeded, numpy only -- no GPU), fits the reference's form, the cost model and the two-regime
cost model CMI_HYB_RULE_COST2 (one launch while the COO part averages <= 3 entries per row: `light_speed` slots per COO entry;
beyond: threshold + relative_speed per entry), prints the ranking and patches `hyb_rule` in the table.

    python tools/autotune_hyb_refit.py archive/profiles/r02_autotune_hyb.jsonl.gz [--out cusp-autotuned_amd/tuned/gfx950.json] [--dry-run]
"""
import argparse
import collections
import gzip
import itertools
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import autotune_hyb as ah  # noqa: E402  (rule restatements, distributions)

LIGHT_LIMIT = 3.0  # = kHybFusedMaxPerRow (csrc/common.h): the plan's own one-launch limit


def cost2_rule_width(hist, num_rows, relative_speed, threshold, light_speed):
    """CMI_HYB_RULE_COST2 on a host histogram, as cmi_hyb_entries_per_row walks it (k downwards, strict comparison)."""
    max_len = len(hist) - 1
    best, K = float(num_rows) * max_len, max_len
    longer, coo = 0, 0.0
    for k in range(max_len - 1, -1, -1):
        longer += int(hist[k + 1])
        coo += float(longer)
        light = coo <= LIGHT_LIMIT * num_rows
        cost = float(num_rows) * k + (light_speed * coo if light else threshold + relative_speed * coo)
        if cost < best:
            best, K = cost, k
    return K


def time_at(tk, K):
    ks = sorted(tk)
    if K in tk:
        return tk[K]
    lo = max([k for k in ks if k < K], default=ks[0])
    hi = min([k for k in ks if k > K], default=ks[-1])
    return tk[lo] if lo == hi else tk[lo] + (tk[hi] - tk[lo]) * (K - lo) / (hi - lo)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("log")
    ap.add_argument("--out", default=os.path.join(ah.ROOT, "cusp-autotuned_amd", "tuned", "gfx950.json"))
    ap.add_argument("--dry-run", action="store_true")
    args = ap.parse_args()
    opener = gzip.open if args.log.endswith(".gz") else open
    times = collections.defaultdict(dict)
    for line in opener(args.log, "rt"):
        r = json.loads(line)
        if "K" in r and r.get("status") == "Ok":
            times[(r["dtype"], r["matrix"])][r["K"]] = r["ms"]
    hists = {}
    m = 3162
    lens = np.full(m * m, 5, np.int64)  # poisson5pt: 5 minus the missing neighbours on the boundary
    idx = np.arange(m * m)
    lens -= (idx % m == 0).astype(np.int64) + (idx % m == m - 1) + (idx < m) + (idx >= m * (m - 1))
    hists["poisson5pt_3162"] = (len(lens), np.bincount(lens))
    for name, l in ah.distributions(False):
        hists[name] = (len(l), np.bincount(l))
    rules = {}
    for tag in ("f64", "f32"):
        sweeps = [(name, hists[name][0], hists[name][1], tk) for (dt, name), tk in times.items() if dt == tag and name in hists]
        if not sweeps:
            continue

        def score(width_of):
            logs = [math.log(time_at(tk, min(width_of(hist, rows), max(tk))) / min(tk.values())) for _, rows, hist, tk in sweeps]
            return math.exp(sum(logs) / len(logs)), math.exp(max(logs))

        ref = score(lambda h, n: ah.rule_width(h, n, 3.0, 4096))
        speeds = [round(0.8 + 0.1 * i, 1) for i in range(0, 33)]
        ths = (0, 100_000, 300_000, 500_000, 1_000_000, 2_000_000, 3_000_000, 5_000_000)
        g_cost = sorted((score(lambda h, n: ah.cost_rule_width(h, n, rs, th)) + (rs, th)) for rs, th in itertools.product(speeds, ths))
        g2 = sorted((score(lambda h, n: cost2_rule_width(h, n, rs, th, a)) + (rs, th, a))
                    for rs, th, a in itertools.product((0.8, 0.9, 1.0, 1.1, 1.2, 1.3, 1.4, 1.6), ths, (1.0, 1.25, 1.5, 1.75, 2.0, 2.5, 3.0, 3.5)))
        print(f"{tag}: reference constants (3.0, 4096): {ref[0]:.4f} (worst {ref[1]:.3f}); best cost (rs {g_cost[0][2]}, th {g_cost[0][3]}): "
              f"{g_cost[0][0]:.4f} (worst {g_cost[0][1]:.3f}); best cost2 (rs {g2[0][2]}, th {g2[0][3]}, light {g2[0][4]}): {g2[0][0]:.4f} (worst {g2[0][1]:.3f})")
        s, w, rs, th, a = g2[0]
        for name, rows, hist, tk in sweeps:
            K = min(cost2_rule_width(hist, rows, rs, th, a), max(tk))
            kb = min(tk, key=tk.get)
            print(f"   {name}: cost2 K {K} ({time_at(tk, K) * 1e3:.1f} us)  best K {kb} ({tk[kb] * 1e3:.1f} us)")
        if g2[0][0] <= g_cost[0][0]:
            rules[tag] = {"kind": "cost2", "relative_speed": rs, "threshold": th, "light_speed": a}
        else:
            rules[tag] = {"kind": "cost", "relative_speed": g_cost[0][2], "threshold": g_cost[0][3]}
    print("hyb_rule", rules)
    if args.dry_run:
        return
    doc = json.load(open(args.out))
    doc["hyb_rule"] = rules
    doc["hyb_rule_source"] = ("tools/autotune_hyb.py on MI355X (width sweeps over the headline matrix, SuiteSparse-like and synthetic row-length "
                              "distributions), rule fitted from that log by tools/autotune_hyb_refit.py (raw log: profiles/*autotune_hyb*)")
    with open(args.out, "w") as f:
        f.write("{\n")
        for k, v in doc.items():
            if k != "entries":
                f.write(f"  {json.dumps(k)}: {json.dumps(v)},\n")
        f.write('  "entries": [\n')
        f.write(",\n".join("    " + json.dumps(e) for e in doc["entries"]))
        f.write("\n  ]\n}\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
