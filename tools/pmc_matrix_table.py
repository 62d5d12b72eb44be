#!/usr/bin/env python3
"""Joins the separate rocprofv3 --pmc passes of tools/pmc_matrix_probe.py into one table: variant x counter, mean per launch.

    python tools/pmc_matrix_table.py <manifest (stdout of one probe run)> <dir with one sub-directory per pass> [out.json]

Dispatches are mapped back to variants by order among the SpMV kernels (the first launch of every variant -- the validation
launch, cold -- is dropped).  FETCH_SIZE / WRITE_SIZE (KB) are converted to bytes with the factors measured on the calibration
kernel of the SAME pass (MI355X_MICROARCH.md, HBM: gfx950 tallies wide streaming reads at half their bytes)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = ("csr_stream_kernel", "csr_wave", "csr_balanced_kernel", "csr_vector_kernel")


def main():
    manifest, root = sys.argv[1], sys.argv[2]
    variants, cal = [], None
    for line in open(manifest):
        if line.startswith("MANIFEST\t"):
            f = line.rstrip("\n").split("\t")
            variants.append((f[1], int(f[2]), float(f[3]), f[4], f[5] if len(f) > 5 else ""))
        elif line.startswith("{") and "calibration_kernel" in line:
            cal = json.loads(line)
    table = defaultdict(dict)
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        rows, names = defaultdict(dict), {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                d = int(r["Dispatch_Id"])
                rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                names[d] = r["Kernel_Name"]
        seq = [d for d in sorted(rows) if any(k in names[d] for k in KERNELS)]
        need = sum(v[1] for v in variants)
        if len(seq) != need:
            print(f"# {f}: {len(seq)} SpMV dispatches, manifest wants {need}: skipped", file=sys.stderr)
            continue
        factor = {}
        if cal:
            cd = [d for d in sorted(rows) if cal["calibration_kernel"] in names[d]][1:]
            for c, known in (("FETCH_SIZE", cal["calibration_read_bytes"]), ("WRITE_SIZE", cal["calibration_write_bytes"])):
                vals = [rows[d][c] for d in cd if c in rows[d]]
                if vals:
                    factor[c] = known / (sum(vals) / len(vals) * 1024)
        pos = 0
        for name, k, _, _, _ in variants:
            ds = seq[pos + 1:pos + k]
            pos += k
            for c in rows[ds[0]]:
                table[name][c] = sum(rows[d][c] for d in ds) / len(ds)
                if c in factor:
                    table[name][c + "_bytes"] = table[name][c] * 1024 * factor[c]
                    table[name][c + "_factor"] = factor[c]
            table[name]["kernel"] = names[ds[0]][:70]
    out = {}
    for name, k, alg, exact, desc in variants:
        t = dict(table.get(name, {}))
        t.update({"algorithmic_bytes": alg, "bit_exact": exact, "config": desc})
        if "FETCH_SIZE_bytes" in t and "WRITE_SIZE_bytes" in t:
            t["hbm_bytes"] = t["FETCH_SIZE_bytes"] + t["WRITE_SIZE_bytes"]
            t["traffic_over_algorithmic"] = t["hbm_bytes"] / alg
        if "TCC_HIT_sum" in t and "TCC_MISS_sum" in t:
            t["l2_hit_rate"] = t["TCC_HIT_sum"] / max(1.0, t["TCC_HIT_sum"] + t["TCC_MISS_sum"])
        if "SQ_WAVE_CYCLES" in t and "SQ_WAIT_INST_ANY" in t:
            t["wait_any_over_wave_cycles"] = t["SQ_WAIT_INST_ANY"] / t["SQ_WAVE_CYCLES"]
        if "SQ_LDS_IDX_ACTIVE" in t and "SQ_LDS_BANK_CONFLICT" in t:
            t["lds_conflict_share"] = t["SQ_LDS_BANK_CONFLICT"] / max(1.0, t["SQ_LDS_IDX_ACTIVE"])
        out[name] = t
    for name, t in out.items():
        print(name)
        for c in sorted(t):
            v = t[c]
            print(f"    {c:34s} {v:.6g}" if isinstance(v, float) else f"    {c:34s} {v}")
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
