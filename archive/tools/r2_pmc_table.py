#!/usr/bin/env python3
"""Joins the rocprofv3 --pmc passes of `tools/bin/r2_probe --pmc K` into one table: variant x counter (mean per launch).

    python tools/r2_pmc_table.py <manifest file (stdout of one r2_probe --pmc run)> <dir with one sub-directory per pass> [out.json]

The probe launches every variant K times in manifest order; the counter rows are mapped back by dispatch order among the
probe's own kernels (csr_stream / mix / csrx / dia_row / diax).  The first launch of every variant is dropped (cold).
FETCH_SIZE is reported raw and x2 (MI355X_MICROARCH.md: gfx950 tallies 128-byte requests at 64 bytes for wide streaming reads)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = ("csr_stream_kernel", "mix_kernel", "csrx_kernel", "dia_row", "diax_kernel", "csr_balanced_kernel", "ell_row_kernel")


def main():
    manifest, root = sys.argv[1], sys.argv[2]
    variants = []
    for line in open(manifest):
        if line.startswith("MANIFEST\t"):
            _, name, k, b = line.rstrip("\n").split("\t")
            variants.append((name, int(k), float(b)))
    table = defaultdict(dict)
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        rows = defaultdict(dict)  # dispatch id -> {counter: value}, kernel name
        names = {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                d = int(r["Dispatch_Id"])
                rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                names[d] = r["Kernel_Name"]
        seq = [d for d in sorted(rows) if any(k in names[d] for k in KERNELS)]
        need = sum(k for _, k, _ in variants)
        if len(seq) != need:
            print(f"# {f}: {len(seq)} probe dispatches, manifest wants {need}: skipped", file=sys.stderr)
            continue
        pos = 0
        for name, k, _ in variants:
            ds = seq[pos + 1:pos + k]  # drop the first (cold) launch
            pos += k
            for c in rows[ds[0]]:
                table[name][c] = sum(rows[d][c] for d in ds) / len(ds)
            table[name]["kernel"] = names[ds[0]][:60]
    counters = sorted({c for v in table.values() for c in v if c != "kernel"})
    print("variant".ljust(40) + "".join(c[-22:].rjust(24) for c in counters))
    for name, _, _ in variants:
        if name in table:
            print(name.ljust(40) + "".join((f"{table[name].get(c, float('nan')):.4g}").rjust(24) for c in counters))
    if len(sys.argv) > 3:
        json.dump({"counters": counters, "variants": {n: table[n] for n, _, _ in variants if n in table},
                   "algorithmic_bytes": {n: b for n, _, b in variants}}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
