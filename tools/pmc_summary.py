#!/usr/bin/env python3
"""Turns the CSVs of separate rocprofv3 --pmc passes into profiles/<round>_pmc.json.

    python tools/pmc_summary.py <dir with *counter_collection.csv> <probe json line file> <out.json>

Per kernel: mean FETCH_SIZE / WRITE_SIZE per launch (KB -> bytes), the correction factors derived
from the calibration kernel's known byte counts (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE
reports half the bytes of a wide coalesced streaming read -- calibrate, then correct), and the
corrected HBM bytes per launch that bench.py reports as roofline.traffic."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, probe_file, out = sys.argv[1], sys.argv[2], sys.argv[3]
    probe = None
    for line in open(probe_file):
        line = line.strip()
        if line.startswith("{") and "calibration_kernel" in line:
            probe = json.loads(line)
    vals = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> [values]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                vals[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    def mean(kern_sub, counter):
        for k, c in vals.items():
            if kern_sub in k and counter in c:
                v = c[counter]
                v = v[len(v) // 2:]  # second half: caches warm, steady state
                return sum(v) / len(v), k
        return None, None
    cal_r, _ = mean(probe["calibration_kernel"], "FETCH_SIZE")
    cal_w, _ = mean(probe["calibration_kernel"], "WRITE_SIZE")
    fr = probe["calibration_read_bytes"] / (cal_r * 1024) if cal_r else None
    fw = probe["calibration_write_bytes"] / (cal_w * 1024) if cal_w else None
    kernels = []
    for k in vals:
        r = vals[k].get("FETCH_SIZE")
        w = vals[k].get("WRITE_SIZE")
        rec = {"kernel": k, "launches": len(r or w or [])}
        if r:
            r = r[len(r) // 2:]
            rec["FETCH_SIZE_KB_per_launch"] = sum(r) / len(r)
        if w:
            w = w[len(w) // 2:]
            rec["WRITE_SIZE_KB_per_launch"] = sum(w) / len(w)
        if r and w and fr and fw:
            rec["hbm_read_bytes_per_launch"] = rec["FETCH_SIZE_KB_per_launch"] * 1024 * fr
            rec["hbm_write_bytes_per_launch"] = rec["WRITE_SIZE_KB_per_launch"] * 1024 * fw
            rec["hbm_bytes_per_launch"] = rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"]
        for cname, cv in vals[k].items():
            if cname not in ("FETCH_SIZE", "WRITE_SIZE"):
                cv = cv[len(cv) // 2:]
                rec[cname + "_per_launch"] = sum(cv) / len(cv)
        kernels.append(rec)
    doc = {"method": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE); counters in KB; corrected by the factors "
                     "below, calibrated on axpby_kernel (known bytes, 16 B/lane streaming) per MI355X_MICROARCH.md HBM section",
           "probe": probe, "fetch_correction_factor": fr, "write_correction_factor": fw, "kernels": kernels}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1)[:3000])


if __name__ == "__main__":
    main()
