"""Print the kernels of a rocprofv3 --kernel-trace CSV in start order with their durations (tools).
usage: kernel_timeline.py <dir-or-csv> [first] [count]"""
import csv
import glob
import os
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
f = path if path.endswith(".csv") else sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = int(rows[first]["Start_Timestamp"]) if rows else 0
for r in rows[first:first + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f us  dur %8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
