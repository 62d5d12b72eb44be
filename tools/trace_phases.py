#!/usr/bin/env python3
"""Cuts a rocprofv3 --kernel-trace of `bench.py` into the phases bench.py says it launched (its `kernel_launch_ledger`): the replay
roofline leg, the timed steps and the cold leg all launch the SAME kernel, so `--stats` gives one average for the three; per phase
the trace must agree with the line's own HIP-event figures (`roofline.kernel_avg_ms`, `roofline_cold.kernel_avg_ms`).

    rocprofv3 --kernel-trace --stats --output-format csv -d out -o bench -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > line.json
    python3 tools/trace_phases.py out/bench_kernel_trace.csv line.json > profiles/rNN_bench_driver_cmd_trace_phases.txt
"""
import csv
import json
import re
import sys
from collections import Counter


def main():
    trace, line = sys.argv[1], json.load(open(sys.argv[2]))
    ledger = line.get("kernel_launch_ledger")
    if not ledger:
        raise SystemExit("the bench line has no kernel_launch_ledger (N > 1, or an old bench.py)")
    rows = []
    with open(trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    # the hot path's kernel: the most launched SpMV instance without the fused dot (its last template argument is `false`)
    spmv = re.compile(r"cmi::(csr_wave|csr_wavev|csr_wavex|csr_stream|csr_balanced|ell_row|ell_slices|dia_row2?|coo_tile|hyb_tile)_kernel<")
    counts = Counter(n for _, n, _ in rows if spmv.search(n) and not re.search(r"true\s*>\(", n.split(">(")[0] + ">("))
    name = counts.most_common(1)[0][0]
    durs = [d for _, n, d in rows if n == name]
    print(f"kernel: {name[:110]}")
    print(f"launches in the trace: {len(durs)}; in the ledger: {sum(n for _, n in ledger)} (the trace also holds the launches of later legs: CG's first multiplies, the opt-in plan's reference run)")
    pos = 0
    out = {}
    for phase, n in ledger:
        seg = durs[pos:pos + n]
        pos += n
        if seg:
            out[phase] = sum(seg) / len(seg) / 1e3
            print(f"  {phase:18s} {len(seg):5d} launches   average {out[phase]:8.2f} us   min {min(seg) / 1e3:8.2f}   max {max(seg) / 1e3:8.2f}")
    rest = durs[pos:]
    if rest:
        print(f"  {'(later legs)':18s} {len(rest):5d} launches   average {sum(rest) / len(rest) / 1e3:8.2f} us")
    ev = line["roofline"]["kernel_avg_ms"] * 1e3
    print(f"roofline leg:  trace {out.get('roofline', float('nan')):.2f} us  vs  HIP events in the line {ev:.2f} us   ({out.get('roofline', 0) / ev:.4f})")
    if "roofline_cold" in line and "kernel_avg_ms" in line["roofline_cold"] and "roofline_cold" in out:
        evc = line["roofline_cold"]["kernel_avg_ms"] * 1e3
        print(f"cold leg:      trace {out['roofline_cold']:.2f} us  vs  HIP events in the line {evc:.2f} us   ({out['roofline_cold'] / evc:.4f})")
    print(f"all launches of this kernel (what --stats averages): {sum(durs) / len(durs) / 1e3:.2f} us over {len(durs)}")


if __name__ == "__main__":
    main()
