#!/bin/bash
# r3 session 40: rocprofv3 --kernel-trace --stats of tools/bin/cg_bench --solvers: the durations of the solvers' fused vector passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s40; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o solvers -- tools/bin/cg_bench --solvers --iterations=60 > $O/cg_bench_solvers_under_rocprof.txt 2> $O/rocprof.err || { echo rocprof failed; tail -3 $O/rocprof.err; exit 3; }
rm -f $O/stats/solvers_kernel_trace.csv
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3s40/stats/*kernel_stats.csv')[0]
for r in csv.reader(open(f)):
    if r[0]=="Name": continue
    print(f"{r[0][:70]:70s} calls {r[1]:>6s} avg {float(r[3])/1e3:8.1f} us")
PY
