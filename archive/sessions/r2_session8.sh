#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s8; mkdir -p $O
tools/bin/cg_bench --iterations=200 > $O/cg_bench_csr.txt 2>&1; cat $O/cg_bench_csr.txt
for f in ell dia coo hyb; do tools/bin/cg_bench --iterations=100 --format=$f > $O/cg_bench_$f.txt 2>&1; grep fused $O/cg_bench_$f.txt | tail -1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cgstats -o cg -- tools/bin/cg_bench --iterations=100 > $O/cg_under_rocprof.txt 2> $O/cg.err
find $O/cgstats -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/s8/cgstats/cg_kernel_stats.csv')):
    n=r['Name']
    if 'cmi::' in n: print(f"  {n.split('(')[0].replace('void cmi::','')[:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 12 $O/pytest_gpu.txt
