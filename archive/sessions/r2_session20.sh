#!/bin/bash
# final check of the round: full GPU suite, smoke, the driver's bench command plain and under rocprofv3 --stats, CG through the C++ layer
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s20; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 6 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { echo bench failed; tail -5 $O/bench_driver_cmd.err; exit 2; }
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("csr", d["value"], d["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["kernel_avg_over_ms_per_step"]); print("cg", d["cg"]["us_per_iteration"]); c=d["compressed_index_plan"]; print("c16", c["kernel_avg_ms"], c["gflops"], c["cg_us_per_iteration"]); print(d["cpu_baseline"]["value"], d["cpu_baseline_omp"]["value"])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || { echo rocprof bench failed; exit 3; }
find $O/stats -name "*kernel_trace.csv" -delete
head -9 $O/stats/bench_kernel_stats.csv | cut -c1-70,200-330
tools/bin/cg_bench --iterations=200 > $O/cg_bench_csr.txt 2>&1; grep fused $O/cg_bench_csr.txt
CMI_COMPRESS_INDICES=1 tools/bin/cg_bench --iterations=200 > $O/cg_bench_csr_c16.txt 2>&1; grep fused $O/cg_bench_csr_c16.txt
