#!/bin/bash
# the round's closing numbers for the caller: CG per format through the C++ layer (+ rocprofv3 --stats of the CSR solve), the driver's bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s33; mkdir -p $O
for f in csr ell dia coo hyb; do tools/bin/cg_bench --iterations=200 --format=$f > $O/cg_bench_$f.txt 2>&1; echo "$f: $(grep fused $O/cg_bench_$f.txt | tail -1)"; done
CMI_COMPRESS_INDICES=1 tools/bin/cg_bench --iterations=200 > $O/cg_bench_csr_c16.txt 2>&1; echo "csr_c16: $(grep fused $O/cg_bench_csr_c16.txt | tail -1)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cgstats -o cg -- tools/bin/cg_bench --iterations=100 > $O/cg_under_rocprof.txt 2> $O/cg.err
find $O/cgstats -name "*kernel_trace.csv" -delete
head -8 $O/cgstats/cg_kernel_stats.csv | cut -c1-70,180-300
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { echo bench failed; tail -5 $O/bench_driver_cmd.err; exit 2; }
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("csr", d["value"], d["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["kernel_avg_over_ms_per_step"], r["kernel_avg_le_step_x1.02"]); print("cg", d["cg"]["us_per_iteration"]); c=d["compressed_index_plan"]; print("c16", c["kernel_avg_ms"], c["gflops"], c["cg_us_per_iteration"])
PY
for f in ell dia coo hyb; do python bench.py --format $f --no-cpu-baseline --steps 200 > $O/bench_n1_$f.json 2>/dev/null || echo "bench $f failed"; done
python - <<PY
import json
for f in ("ell","dia","coo","hyb"):
    e=json.loads(open("$O/bench_n1_%s.json"%f).read().strip().splitlines()[-1]); r=e["roofline"]; print(f, e["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["traffic"])
PY
