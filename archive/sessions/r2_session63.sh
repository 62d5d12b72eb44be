#!/bin/bash
# session 63 (final code: + the 16-bit copy tiled per wave): FETCH_SIZE / WRITE_SIZE passes for the copy's kernel, rocprofv3 --stats of the
# driver's bench command, then that command itself
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s63; mkdir -p $O
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/c16pmc -o $pass -- python3 tools/pmc_probe.py csr16 > $O/c16_probe_$pass.json 2> $O/c16pmc_$pass.err || { echo c16 pmc $pass failed; tail -3 $O/c16pmc_$pass.err; }
done
find $O/c16pmc -name "*kernel_trace.csv" -delete
python tools/pmc_summary.py $O/c16pmc $O/c16_probe_FETCH_SIZE.json $O/c16_pmc.json > $O/c16_pmc.txt 2>&1
python - <<PY
import json
doc=json.load(open("$O/c16_pmc.json"))
for k in doc["kernels"]:
    if k["launches"]>=5 and "cmi::" in k["kernel"]: print("pmc", k["kernel"][10:60], k["launches"], round(k["hbm_bytes_per_launch"]/1e6,1))
PY
cp $O/c16_pmc.json profiles/r02_c16_pmc.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || { echo rocprof bench failed; exit 3; }
find $O/stats -name "*kernel_trace.csv" -delete
grep "csr_wave\|csr_stream\|cg_\|dot_fold" $O/stats/bench_kernel_stats.csv | sed 's/(long[^"]*"/"/' | cut -c1-130
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { echo bench failed; tail -5 $O/bench_driver_cmd.err; exit 2; }
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("csr", d["value"], d["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["kernel_avg_over_ms_per_step"], r["traffic"]); print("cg", d["cg"]["us_per_iteration"], d["cg"]["us_per_marginal_iteration"]); c=d["compressed_index_plan"]; print("c16", c["kernel_avg_ms"], c["gflops"], c["cg_us_per_iteration"], c["speedup_over_the_headline_kernel"], c["traffic"], c["moved_frac_of_peak"])
PY
