#!/bin/bash
# closing numbers (final code, csr_wave + lane-strided csr_stream + re-tuned short-row keys): full GPU suite, smoke, the driver's bench
# command (+ rocprofv3 --stats of the same command), CG per format, the COO bench line, FETCH_SIZE / WRITE_SIZE passes per format
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s58; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 4 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { echo bench failed; tail -5 $O/bench_driver_cmd.err; exit 2; }
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("csr", d["value"], d["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["kernel_avg_over_ms_per_step"], d["config"]["kernel_config"]); print("cg", d["cg"]["us_per_iteration"], d["cg"]["us_per_marginal_iteration"]); c=d["compressed_index_plan"]; print("c16", c["kernel_avg_ms"], c["gflops"], c["cg_us_per_iteration"], c["speedup_over_the_headline_kernel"]); print(d["cpu_baseline"]["value"], d["cpu_baseline_omp"]["value"])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || { echo rocprof bench failed; exit 3; }
find $O/stats -name "*kernel_trace.csv" -delete
grep "csr_wave\|csr_stream\|cg_\|dot_fold" $O/stats/bench_kernel_stats.csv | sed 's/(long[^"]*"/"/' | cut -c1-130
for f in csr ell dia coo hyb; do tools/bin/cg_bench --iterations=200 --format=$f > $O/cg_bench_$f.txt 2>&1; echo "$f: $(grep fused $O/cg_bench_$f.txt | tail -1 | cut -c1-170)"; done
CMI_COMPRESS_INDICES=1 tools/bin/cg_bench --iterations=200 > $O/cg_bench_csr_c16.txt 2>&1; echo "csr_c16: $(grep fused $O/cg_bench_csr_c16.txt | tail -1 | cut -c1-170)"
python bench.py --format coo --no-cpu-baseline --steps 200 > $O/bench_n1_coo.json 2>$O/bench_coo.err || { echo "bench coo failed"; tail -5 $O/bench_coo.err; }
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/fmtpmc -o $pass -- python3 tools/pmc_probe.py csr,ell,dia,coo > $O/fmt_probe_$pass.json 2> $O/fmtpmc_$pass.err || { echo fmt pmc $pass failed; tail -3 $O/fmtpmc_$pass.err; }
done
find $O/fmtpmc -name "*kernel_trace.csv" -delete
python tools/pmc_summary.py $O/fmtpmc $O/fmt_probe_FETCH_SIZE.json $O/formats_pmc.json > $O/formats_pmc.txt 2>&1
python - <<PY
import json
doc=json.load(open("$O/formats_pmc.json"))
for k in doc["kernels"]:
    if k["launches"]>=5 and "cmi::" in k["kernel"] and "axpby" not in k["kernel"]:
        print("pmc", k["kernel"][10:60], k["launches"], round(k["hbm_bytes_per_launch"]/1e6,1))
e=json.loads(open("$O/bench_n1_coo.json").read().strip().splitlines()[-1]); r=e["roofline"]; print("coo", e["ms_per_step"], r["frac"], r["kernel_avg_ms"], e["config"]["kernel_config"])
PY
