#!/bin/bash
# r4 final measurement set on the final tree (after the regret run's rule changes): the driver's bench command (+ under rocprofv3 --kernel-trace --stats, cut into
# its phases), FETCH / WRITE counters of the three kernels that read the headline matrix, the long-row class's timing (f64, f32), the other formats' bench lines, CG, smoke()
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s23; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { echo bench failed; tail -5 $O/bench_driver_cmd.err; exit 2; }
python3 -c "
import json; d=json.load(open('$O/bench_driver_cmd.json'))
print({k: d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['kernel_avg_ms'], d.get('roofline_cold',{}).get('frac'), d.get('cg'))
for k in ('compressed_index_plan','packed_tile_plan'):
    v=d.get(k,{}); print(k, {q: v.get(q) for q in ('granted','kernel_avg_ms','moved_frac_of_peak','speedup_over_the_headline_kernel','speedup_over_the_16_bit_plan','traffic','error')}, v.get('cold'))
print(d.get('cpu_baseline'))
"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || { echo rocprof bench failed; tail -3 $O/rocprof.err; exit 3; }
python3 tools/trace_phases.py $O/stats/bench_kernel_trace.csv $O/bench_under_rocprof.json > $O/trace_phases.txt 2>&1; cat $O/trace_phases.txt | cut -c1-200
rm -f $O/stats/bench_kernel_trace.csv
head -8 $O/stats/bench_kernel_stats.csv | cut -c1-200
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/fmtpmc -o $pass -- python3 tools/pmc_probe.py csr,csr16,csr16p > $O/fmt_probe_$pass.json 2> $O/fmtpmc_$pass.err; rc=$?; echo "fmt pmc $pass exit $rc"; [ $rc -ge 124 ] && exit $rc
done
find $O/fmtpmc -name "*kernel_trace.csv" -delete
python3 tools/pmc_summary.py $O/fmtpmc $O/fmt_probe_FETCH_SIZE.json $O/headline_plans_pmc_final.json > $O/headline_plans_pmc_final.txt 2>&1
python3 - <<PY
import json
d=json.load(open("$O/headline_plans_pmc_final.json"))
for k in d["kernels"]:
    if "hbm_bytes_per_launch" in k and ("csr_wave" in k["kernel"]): print(k["kernel"][:60], k["launches"], round(k["hbm_read_bytes_per_launch"]/1e6,1), round(k["hbm_write_bytes_per_launch"]/1e6,1), round(k["hbm_bytes_per_launch"]/1e6,1))
print(d["probe"]["spmv_algorithmic_bytes"])
PY
find $O/fmtpmc -name "*counter_collection.csv" -delete
PMC_WAVEV=4 PMC_WAVEV_POL=3 PMC_WAVER=4 PMC_C16=1 PMC_PLAN_AGAIN=1 timeout -k 10 400 python3 tools/pmc_matrix_probe.py thermal2,ldoor,nlpkkt120 --time > $O/long_rows_time_final.txt 2>&1; grep TIME $O/long_rows_time_final.txt | cut -c1-110
PMC_DTYPE=f32 PMC_WAVEV=4 PMC_WAVEV_POL=3 PMC_WAVER=4 PMC_PLAN_AGAIN=1 timeout -k 10 400 python3 tools/pmc_matrix_probe.py thermal2,ldoor,nlpkkt120 --time > $O/long_rows_time_final_f32.txt 2>&1; grep TIME $O/long_rows_time_final_f32.txt | cut -c1-110
for f in ell dia coo hyb; do python3 bench.py --format $f --no-cpu-baseline --steps 200 > $O/bench_n1_$f.json 2>/dev/null || echo "bench $f failed"; python3 -c "import json; d=json.load(open('$O/bench_n1_$f.json')); print('$f', d['value'], d['roofline']['frac'], d.get('roofline_cold',{}).get('frac'))"; done
tools/bin/cg_bench --iterations=200 > $O/cg_bench_csr.txt 2>&1; grep fused $O/cg_bench_csr.txt | cut -c1-200
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-250
