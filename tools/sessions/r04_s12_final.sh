#!/bin/bash
# r4 final measurement set on the final tree: the driver's bench command (+ under rocprofv3 --kernel-trace --stats, cut into its phases), counters and
# timing of the long-row class (f64; f32 timing), the other formats' bench lines, CG, smoke()
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s12; mkdir -p $O
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
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  PMC_WAVEV=4 PMC_WAVEV_POL=3 PMC_WAVER=4 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/longpmc/p$i -- python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 > $O/long_manifest_$i.txt 2> $O/longpmc_$i.err
  rc=$?; echo "long-row pass $i ($set) exit $rc"; [ $rc -ge 124 ] && exit $rc
done
python3 tools/pmc_matrix_table.py $O/long_manifest_1.txt $O/longpmc $O/long_rows_pmc_final.json > $O/long_rows_pmc_final.txt 2>&1
find $O/longpmc -name "*kernel_trace.csv" -delete; find $O/longpmc -name "*counter_collection.csv" -delete
grep -E "^[a-z]|traffic_over|wait_any|lds_conflict|l2_hit" $O/long_rows_pmc_final.txt | cut -c1-120
PMC_WAVEV=4 PMC_WAVEV_POL=3 PMC_WAVER=4 PMC_C16=1 PMC_PLAN_AGAIN=1 timeout -k 10 400 python3 tools/pmc_matrix_probe.py thermal2,ldoor,nlpkkt120 --time > $O/long_rows_time_final.txt 2>&1; grep TIME $O/long_rows_time_final.txt | cut -c1-110
PMC_DTYPE=f32 PMC_WAVEV=4 PMC_WAVEV_POL=3 PMC_WAVER=4 PMC_PLAN_AGAIN=1 timeout -k 10 400 python3 tools/pmc_matrix_probe.py thermal2,ldoor,nlpkkt120 --time > $O/long_rows_time_final_f32.txt 2>&1; grep TIME $O/long_rows_time_final_f32.txt | cut -c1-110
for f in ell dia coo hyb; do python3 bench.py --format $f --no-cpu-baseline --steps 200 > $O/bench_n1_$f.json 2>/dev/null || echo "bench $f failed"; python3 -c "import json; d=json.load(open('$O/bench_n1_$f.json')); print('$f', d['value'], d['roofline']['frac'], d.get('roofline_cold',{}).get('frac'))"; done
tools/bin/cg_bench --iterations=200 > $O/cg_bench_csr.txt 2>&1; grep fused $O/cg_bench_csr.txt | cut -c1-200
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-250
