#!/bin/bash
# r4 session 4: the round-4 test file on the new tree (paired LDS writes, cap 3 / 4, packed 16-bit wave tiles, real-file branch, torchrun N = 1), the long-row
# timing again (settle by time; the plan's variant first AND last), bench.py with the packed leg, fold-ahead CG, counters of the DOT instance inside CG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s4; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_round4_gpu.py -x -q -m gpu > $O/tests.txt 2>&1; echo "pytest exit $?"; tail -12 $O/tests.txt | cut -c1-250
PMC_WAVEV=4 PMC_WAVEV_POL=3 PMC_WAVER=4 PMC_PLAN_AGAIN=1 timeout -k 10 400 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/long_rows_time.txt 2>&1; grep TIME $O/long_rows_time.txt | cut -c1-100
CMI_WAVER_CAP=4 PMC_WAVEV= PMC_WAVER=4 PMC_PACKED=0 timeout -k 10 400 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/long_rows_time_cap4.txt 2>&1; grep TIME $O/long_rows_time_cap4.txt | cut -c1-100
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { echo bench failed; tail -5 $O/bench_driver_cmd.err; }
python3 -c "
import json; d=json.load(open('$O/bench_driver_cmd.json'))
print({k: d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['kernel_avg_ms'], d.get('roofline_cold',{}).get('frac'), d.get('cg'))
for k in ('compressed_index_plan','packed_tile_plan'):
    v=d.get(k,{}); print(k, {q: v.get(q) for q in ('granted','kernel_avg_ms','moved_frac_of_peak','speedup_over_the_headline_kernel','speedup_over_the_16_bit_plan','error')}, v.get('cold'))
"
for fa in 0 1; do CMI_CG_FOLD_AHEAD=$fa timeout -k 10 200 tools/bin/cg_bench --iterations=200 > $O/cg_fold_ahead_$fa.txt 2>&1; echo "fold ahead $fa:"; grep -E "^fused" $O/cg_fold_ahead_$fa.txt | head -3; done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/cgpmc/p$i -- tools/bin/cg_bench --iterations=30 > $O/cg_pmc_$i.txt 2> $O/cg_pmc_$i.err
  rc=$?; echo "cg pmc pass $i ($set) exit $rc"; [ $rc -ge 124 ] && exit $rc
done
python3 tools/cg_pmc_table.py $O/cgpmc > $O/cg_pmc_table.txt 2>&1; cat $O/cg_pmc_table.txt | cut -c1-200
find $O/cgpmc -name "*kernel_trace.csv" -delete; find $O/cgpmc -name "*counter_collection.csv" -delete
