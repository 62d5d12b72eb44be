#!/bin/bash
# r3 session 13: the measurement set on the final tree -- the driver's bench command (+ under rocprofv3 --kernel-trace --stats), PMC passes for
# the headline kernel and for the long-row matrices (plan's choice, table csr_stream, csr_wavev), formats, CG (single GPU and sharded with a
# 1-rank RCCL communicator), smoke()
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s13; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { echo bench failed; tail -5 $O/bench_driver_cmd.err; exit 2; }
python3 -c "import json; d=json.load(open('$O/bench_driver_cmd.json')); print({k: d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['kernel_avg_ms'], d.get('roofline_cold'), d.get('cg'))"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || { echo rocprof bench failed; tail -3 $O/rocprof.err; exit 3; }
find $O/stats -name "*kernel_trace.csv" -delete
head -6 $O/stats/bench_kernel_stats.csv | cut -c1-220
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/fmtpmc -o $pass -- python3 tools/pmc_probe.py csr,ell,dia,coo > $O/fmt_probe_$pass.json 2> $O/fmtpmc_$pass.err; rc=$?; echo "fmt pmc $pass exit $rc"; [ $rc -ge 124 ] && exit $rc
done
find $O/fmtpmc -name "*kernel_trace.csv" -delete
python3 tools/pmc_summary.py $O/fmtpmc $O/fmt_probe_FETCH_SIZE.json $O/formats_pmc.json > $O/formats_pmc.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  PMC_WAVEV=4 PMC_WAVEV_POL=3 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/longpmc/p$i -- python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 > $O/long_manifest_$i.txt 2> $O/longpmc_$i.err
  rc=$?; echo "long-row pass $i ($set) exit $rc"; [ $rc -ge 124 ] && exit $rc
done
python3 tools/pmc_matrix_table.py $O/long_manifest_1.txt $O/longpmc $O/long_rows_pmc.json > $O/long_rows_pmc.txt 2>&1
find $O/longpmc -name "*kernel_trace.csv" -delete; find $O/longpmc -name "*counter_collection.csv" -delete
grep -E "^[a-z]|traffic_over|wait_any|lds_conflict|l2_hit" $O/long_rows_pmc.txt
PMC_WAVEV=4 PMC_WAVEV_POL=3 timeout -k 10 300 python3 tools/pmc_matrix_probe.py thermal2,ldoor,nlpkkt120 --time > $O/long_rows_time.txt 2>&1; grep TIME $O/long_rows_time.txt | cut -c1-110
for f in ell dia coo hyb; do python3 bench.py --format $f --no-cpu-baseline --steps 200 > $O/bench_n1_$f.json 2>/dev/null || echo "bench $f failed"; done
tools/bin/cg_bench --iterations=200 > $O/cg_bench_csr.txt 2>&1; grep fused $O/cg_bench_csr.txt
for f in ell dia coo hyb; do tools/bin/cg_bench --iterations=100 --format=$f > $O/cg_bench_$f.txt 2>&1; grep fused $O/cg_bench_$f.txt | tail -1; done
tools/bin/cmi_launch -n 1 -- tools/bin/cg_bench --sharded --grid=3162 --iterations=200 > $O/cg_sharded_1rank.txt 2>&1; grep -E "sharded|per step|cusp::krylov" $O/cg_sharded_1rank.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cgstats -o cg -- tools/bin/cg_bench --iterations=100 > $O/cg_under_rocprof.txt 2> $O/cg.err
find $O/cgstats -name "*kernel_trace.csv" -delete
head -8 $O/cgstats/cg_kernel_stats.csv | cut -c1-160
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
