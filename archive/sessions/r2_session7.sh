#!/bin/bash
# Round-2 GPU session 7: COO re-tune, then the measurement set: driver's bench command (+ under rocprofv3 --stats), formats, CG, PMC
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s7; mkdir -p $O
timeout -k 10 600 python tools/autotune.py --formats coo --merge --skip-synthetic --log $O/autotune_coo.jsonl > $O/autotune_coo.txt 2>&1; rc=$?; echo "autotune coo exit $rc"; tail -n 5 $O/autotune_coo.txt
cp cusp-autotuned_amd/tuned/gfx950.json $O/table_after.json
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { echo bench failed; tail -5 $O/bench_driver_cmd.err; exit 2; }
cat $O/bench_driver_cmd.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || { echo rocprof bench failed; exit 3; }
find $O/stats -name "*kernel_trace.csv" -delete
head -8 $O/stats/bench_kernel_stats.csv | cut -c1-200
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/fmtpmc -o $pass -- python3 tools/pmc_probe.py csr,ell,dia,coo > $O/fmt_probe_$pass.json 2> $O/fmtpmc_$pass.err || { echo fmt pmc $pass failed; tail -3 $O/fmtpmc_$pass.err; }
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/hybpmc -o $pass -- python3 tools/pmc_probe.py hyb > $O/hyb_probe_$pass.json 2> $O/hybpmc_$pass.err || { echo hyb pmc $pass failed; }
done
find $O/fmtpmc $O/hybpmc -name "*kernel_trace.csv" -delete
python tools/pmc_summary.py $O/fmtpmc $O/fmt_probe_FETCH_SIZE.json $O/formats_pmc.json > $O/formats_pmc.txt 2>&1
python tools/pmc_summary.py $O/hybpmc $O/hyb_probe_FETCH_SIZE.json $O/hyb_pmc.json > $O/hyb_pmc.txt 2>&1
for f in ell dia coo hyb; do python bench.py --format $f --no-cpu-baseline --steps 200 > $O/bench_n1_$f.json 2>/dev/null || echo "bench $f failed"; done
tools/bin/cg_bench --iterations=200 > $O/cg_bench_csr.txt 2>&1; cat $O/cg_bench_csr.txt
for f in ell dia coo hyb; do tools/bin/cg_bench --iterations=100 --format=$f > $O/cg_bench_$f.txt 2>&1; grep fused $O/cg_bench_$f.txt | tail -1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cgstats -o cg -- tools/bin/cg_bench --iterations=100 > $O/cg_under_rocprof.txt 2> $O/cg.err
find $O/cgstats -name "*kernel_trace.csv" -delete
head -8 $O/cgstats/cg_kernel_stats.csv | cut -c1-160
