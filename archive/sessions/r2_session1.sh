#!/bin/bash
# Round-2 GPU session 1: probe timings, PMC passes over the probe (one counter set per pass, no trace domains besides
# --kernel-trace), the driver's bench command plain and under rocprofv3 --kernel-trace --stats, PMC of the other formats.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s1; mkdir -p $O
P=tools/bin/r2_probe
timeout -k 10 120 $P > $O/probe_timing.txt 2>&1 || { echo probe failed; tail -5 $O/probe_timing.txt; exit 1; }
tail -n 70 $O/probe_timing.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/p$i -o p -- $P --pmc 5 > $O/manifest_$i.txt 2> $O/pmc_$i.err || { echo "pmc pass $i failed"; tail -3 $O/pmc_$i.err; }
  echo "pass $i done $(date +%T)"
done
python3 tools/r2_pmc_table.py $O/manifest_1.txt $O/pmc $O/probe_pmc.json > $O/probe_pmc.txt 2> $O/probe_pmc.err; tail -n 60 $O/probe_pmc.txt
# remove the bulky raw CSVs except the counter collections (they are small); keep kernel traces out
find $O/pmc -name "*kernel_trace.csv" -delete
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { echo bench failed; tail -5 $O/bench_driver_cmd.err; exit 2; }
cat $O/bench_driver_cmd.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || { echo rocprof bench failed; exit 3; }
find $O/stats -name "*kernel_trace.csv" -delete
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/fmtpmc -o $pass -- python3 tools/pmc_probe.py csr,ell,dia,coo > $O/fmt_probe_$pass.json 2> $O/fmtpmc_$pass.err || { echo fmt pmc $pass failed; tail -3 $O/fmtpmc_$pass.err; }
done
find $O/fmtpmc -name "*kernel_trace.csv" -delete
python tools/pmc_summary.py $O/fmtpmc $O/fmt_probe_FETCH_SIZE.json $O/formats_pmc.json > $O/formats_pmc.txt 2>&1
for f in ell dia coo hyb; do python bench.py --format $f --no-cpu-baseline --steps 200 > $O/bench_n1_$f.json 2>/dev/null || echo "bench $f failed"; done
ls $O
