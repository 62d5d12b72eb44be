#!/bin/bash
# r4 session 5: pieces of 3 against pieces of 4 in ONE process (ldoor-like, nlpkkt120-like), configs[4]'s per-rank shape on one GPU,
# counters of the headline kernel / the 16-bit plan / the packed tiles on the headline matrix, the -m gpu suite on the new tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s5; mkdir -p $O
PMC_WAVEV= PMC_WAVER=4 PMC_WAVER_CAP=0,3,4,3,4 PMC_PACKED=0 PMC_PLAN_AGAIN=1 timeout -k 10 400 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/caps_time.txt 2>&1; grep TIME $O/caps_time.txt | cut -c1-100
timeout -k 10 500 python3 tools/configs4_rank_shape_probe.py > $O/configs4_rank_shape.txt 2>&1; grep -v RESULT $O/configs4_rank_shape.txt | cut -c1-260
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/fmtpmc -o $pass -- python3 tools/pmc_probe.py csr,csr16,csr16p > $O/fmt_probe_$pass.json 2> $O/fmtpmc_$pass.err; rc=$?; echo "fmt pmc $pass exit $rc"; [ $rc -ge 124 ] && exit $rc
done
find $O/fmtpmc -name "*kernel_trace.csv" -delete
python3 tools/pmc_summary.py $O/fmtpmc $O/fmt_probe_FETCH_SIZE.json $O/headline_plans_pmc.json > $O/headline_plans_pmc.txt 2>&1; cat $O/headline_plans_pmc.txt | cut -c1-220
find $O/fmtpmc -name "*counter_collection.csv" -delete
timeout -k 10 1150 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; echo "full pytest exit $?"; tail -n 8 $O/pytest_gpu.txt | cut -c1-250
