#!/bin/bash
# r4 session 6: counters of the headline kernel / the 16-bit plan / the packed tiles on the headline matrix (FETCH, WRITE), thermal2-like wave tiles
# V = 1 against the table kernel in one process, the whole -m gpu suite on the final tree, smoke()
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s6; mkdir -p $O
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/fmtpmc -o $pass -- python3 tools/pmc_probe.py csr,csr16,csr16p > $O/fmt_probe_$pass.json 2> $O/fmtpmc_$pass.err; rc=$?; echo "fmt pmc $pass exit $rc"; [ $rc -ge 124 ] && exit $rc
done
find $O/fmtpmc -name "*kernel_trace.csv" -delete
python3 tools/pmc_summary.py $O/fmtpmc $O/fmt_probe_FETCH_SIZE.json $O/headline_plans_pmc.json > $O/headline_plans_pmc.txt 2>&1
python3 - <<PY
import json
d=json.load(open("$O/headline_plans_pmc.json"))
for k in d["kernels"]:
    if "hbm_bytes_per_launch" in k and ("csr_wave" in k["kernel"]): print(k["kernel"][:60], k["launches"], round(k["hbm_read_bytes_per_launch"]/1e6,1), round(k["hbm_write_bytes_per_launch"]/1e6,1), round(k["hbm_bytes_per_launch"]/1e6,1))
print(d["probe"]["spmv_algorithmic_bytes"])
PY
find $O/fmtpmc -name "*counter_collection.csv" -delete
PMC_WAVEV=1,2 PMC_WAVEV_POL=0 PMC_WAVER= PMC_PLAN_AGAIN=1 timeout -k 10 300 python3 tools/pmc_matrix_probe.py thermal2 --time > $O/thermal2_time.txt 2>&1; grep TIME $O/thermal2_time.txt | cut -c1-110
timeout -k 10 1150 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "full pytest exit $?"; tail -n 8 $O/pytest_gpu.txt | cut -c1-250
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-250
