#!/bin/bash
# COO plans on sorted entries = CSR kernels on plan-built row offsets: full suite, COO bench line (+ tile kernel beside it), CG on COO, formats PMC again
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s29; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 15 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
python bench.py --format coo --no-cpu-baseline --steps 200 > $O/bench_n1_coo.json 2>$O/bench_coo.err || { echo "bench coo failed"; tail -5 $O/bench_coo.err; }
python - <<PY
import json
e=json.loads(open("$O/bench_n1_coo.json").read().strip().splitlines()[-1]); r=e["roofline"]; print("coo", e["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["traffic"], e["config"]["kernel_config"]); print(json.dumps(e.get("coo_tile_kernel")))
PY
tools/bin/cg_bench --iterations=200 --format=coo > $O/cg_bench_coo.txt 2>&1; grep fused $O/cg_bench_coo.txt | tail -1
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/fmtpmc -o $pass -- python3 tools/pmc_probe.py csr,ell,dia,coo > $O/fmt_probe_$pass.json 2> $O/fmtpmc_$pass.err || { echo fmt pmc $pass failed; tail -3 $O/fmtpmc_$pass.err; }
  CMI_COO_PLAN_OFFSETS=0 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/tilepmc -o $pass -- python3 tools/pmc_probe.py coo > $O/tile_probe_$pass.json 2> $O/tilepmc_$pass.err || { echo tile pmc $pass failed; }
done
find $O/fmtpmc $O/tilepmc -name "*kernel_trace.csv" -delete
python tools/pmc_summary.py $O/fmtpmc $O/fmt_probe_FETCH_SIZE.json $O/formats_pmc.json > $O/formats_pmc.txt 2>&1
python tools/pmc_summary.py $O/tilepmc $O/tile_probe_FETCH_SIZE.json $O/coo_tile_pmc.json > $O/coo_tile_pmc.txt 2>&1
python - <<PY
import json
for f in ("formats_pmc","coo_tile_pmc"):
    doc=json.load(open("$O/%s.json"%f))
    for k in doc["kernels"]:
        if k["launches"]>=5 and "cmi::" in k["kernel"] and "axpby" not in k["kernel"]:
            print(f, k["kernel"][10:60], k["launches"], round(k["hbm_bytes_per_launch"]/1e6,1))
PY
