#!/bin/bash
# session 60: soak of the final code (every kernel thousands of times, results never change), then the driver's bench command once more
# (its `traffic` now read from the closing session's PMC summary)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s60; mkdir -p $O
timeout -k 10 800 python tools/soak.py > $O/soak.txt 2>&1; echo "soak exit $?"; grep -v amdgpu.ids $O/soak.txt | cut -c1-200
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { echo bench failed; tail -5 $O/bench_driver_cmd.err; exit 2; }
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("csr", d["value"], d["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["kernel_avg_over_ms_per_step"], r["traffic"], r["traffic_source"][:40]); print("cg", d["cg"]["us_per_iteration"], d["cg"]["us_per_marginal_iteration"]); c=d["compressed_index_plan"]; print("c16", c["kernel_avg_ms"], c["gflops"], c["cg_us_per_iteration"], c["speedup_over_the_headline_kernel"])
PY
