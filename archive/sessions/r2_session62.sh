#!/bin/bash
# session 62: the 16-bit copy tiled per wave (csr_wave16): the whole GPU suite, then the driver's bench command (compressed_index_plan leg)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s62; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 12 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { echo bench failed; tail -5 $O/bench_driver_cmd.err; exit 2; }
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("csr", d["value"], d["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["kernel_avg_over_ms_per_step"]); print("cg", d["cg"]["us_per_iteration"], d["cg"]["us_per_marginal_iteration"]); c=d["compressed_index_plan"]; print("c16", json.dumps(c)[:900])
PY
