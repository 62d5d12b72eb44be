#!/bin/bash
# session 56: csr_wave in the library -- the whole GPU suite, then the driver's bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s56; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 12 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/s56/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, {k:d["roofline"][k] for k in ("frac","kernel_avg_ms","kernel_avg_over_ms_per_step")}, d.get("cg"), {k:d["compressed_index_plan"].get(k) for k in ("kernel_config","kernel_avg_ms","speedup_over_the_headline_kernel","cg_us_per_iteration")}, d["config"])
PY
