#!/bin/bash
# round 4 session 21: the whole -m gpu suite and the driver's bench command after the rule changes of the regret run
# (f64 stencil rows of 5..8 -> wave tiles V = 1; stencil rows with runs -> the copy; short irregular rows V = 2; partial capacity 2^17)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s21; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -15 $O/pytest_gpu.txt | cut -c1-250
[ $rc -ne 0 ] && exit $rc
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err || { echo bench failed; tail -5 $O/bench_driver_cmd.err; exit 2; }
python3 -c "
import json; d=json.load(open('$O/bench_driver_cmd.json'))
print({k: d[k] for k in ('value','ms_per_step')}, d['roofline'], d.get('roofline_cold',{}).get('frac'), d.get('cg'))
print(d['config'].get('kernel_config'))
"
