#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s12; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_plan_gpu.py tests/test_spmv_gpu.py -m gpu -x -q -k "hyb or ell or golden or random or plan" > $O/pytest_new.txt 2>&1; rc=$?; echo "pytest(new) exit $rc"; tail -n 15 $O/pytest_new.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/ell_wide_probe.py f64 > $O/ell_wide_f64.txt 2>&1; echo "ell probe exit $?"; cat $O/ell_wide_f64.txt
for f in hyb coo ell; do
  timeout -k 10 200 python bench.py --format $f --steps 100 --no-cpu-baseline --cg-iterations 0 > $O/bench_$f.json 2>$O/bench_$f.err || { echo "bench $f failed"; tail -n 5 $O/bench_$f.err; }
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$f", d["ms_per_step"], r["frac"], r.get("kernel_avg_ms"), r.get("kernel"))
except Exception as e: print("$f parse", e)
PY
done
timeout -k 10 600 python tools/suitesparse_sweep.py > $O/suitesparse_like_sweep.txt 2>&1; rc=$?; echo "sweep exit $rc"; grep -E "^==|table|plan|\*" $O/suitesparse_like_sweep.txt | cut -c1-150
