#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s13; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_csr16_gpu.py tests/test_plan_gpu.py tests/test_cpp_layer.py -m gpu -x -q > $O/pytest_new.txt 2>&1; rc=$?; echo "pytest(new) exit $rc"; tail -n 25 $O/pytest_new.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_csr.json 2>$O/bench_csr.err || { echo "bench failed"; tail -n 5 $O/bench_csr.err; }
python - <<PY
import json
d=json.loads(open("$O/bench_csr.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("csr", d["ms_per_step"], r["frac"], r["kernel_avg_ms"]); print("cg", d.get("cg")); print("c16", json.dumps(d.get("compressed_index_plan")))
PY
timeout -k 10 300 python tools/ell_wide_probe.py f64 > $O/ell_wide_f64.txt 2>&1; echo "ell probe exit $?"; grep -v amdgpu.ids $O/ell_wide_f64.txt
timeout -k 10 900 python tools/suitesparse_sweep.py --shapes > $O/suitesparse_like_shapes.txt 2>&1; rc=$?; echo "sweep exit $rc"; grep -E "^==|table|plan|block|\*" $O/suitesparse_like_shapes.txt | cut -c1-170
