#!/bin/bash
# HYB width rule re-fit with the final kernels (one launch for light COO parts, ELL + CSR-on-offsets for heavy ones), then the full suite with that table + HYB lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s31; mkdir -p $O
cp cusp-autotuned_amd/tuned/gfx950.json $O/gfx950.json
timeout -k 10 1100 python tools/autotune_hyb.py --out $O/gfx950.json --log $O/autotune_hyb.jsonl > $O/autotune_hyb.txt 2>&1; echo "autotune hyb exit $?"; grep -v amdgpu.ids $O/autotune_hyb.txt | grep "tuned rule\|rule K\|best K" | cut -c1-230
cp $O/gfx950.json cusp-autotuned_amd/tuned/gfx950.json
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 6 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
python bench.py --format hyb --no-cpu-baseline --steps 200 > $O/bench_n1_hyb.json 2>/dev/null || echo "bench hyb failed"
python - <<PY
import json
e=json.loads(open("$O/bench_n1_hyb.json").read().strip().splitlines()[-1]); r=e["roofline"]; print("hyb", e["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["traffic"])
PY
for f in coo hyb; do tools/bin/cg_bench --iterations=200 --format=$f 2>&1 | grep fused | tail -1; done
