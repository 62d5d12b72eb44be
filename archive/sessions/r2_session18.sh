#!/bin/bash
# HYB width rule re-fit (one launch for light COO parts, two for heavy ones), then the full GPU suite with that table, HYB / COO bench lines, HYB PMC
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s18; mkdir -p $O
cp cusp-autotuned_amd/tuned/gfx950.json $O/gfx950.json
timeout -k 10 1100 python tools/autotune_hyb.py --out $O/gfx950.json --log $O/autotune_hyb.jsonl > $O/autotune_hyb.txt 2>&1; echo "autotune hyb exit $?"; grep -v amdgpu.ids $O/autotune_hyb.txt | grep "tuned rule\|rule K" | cut -c1-260
cp $O/gfx950.json cusp-autotuned_amd/tuned/gfx950.json
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 6 $O/pytest_gpu.txt
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
for f in hyb coo; do python bench.py --format $f --no-cpu-baseline --steps 200 > $O/bench_n1_$f.json 2>/dev/null || echo "bench $f failed"; done
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/hybpmc -o $pass -- python3 tools/pmc_probe.py hyb > $O/hyb_probe_$pass.json 2> $O/hybpmc_$pass.err || { echo hyb pmc $pass failed; }
done
find $O/hybpmc -name "*kernel_trace.csv" -delete
python tools/pmc_summary.py $O/hybpmc $O/hyb_probe_FETCH_SIZE.json $O/hyb_pmc.json > $O/hyb_pmc.txt 2>&1
python - <<PY
import json
for f in ("hyb","coo"):
    e=json.loads(open("$O/bench_n1_%s.json"%f).read().strip().splitlines()[-1]); r=e["roofline"]; print(f, e["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["kernel_avg_over_ms_per_step"], r["traffic"])
PY
