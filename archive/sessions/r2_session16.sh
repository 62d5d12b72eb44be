#!/bin/bash
# re-fit the HYB width rule now that a HYB multiply is one launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s16; mkdir -p $O
cp cusp-autotuned_amd/tuned/gfx950.json $O/gfx950.json
timeout -k 10 1100 python tools/autotune_hyb.py --out $O/gfx950.json --log $O/autotune_hyb.jsonl > $O/autotune_hyb.txt 2>&1; echo "autotune hyb exit $?"; grep -v amdgpu.ids $O/autotune_hyb.txt | tail -n 40 | cut -c1-260
python bench.py --format hyb --no-cpu-baseline --steps 200 > $O/bench_n1_hyb.json 2>/dev/null || echo "bench hyb failed"
python - <<PY
import json
e=json.loads(open("$O/bench_n1_hyb.json").read().strip().splitlines()[-1]); r=e["roofline"]; print("hyb", e["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["kernel_avg_over_ms_per_step"])
PY
