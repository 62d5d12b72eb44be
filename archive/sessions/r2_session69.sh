#!/bin/bash
# session 69: the per-format bench lines on the final code
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s69; mkdir -p $O
for f in ell dia coo hyb; do
  python bench.py --format $f --no-cpu-baseline --steps 200 > $O/bench_n1_$f.json 2> $O/bench_$f.err || { echo "bench $f failed"; tail -3 $O/bench_$f.err; }
done
python - <<PY
import json
for f in ("ell","dia","coo","hyb"):
    e=json.loads(open("$O/bench_n1_%s.json"%f).read().strip().splitlines()[-1]); r=e["roofline"]
    print(f, e["value"], e["ms_per_step"], r["frac"], r["kernel_avg_ms"], r["traffic"])
PY
