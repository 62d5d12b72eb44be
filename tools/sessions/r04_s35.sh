#!/bin/bash
# round 4 session 35: the size gate of the wave tiles on gather-bound band matrices (regret set 3)
set -o pipefail
mkdir -p gpurun_out/r4s35
cd /root/repo
timeout -k 10 1000 python3 tools/auto_regret.py --set 3 --log gpurun_out/r4s35/auto_regret_set3.jsonl > gpurun_out/r4s35/auto_regret_set3.txt 2>&1; echo "regret exit $?"
python3 - <<'PY'
import json
for l in open("gpurun_out/r4s35/auto_regret_set3.jsonl"):
    r = json.loads(l); c = r["candidates"]
    g = lambda k: c[k][0] if k in c and c[k][0] else float("nan")
    print(f"{r['matrix'][:62]:62s} {r['dtype']}  AUTO k{r['auto_kernel']} {r['auto_us']:6.1f}  table {g('csr_stream (table)'):6.1f}  V1 {g('wave tiles V=1'):6.1f}  V2 {g('wave tiles V=2'):6.1f}  V4 {g('wave tiles V=4'):6.1f}  V4+window {g('wave tiles V=4 + x window'):6.1f}")
PY
