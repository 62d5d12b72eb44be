#!/bin/bash
# round 4 session 18: tools/autotune_waver.py with re-timed finalists and the denser size gate; the patched table comes back under gpurun_out/
set -o pipefail
mkdir -p gpurun_out/r4s18
cd /root/repo
cp cusp-autotuned_amd/tuned/gfx950.json gpurun_out/r4s18/gfx950.json
timeout -k 10 900 python3 tools/autotune_waver.py --log gpurun_out/r4s18/autotune_waver.jsonl --patch gpurun_out/r4s18/gfx950.json > gpurun_out/r4s18/autotune_waver.txt 2>&1; echo "autotune exit $?"
grep -v "refused" gpurun_out/r4s18/autotune_waver.txt | tail -60
