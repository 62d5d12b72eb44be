#!/bin/bash
# r3 session 15: buckets 0 and 1 of ELL / DIA / COO again (session 11 tuned bucket 1 on a width-2 matrix whose mean, 1.9999999, is bucket 0's)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s15; mkdir -p $O
cp cusp-autotuned_amd/tuned/gfx950.json $O/gfx950.json
timeout -k 10 600 python3 tools/autotune.py --formats ell,dia,coo --per-bucket --buckets 0,1 --skip-headline --merge --out $O/gfx950.json --log $O/autotune.jsonl > $O/autotune.txt 2>&1; echo "autotune exit $?"
grep -E "^(ell|dia|coo|coo_sorted)/|bucket|wrote" $O/autotune.txt | cut -c1-240
gzip -f $O/autotune.jsonl
timeout -k 10 800 python3 tools/wavev_ab.py > $O/wavev_ab.txt 2>&1; echo "ab exit $?"; grep -v amdgpu.ids $O/wavev_ab.txt | cut -c1-620
