#!/bin/bash
# r3 session 11: ELL / DIA / COO re-tuned PER BUCKET -- the table is written under gpurun_out/ and copied into the tree afterwards
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s11; mkdir -p $O
cp cusp-autotuned_amd/tuned/gfx950.json $O/gfx950.json
timeout -k 10 1000 python3 tools/autotune.py --formats ell,dia,coo --per-bucket --merge --out $O/gfx950.json --log $O/autotune.jsonl > $O/autotune.txt 2>&1; echo "autotune exit $?"
grep -E "^(ell|dia|coo|coo_sorted)/|bucket|wrote" $O/autotune.txt | cut -c1-260
gzip -f $O/autotune.jsonl
