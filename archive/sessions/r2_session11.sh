#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s11; mkdir -p $O
cp cusp-autotuned_amd/tuned/gfx950.json $O/table_before.json
timeout -k 10 1100 python tools/autotune.py --formats csr --merge --log $O/autotune_csr.jsonl > $O/autotune_csr.txt 2>&1; rc=$?; echo "autotune csr exit $rc"; grep -E "^csr/|rows," $O/autotune_csr.txt | cut -c1-260
cp cusp-autotuned_amd/tuned/gfx950.json $O/table_after.json
