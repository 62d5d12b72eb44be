#!/bin/bash
# Round-2 GPU session 4: parity tests; COO re-tune (tile kernel rewritten); HYB rule (two rule kinds); SuiteSparse-like sweep
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 25 $O/pytest_gpu.txt
[ $rc -ge 124 ] && exit $rc
timeout -k 10 600 python tools/autotune.py --formats coo --merge --skip-synthetic --log $O/autotune_coo.jsonl > $O/autotune_coo.txt 2>&1; rc=$?; echo "autotune coo exit $rc"; tail -n 6 $O/autotune_coo.txt
[ $rc -ge 124 ] && exit $rc
timeout -k 10 1200 python tools/autotune_hyb.py --log $O/autotune_hyb.jsonl > $O/autotune_hyb.txt 2>&1; rc=$?; echo "autotune_hyb exit $rc"; tail -n 34 $O/autotune_hyb.txt
[ $rc -ge 124 ] && exit $rc
cp cusp-autotuned_amd/tuned/gfx950.json $O/table_after.json
timeout -k 10 900 python tools/suitesparse_sweep.py > $O/suitesparse_like_sweep.txt 2>&1; rc=$?; echo "sweep exit $rc"; cat $O/suitesparse_like_sweep.txt | tail -n 75
