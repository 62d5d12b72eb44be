#!/bin/bash
# round 4 session 33: equal row lengths whose columns are not a stencil's -- the regret rows again after the rule, its test, the whole suite
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s33; mkdir -p $O
timeout -k 10 600 python3 tools/auto_regret.py --only "per row exactly" --log $O/auto_regret_equal_lengths.jsonl > $O/auto_regret_equal_lengths.txt 2>&1; echo "regret exit $?"
grep -A12 "== regret" $O/auto_regret_equal_lengths.txt | cut -c1-230
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -8 $O/pytest_gpu.txt | cut -c1-250
