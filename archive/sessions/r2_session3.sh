#!/bin/bash
# Round-2 GPU session 3: parity tests (plans, COO tile kernel, HYB rule), then the offline autotune of ELL / DIA / COO
# (XCD dealing added to their spaces; sorted-COO key) and of the HYB width rule.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s3; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 25 $O/pytest_gpu.txt
[ $rc -ge 124 ] && exit $rc
cp cusp-autotuned_amd/tuned/gfx950.json $O/table_before.json
timeout -k 10 900 python tools/autotune.py --formats ell,dia,coo --merge --skip-synthetic --log $O/autotune_ell_dia_coo.jsonl > $O/autotune.txt 2>&1; rc=$?; echo "autotune exit $rc"; tail -n 14 $O/autotune.txt
[ $rc -ge 124 ] && exit $rc
timeout -k 10 1200 python tools/autotune_hyb.py --log $O/autotune_hyb.jsonl > $O/autotune_hyb.txt 2>&1; rc=$?; echo "autotune_hyb exit $rc"; tail -n 30 $O/autotune_hyb.txt
cp cusp-autotuned_amd/tuned/gfx950.json $O/table_after.json
