#!/bin/bash
# round 4 session 17: the offline autotune of csr_waver's rule (tools/autotune_waver.py) + the test that the table's rule steers AUTO plans
set -o pipefail
mkdir -p gpurun_out/r4s17
cd /root/repo
timeout -k 10 700 python3 tools/autotune_waver.py --log gpurun_out/r4s17/autotune_waver.jsonl > gpurun_out/r4s17/autotune_waver.txt 2>&1; echo "autotune exit $?"
tail -30 gpurun_out/r4s17/autotune_waver.txt
timeout -k 10 300 python3 -m pytest tests/test_round4_gpu.py -m gpu -x -q -k "steers or refusals or fall_back" > gpurun_out/r4s17/pytest.txt 2>&1; echo "pytest exit $?"
tail -5 gpurun_out/r4s17/pytest.txt
