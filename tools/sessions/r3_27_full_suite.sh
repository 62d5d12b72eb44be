#!/bin/bash
# r3 session 27: the FULL -m gpu suite on the tree with the device COO sort, the ablation instances and the property-based tests; smoke(); the driver's bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s27; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --hypothesis-show-statistics > $O/pytest_gpu.txt 2>&1; rc=$?
echo "pytest exit $rc"; tail -n 6 $O/pytest_gpu.txt | cut -c1-300; grep -n "passing examples\|failing examples" $O/pytest_gpu.txt | head
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke exit $?"; tail -2 $O/smoke.txt
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; echo "bench exit $?"; cut -c1-600 $O/bench_driver_cmd.json
