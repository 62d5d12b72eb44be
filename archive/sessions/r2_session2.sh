#!/bin/bash
# Round-2 GPU session 2: the library with register row pointers (CSR fast path) and XCD chunk dealing in ELL / DIA:
# probe timings, then the GPU parity tests.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s2; mkdir -p $O
timeout -k 10 180 tools/bin/r2_probe > $O/probe_timing.txt 2>&1 || { echo probe failed; tail -5 $O/probe_timing.txt; exit 1; }
grep -v "^lib ell block\|^lib dia block" $O/probe_timing.txt | tail -n 40
sort -k6 -n <(grep "^lib ell block" $O/probe_timing.txt) | head -5
sort -k6 -n <(grep "^lib dia block" $O/probe_timing.txt) | head -5
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?"; tail -n 8 $O/pytest_gpu.txt
