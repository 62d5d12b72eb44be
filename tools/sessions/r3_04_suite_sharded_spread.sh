#!/bin/bash
# r3 session 4: full -m gpu suite on the new code (fenced folds, plan checks, csr_wavev, spread row sums, sharded C++ layer with a
# 1-rank RCCL communicator), then: cg_bench fenced vs relaxed folds, cg_bench --sharded, spread row sums A/B on the long-row matrices,
# and what this box's HBM gives a plain read kernel (tools/membench).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s04; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 12 $O/pytest_gpu.txt
[ $rc -ge 124 ] && exit $rc
timeout -k 10 200 tools/bin/cg_bench --iterations=200 > $O/cg_fenced.txt 2>&1; echo "cg fenced exit $?"; grep -E "^fused" $O/cg_fenced.txt
CMI_FOLD_RELAXED=1 timeout -k 10 200 tools/bin/cg_bench --iterations=200 > $O/cg_relaxed.txt 2>&1; echo "cg relaxed exit $?"; grep -E "^fused" $O/cg_relaxed.txt
timeout -k 10 200 tools/bin/cmi_launch -n 1 -- tools/bin/cg_bench --sharded --grid=3162 --iterations=200 > $O/cg_sharded_1rank.txt 2>&1; echo "sharded exit $?"; cat $O/cg_sharded_1rank.txt
for sp in 0 1; do
  CMI_CSR_SPREAD=$sp PMC_WAVEV= timeout -k 10 300 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120,thermal2 --time > $O/time_spread$sp.txt 2>&1 || { tail -5 $O/time_spread$sp.txt; exit 1; }
  echo "spread=$sp"; grep -E "^TIME" $O/time_spread$sp.txt | cut -c1-100
done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o /tmp/membench 2> $O/membench_build.err && timeout -k 10 120 /tmp/membench > $O/membench.txt 2>&1; echo "membench exit $?"; cat $O/membench.txt | head -40
