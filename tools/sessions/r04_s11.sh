#!/bin/bash
# r4 session 11: f32 size gate of csr_waver (stand-ins at 0.12 / 0.25 of their size), what the round-4 plans cost to make, the -m gpu suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s11; mkdir -p $O
for sc in 0.12 0.25; do
  PMC_DTYPE=f32 PMC_SCALE=$sc PMC_WAVEV= PMC_WAVER=4,2 PMC_PACKED=0 timeout -k 10 300 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/f32_scale_$sc.txt 2>&1; echo "== f32 scale $sc"; grep -E "^#|TIME" $O/f32_scale_$sc.txt | cut -c1-130
done
timeout -k 10 400 python3 tools/plan_cost_probe.py > $O/plan_cost.txt 2>&1; cat $O/plan_cost.txt | cut -c1-200
timeout -k 10 1150 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "full pytest exit $?"; tail -n 6 $O/pytest_gpu.txt | cut -c1-250
