#!/bin/bash
# r4 session 7: csr_waver ablations (what do the product stage and the sum phase cost now that the kernel is not stream-bound?), the PV variant (values
# loaded per piece, products parked: no product stage), the run-compressed copy and the wave tiles below their size gates, the -m gpu suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s7; mkdir -p $O
for ab in 0 1 2 3; do
  CMI_WAVER_ABLATE=$ab PMC_WAVEV= PMC_WAVER=4 PMC_PACKED=0 timeout -k 10 300 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/ablate_$ab.txt 2>&1; echo "== CMI_WAVER_ABLATE=$ab"; grep "TIME.*waver4" $O/ablate_$ab.txt | cut -c1-90
done
for pv in 0 1 0 1; do
  CMI_WAVER_PV=$pv PMC_WAVEV= PMC_WAVER=4,2 PMC_WAVER_POL=0,2 timeout -k 10 300 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/pv_$pv.txt 2>&1; echo "== CMI_WAVER_PV=$pv"; grep "TIME.*\(waver\|packed\)" $O/pv_$pv.txt | cut -c1-90
done
CMI_WAVER_PV=1 timeout -k 10 600 python -m pytest tests/test_round4_gpu.py -q -m gpu -x -k "waver or configs3_full_size_run or unaligned" > $O/tests_pv.txt 2>&1; echo "PV tests exit $?"; tail -3 $O/tests_pv.txt | cut -c1-200
for sc in 0.12 0.25 0.5; do
  PMC_SCALE=$sc PMC_WAVEV=4 PMC_WAVER=4,2 PMC_PACKED=0 timeout -k 10 300 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/scale_$sc.txt 2>&1; echo "== scale $sc"; grep -E "^#|TIME" $O/scale_$sc.txt | cut -c1-130
done
for sc in 0.3 3.0; do
  PMC_SCALE=$sc PMC_WAVEV=1,2 PMC_WAVER= timeout -k 10 300 python3 tools/pmc_matrix_probe.py thermal2 --time > $O/thermal2_scale_$sc.txt 2>&1; echo "== thermal2 scale $sc"; grep -E "^#|TIME" $O/thermal2_scale_$sc.txt | cut -c1-130
done
timeout -k 10 1150 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "full pytest exit $?"; tail -n 6 $O/pytest_gpu.txt | cut -c1-250
