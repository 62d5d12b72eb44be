#!/bin/bash
# r3 session 21: what each stage of csr_wavev costs on the long-row matrices -- ablated instances ($CMI_WAVEV_ABLATE: 1 no gathers, 2 no LDS / sums,
# 3 both = the bare streams in this launch structure, 4 LDS writes but one read per row, 5 = 4 without gathers); results are wrong by design
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s21; mkdir -p $O
for abl in 0 3 1 2 4 5 0; do
  CMI_WAVEV_ABLATE=$abl PMC_WAVEV=4 PMC_WAVEV_POL=3 timeout -k 10 400 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/abl_$abl.txt 2> $O/abl_$abl.err || { echo "ablate $abl failed"; tail -3 $O/abl_$abl.err; exit 2; }
  echo "== CMI_WAVEV_ABLATE=$abl"; grep "^TIME" $O/abl_$abl.txt | cut -f2-6
done > $O/ablation.txt 2>&1
cat $O/ablation.txt
