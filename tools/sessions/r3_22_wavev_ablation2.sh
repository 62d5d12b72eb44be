#!/bin/bash
# r3 session 22: session 21 said the LDS stage costs csr_wavev 25 % although its sums cost ~1 %.  Is it the ALLOCATION (32 KiB per workgroup, given
# back when the LAST of four waves ends)?  ablate 10 / 11 = no LDS traffic but the allocation kept (11: and no gathers); CMI_WAVEV_WPB = 1 / 2:
# the real kernel with one / two wave tiles per workgroup
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s22; mkdir -p $O
run() { # label, env...
  local label=$1; shift
  env "$@" PMC_WAVEV=4 PMC_WAVEV_POL=3 timeout -k 10 400 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/$label.txt 2> $O/$label.err || { echo "$label failed"; tail -3 $O/$label.err; return 2; }
  echo "== $label ($*)"; grep "^TIME" $O/$label.txt | cut -f2-6
}
{ run base CMI_WAVEV_ABLATE=0 && run abl2 CMI_WAVEV_ABLATE=2 && run abl10 CMI_WAVEV_ABLATE=10 && run abl11 CMI_WAVEV_ABLATE=11 && run wpb1 CMI_WAVEV_WPB=1 && run wpb2 CMI_WAVEV_WPB=2 && run base2 CMI_WAVEV_ABLATE=0; } > $O/ablation2.txt 2>&1
cat $O/ablation2.txt
