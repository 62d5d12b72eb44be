#!/bin/bash
# r3 session 23: session 22: the 32 KiB LDS ALLOCATION of a csr_wavev workgroup alone (no LDS traffic, no sums) costs what the whole LDS stage costs.
# How small must it be?  ablate 10 with the kept allocation at 1/1, 1/2, 1/4, 1/8, 1/16 of the real one; only ldoor (one process each)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s23; mkdir -p $O
run() { # label, env...
  local label=$1; shift
  env "$@" PMC_WAVEV=4 PMC_WAVEV_POL=3 timeout -k 10 300 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/$label.txt 2> $O/$label.err || { echo "$label failed"; tail -3 $O/$label.err; return 2; }
  echo "== $label ($*)"; grep "^TIME.*wavev" $O/$label.txt | cut -f2-6
}
{ run ldiv1 CMI_WAVEV_ABLATE=10 CMI_WAVEV_LDIV=1 && run ldiv2 CMI_WAVEV_ABLATE=10 CMI_WAVEV_LDIV=2 && run ldiv4 CMI_WAVEV_ABLATE=10 CMI_WAVEV_LDIV=4 && run ldiv8 CMI_WAVEV_ABLATE=10 CMI_WAVEV_LDIV=8 && run ldiv16 CMI_WAVEV_ABLATE=10 CMI_WAVEV_LDIV=16 && run none CMI_WAVEV_ABLATE=2; } > $O/lds_size.txt 2>&1
cat $O/lds_size.txt
