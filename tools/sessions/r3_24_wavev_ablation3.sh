#!/bin/bash
# r3 session 24: the ablation table again with bit 2 FIXED (sessions 21-23: the compiler had masked the loads of lanes >= rows off in the "no LDS" instances:
# they read a third of the matrix -- their 0.96-1.2 "of peak" was not a ceiling).  base, 1 (no gathers), 2 (no LDS, no sums), 3 (= 1 + 2: the bare streams in this
# launch structure), 4 (LDS writes, one read per row), 5, 10, 11
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s24; mkdir -p $O
run() { # label, env...
  local label=$1; shift
  env "$@" PMC_WAVEV=4 PMC_WAVEV_POL=3 timeout -k 10 300 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/$label.txt 2> $O/$label.err || { echo "$label failed"; tail -3 $O/$label.err; return 2; }
  echo "== $label ($*)"; grep "^TIME.*wavev" $O/$label.txt | cut -f2-6
}
{ run base CMI_WAVEV_ABLATE=0 && run abl1 CMI_WAVEV_ABLATE=1 && run abl2 CMI_WAVEV_ABLATE=2 && run abl3 CMI_WAVEV_ABLATE=3 && run abl4 CMI_WAVEV_ABLATE=4 && run abl5 CMI_WAVEV_ABLATE=5 && run abl11 CMI_WAVEV_ABLATE=11 && run base2 CMI_WAVEV_ABLATE=0; } > $O/ablation3.txt 2>&1
cat $O/ablation3.txt
