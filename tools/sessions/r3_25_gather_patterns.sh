#!/bin/bash
# r3 session 25: session 24 (corrected ablation): the x gathers are the whole gap of csr_wavev on the long-row matrices (0.69 -> 0.82 without them; the LDS stage
# costs nothing).  WHAT about them -- the number of cache lines one gather instruction touches, or their latency?  The same row structure with synthetic columns:
# row (one line per row), seq (entry e -> x[e]: fewest lines per instruction), win512 / win4096 (random inside a cache-resident window: many lines per instruction,
# all hits), real.  csr_wavev V = 4 and the table's csr_stream, ldoor-like only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s25; mkdir -p $O
run() { # label, env...
  local label=$1; shift
  env "$@" PMC_WAVEV=4 PMC_WAVEV_POL=3 timeout -k 10 300 python3 tools/pmc_matrix_probe.py ldoor --time > $O/$label.txt 2> $O/$label.err || { echo "$label failed"; tail -3 $O/$label.err; return 2; }
  echo "== $label ($*)"; grep "^TIME" $O/$label.txt | cut -f2-6
}
{ run real PMC_COLS= && run zero PMC_COLS=zero && run row PMC_COLS=row && run seq PMC_COLS=seq && run win64 PMC_COLS=win64 && run win512 PMC_COLS=win512 && run win4096 PMC_COLS=win4096 && run nogather CMI_WAVEV_ABLATE=1; } > $O/gather_patterns.txt 2>&1
cat $O/gather_patterns.txt
