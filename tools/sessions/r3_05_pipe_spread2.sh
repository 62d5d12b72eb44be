#!/bin/bash
# r3 session 5: long-row matrices -- the persistent pipelined csr_stream_pipe (round 1's kernel, never tried on them) and the
# contiguous-chunk dealing of the row sums (CMI_CSR_SPREAD=2), beside csr_stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s05; mkdir -p $O
PMC_WAVEV= PMC_PIPE=256:20,512:40,1024:80,1024:64 timeout -k 10 300 python3 tools/pmc_matrix_probe.py ldoor --time > $O/time_pipe_ldoor.txt 2>&1; grep -E "^TIME" $O/time_pipe_ldoor.txt | cut -c1-100
PMC_WAVEV= PMC_PIPE=256:34,512:70,1024:140,1024:128 timeout -k 10 300 python3 tools/pmc_matrix_probe.py nlpkkt120 --time > $O/time_pipe_nlpkkt.txt 2>&1; grep -E "^TIME" $O/time_pipe_nlpkkt.txt | cut -c1-100
for sp in 0 2; do
  CMI_CSR_SPREAD=$sp PMC_WAVEV= timeout -k 10 300 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/time_spread$sp.txt 2>&1
  echo "spread=$sp"; grep -E "^TIME" $O/time_spread$sp.txt | cut -c1-100
done
