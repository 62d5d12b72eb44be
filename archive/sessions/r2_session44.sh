#!/bin/bash
# session 44: which load shape over CSR's arrays does the memory system serve fastest (tools/r2_probe.hip `shape`), next to the
# library's CSR and ELL kernels on the same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s44; mkdir -p $O
timeout -k 10 240 tools/bin/r2_probe --only "lib csr table|lib ell table|mix x-thrice|shape " > $O/shape.txt 2>&1; rc=$?; echo "probe exit $rc"; cat $O/shape.txt
