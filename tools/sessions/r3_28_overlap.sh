#!/bin/bash
# r3 session 28: the C++ sharded layer with the interior rows overlapped with the two-sided halo exchange: 1 rank through RCCL, 2 / 3 ranks
# sharing the GPU (staged collectives); then cg_bench --sharded with 2 ranks sharing the GPU, halo mode forced two-sided, overlap on / off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s28; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_cpp_layer.py -m gpu -x -q -k "sharded" > $O/pytest_sharded.txt 2>&1; rc=$?
echo "sharded pytest exit $rc"; tail -n 8 $O/pytest_sharded.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
for ov in 1 0; do
  CMI_COMM_STAGED=1 CMI_EXCHANGE_PEER=0 CMI_EXCHANGE_OVERLAP=$ov HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 tools/bin/cmi_launch -n 2 --port 29701 -- tools/bin/cg_bench --sharded --grid=2000 --iterations=50 > $O/cg_sharded_2ranks_overlap$ov.txt 2>&1
  echo "== overlap $ov: exit $?"; cat $O/cg_sharded_2ranks_overlap$ov.txt | cut -c1-300
done
