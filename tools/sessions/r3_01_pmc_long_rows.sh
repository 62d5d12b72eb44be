#!/bin/bash
# r3 session 1: counters that did not exist -- ldoor-like / nlpkkt120-like at full size, the plan's kernel
# and the table's, one rocprofv3 --pmc pass per counter set (program directly after `--`), then HIP-event timing of the same.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s01; mkdir -p $O
M=${1:-ldoor,nlpkkt120}
timeout -k 10 400 python3 tools/pmc_matrix_probe.py $M --time > $O/time.txt 2>&1 || { tail -5 $O/time.txt; exit 1; }
grep -E "^TIME|^#" $O/time.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/p$i -- python3 tools/pmc_matrix_probe.py $M > $O/manifest_$i.txt 2> $O/pmc_$i.err
  rc=$?; echo "pass $i ($set) exit $rc"
  [ $rc -ge 124 ] && exit $rc
done
python3 tools/pmc_matrix_table.py $O/manifest_1.txt $O/pmc $O/pmc_table.json > $O/pmc_table.txt 2> $O/pmc_table.err; tail -3 $O/pmc_table.err
cat $O/pmc_table.txt | head -120
