#!/bin/bash
# Round-2 GPU session 5: probe (fused-dot variants, merge-path kernel), counters stream vs balanced, COO re-tune, full tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/s5; mkdir -p $O
P=tools/bin/r2_probe
timeout -k 10 180 $P > $O/probe_timing.txt 2>&1 || { echo probe failed; tail -5 $O/probe_timing.txt; exit 1; }
cat $O/probe_timing.txt
i=0
SEL="lib csr table|lib csr_balanced swz 0|lib csr_balanced accumulate"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_EA0_ATOMIC_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/p$i -o p -- $P --pmc 5 --only "$SEL" > $O/manifest_$i.txt 2> $O/pmc_$i.err || { echo "pmc pass $i failed"; tail -3 $O/pmc_$i.err; }
done
python3 tools/r2_pmc_table.py $O/manifest_1.txt $O/pmc $O/balanced_pmc.json > $O/balanced_pmc.txt 2> $O/balanced_pmc.err; cat $O/balanced_pmc.err | head -5
find $O/pmc -name "*kernel_trace.csv" -delete
timeout -k 10 600 python tools/autotune.py --formats coo --merge --skip-synthetic --log $O/autotune_coo.jsonl > $O/autotune_coo.txt 2>&1; rc=$?; echo "autotune coo exit $rc"; tail -n 6 $O/autotune_coo.txt
cp cusp-autotuned_amd/tuned/gfx950.json $O/table_after.json
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 25 $O/pytest_gpu.txt
