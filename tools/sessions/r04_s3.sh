#!/bin/bash
# r4 session 3: counters for the run-compressed kernel on ldoor-like / nlpkkt120-like (plan's choice = waver, table csr_stream, csr_wavev, waver, packed),
# the timing again with a settle phase per variant, the DOT instance's ablations on the headline matrix, plan-less wave tiles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s3; mkdir -p $O
PMC_WAVEV=4 PMC_WAVEV_POL=3 PMC_WAVER=4 timeout -k 10 400 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/long_rows_time.txt 2>&1; grep TIME $O/long_rows_time.txt | cut -c1-100
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  PMC_WAVEV=4 PMC_WAVEV_POL=3 PMC_WAVER=4 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/longpmc/p$i -- python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 > $O/long_manifest_$i.txt 2> $O/longpmc_$i.err
  rc=$?; echo "long-row pass $i ($set) exit $rc"; [ $rc -ge 124 ] && exit $rc
done
python3 tools/pmc_matrix_table.py $O/long_manifest_1.txt $O/longpmc $O/long_rows_pmc.json > $O/long_rows_pmc.txt 2>&1
find $O/longpmc -name "*kernel_trace.csv" -delete; find $O/longpmc -name "*counter_collection.csv" -delete
grep -E "^[a-z]|traffic_over|wait_any|lds_conflict|l2_hit" $O/long_rows_pmc.txt | cut -c1-200
# the fused <y, w> instance of the headline kernel: what do the w load and the workgroup combine cost?
for ab in 0 1 2 3; do
  CMI_DOT_ABLATE=$ab timeout -k 10 200 tools/bin/cg_bench --iterations=200 > $O/cg_dot_ablate_$ab.txt 2>&1; echo "dot ablate $ab:"; grep -E "fused|per iteration" $O/cg_dot_ablate_$ab.txt | head -3
done
timeout -k 10 300 python3 tools/planless_wave_probe.py > $O/planless_wave.txt 2>&1; cat $O/planless_wave.txt | tail -30
