#!/bin/bash
# r3 session 2: the wave-private 16-byte-vector kernel (CMI_CSR_STREAM_WAVEV): parity tests, then timing + counters beside csr_stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s02; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_round3_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest exit $rc"; tail -n 15 $O/pytest.txt
[ $rc -eq 0 ] || exit 1
M=${1:-ldoor,nlpkkt120,thermal2}
PMC_WAVEV=1,2,4 timeout -k 10 400 python3 tools/pmc_matrix_probe.py $M --time > $O/time.txt 2>&1 || { tail -5 $O/time.txt; exit 1; }
grep -E "^TIME|^#" $O/time.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  PMC_WAVEV=2,4 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/p$i -- python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 > $O/manifest_$i.txt 2> $O/pmc_$i.err
  rc=$?; echo "pass $i ($set) exit $rc"
  [ $rc -ge 124 ] && exit $rc
done
python3 tools/pmc_matrix_table.py $O/manifest_1.txt $O/pmc $O/pmc_table.json > $O/pmc_table.txt 2> $O/pmc_table.err; tail -3 $O/pmc_table.err
grep -E "^[a-z]|traffic_over|wait_any|lds_conflict|SQ_WAVE_CYCLES|GRBM" $O/pmc_table.txt
