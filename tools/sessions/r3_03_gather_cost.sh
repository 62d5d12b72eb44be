#!/bin/bash
# r3 session 3: is the 25-80 entries/row class bound by the x GATHERS (L1 / texture addresser), not by HBM?  Same matrices with the
# column indices replaced (PMC_COLS=zero / row), and the L1-side counters of the real ones.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s03; mkdir -p $O
for mode in "" zero row; do
  PMC_COLS=$mode PMC_WAVEV=2,4 timeout -k 10 300 python3 tools/pmc_matrix_probe.py ldoor,nlpkkt120 --time > $O/time_$mode.txt 2>&1 || { tail -5 $O/time_$mode.txt; exit 1; }
  grep -E "^TIME" $O/time_$mode.txt | cut -c1-90
done
rocprofv3 -L > $O/counters_available.txt 2>&1
grep -ciE "TCP_|TA_|TD_" $O/counters_available.txt
i=0
for set in "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum" "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TAGCONFLICT_STALL_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM"; do
  i=$((i+1))
  PMC_WAVEV=4 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/p$i -- python3 tools/pmc_matrix_probe.py ldoor > $O/manifest_$i.txt 2> $O/pmc_$i.err
  rc=$?; echo "pass $i ($set) exit $rc"; [ $rc -ne 0 ] && tail -3 $O/pmc_$i.err
  [ $rc -ge 124 ] && exit $rc
done
python3 tools/pmc_matrix_table.py $O/manifest_1.txt $O/pmc $O/pmc_table.json > $O/pmc_table.txt 2> $O/pmc_table.err; tail -3 $O/pmc_table.err
grep -vE "config|kernel  |algorithmic|bit_exact" $O/pmc_table.txt
