#!/bin/bash
# usage: bash tools/pmc_attn.sh [instances] [first pass]
# PMC passes (separate --pmc runs, no trace domains) over the three 16-channel attention sweeps alone, 256 instances
mkdir -p gpurun_out/pmc_attn; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD TA_BUSY_avr" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" ; do
  i=$((i+1))
  if [ $i -lt ${2:-1} ]; then continue; fi      # (second argument: first pass to run; FETCH_SIZE does not combine with TCC_* in one pass)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_attn/pass$i -- python3 tools/profile_attn.py ${1:-256} 3 > gpurun_out/pmc_attn/pass$i.log 2>&1
  rc=$?; echo "pass$i rc=$rc"; if [ $rc -ne 0 ]; then tail -5 gpurun_out/pmc_attn/pass$i.log; exit $rc; fi   # no further GPU step after a failed one
done
for k in fwd16_tiled_kernel bwdsrc16_tiled_kernel bwddst16_lane_kernel; do python3 tools/summarize_pmc.py gpurun_out/pmc_attn $k; done > gpurun_out/pmc_attn_summary.txt
rm -rf gpurun_out/pmc_attn/pass*/
cat gpurun_out/pmc_attn_summary.txt
