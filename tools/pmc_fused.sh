#!/bin/bash
# PMC passes on the eager Netlib training step of the fused path (tools/profile_step.py netlib); counters in passes
# of their own (no trace domains).  Summaries: tools/summarize_pmc.py gpurun_out/pmc_fused fused_
mkdir -p gpurun_out/pmc_fused; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_fused/pass$i -- python3 tools/profile_step.py netlib 3 > gpurun_out/pmc_fused/pass$i.log 2>&1
  rc=$?; echo "pass$i rc=$rc"; if [ $rc -ne 0 ]; then tail -5 gpurun_out/pmc_fused/pass$i.log; exit $rc; fi
done
python3 tools/summarize_pmc.py gpurun_out/pmc_fused fused_ > gpurun_out/fused_pmc.txt 2>&1
rm -rf gpurun_out/pmc_fused/pass*/
