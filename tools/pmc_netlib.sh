#!/bin/bash
# PMC passes on the eager Netlib training step (tools/profile_step.py netlib)
mkdir -p gpurun_out/pmc_netlib; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE SQ_WAVES" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_netlib/pass$i -- python3 tools/profile_step.py netlib 3 > gpurun_out/pmc_netlib/pass$i.log 2>&1
  rc=$?; echo "pass$i rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
