#!/bin/bash
# PMC passes (each its own rocprofv3 run, counters only -- no trace domains) on tools/profile_spmm.py
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
i=0
for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc/pass$i -- python3 tools/profile_spmm.py ${1:-32} 3 ${2:-spmm} > gpurun_out/pmc/pass$i.log 2>&1
  rc=$?; echo "pass$i ($set) rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
