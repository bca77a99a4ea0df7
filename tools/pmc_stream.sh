#!/bin/bash
# PMC passes on the streamed SpMM (tools/profile_stream.py <instances> 3): SQ / LDS counters, then the vector-memory path
mkdir -p gpurun_out/pmc_stream; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_stream/pass$i -- python3 tools/profile_stream.py ${1:-64} 3 > gpurun_out/pmc_stream/pass$i.log 2>&1
  rc=$?; echo "pass$i rc=$rc"; if [ $rc -ne 0 ]; then tail -5 gpurun_out/pmc_stream/pass$i.log; exit $rc; fi   # no further GPU step after a failed one
done
python3 tools/summarize_pmc.py gpurun_out/pmc_stream spmm_stream
