#!/bin/bash
# PMC passes on the tiled SpMM (tools/bench_spmm.py): LDS / VALU / wait breakdown
mkdir -p gpurun_out/pmc_tiled; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_tiled/pass$i -- python3 tools/bench_spmm.py ${1:-64} 3 > gpurun_out/pmc_tiled/pass$i.log 2>&1
  rc=$?; echo "pass$i rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
