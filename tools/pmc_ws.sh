#!/bin/bash
# PMC passes (SQ / LDS counters only) on the tiled SpMM: tools/bench_spmm.py <instances> 3
mkdir -p gpurun_out/pmc_ws; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_ws/pass$i -- python3 tools/bench_spmm.py ${1:-64} 3 > gpurun_out/pmc_ws/pass$i.log 2>&1
  rc=$?; echo "pass$i rc=$rc"; if [ $rc -ne 0 ]; then tail -5 gpurun_out/pmc_ws/pass$i.log; exit $rc; fi
done
python3 tools/summarize_pmc.py gpurun_out/pmc_ws spmm_tiled
