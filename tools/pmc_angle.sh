#!/bin/bash
# MFMA / issue counters of the AngleModel kernels (separate --pmc passes, no trace domains): bash tools/pmc_angle.sh [tag]
tag=${1:-r04_angle}
mkdir -p gpurun_out/$tag; export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/$tag/pass$i -- python3 tools/bench_angle.py 25fv47 256 4 > gpurun_out/$tag/pass$i.log 2>&1
  rc=$?; echo "pass$i rc=$rc"; if [ $rc -ne 0 ]; then tail -5 gpurun_out/$tag/pass$i.log; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
for k in attn_kernel gemm_vec_kernel; do python3 tools/summarize_pmc.py gpurun_out/$tag $k; done > gpurun_out/${tag}_pmc.txt
rm -rf gpurun_out/$tag/pass*/
cat gpurun_out/${tag}_pmc.txt
