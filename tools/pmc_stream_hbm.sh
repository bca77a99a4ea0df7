#!/bin/bash
# HBM traffic of the streamed SpMM (separate passes: FETCH_SIZE, WRITE_SIZE, L2 hit / miss), MI355X_MICROARCH.md HBM section
mkdir -p gpurun_out/pmc_stream_hbm; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" ; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_stream_hbm/pass$i -- python3 tools/profile_stream.py ${1:-256} 3 > gpurun_out/pmc_stream_hbm/pass$i.log 2>&1
  rc=$?; echo "pass$i rc=$rc"; if [ $rc -ne 0 ]; then tail -5 gpurun_out/pmc_stream_hbm/pass$i.log; fi
done
python3 tools/summarize_pmc.py gpurun_out/pmc_stream_hbm spmm_stream
