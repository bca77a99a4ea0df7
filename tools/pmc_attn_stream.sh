#!/bin/bash
# usage: bash tools/pmc_attn_stream.sh [instances] [dst_is_var] [tag]
# kernel trace + PMC passes (separate --pmc runs, no trace domains) over the streamed attention sweeps of one conv
n=${1:-64}; dv=${2:-0}; tag=${3:-r04_attn_stream}
mkdir -p gpurun_out/$tag; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- python3 tools/profile_attn_stream.py $n 3 stream $dv > gpurun_out/$tag/trace.log 2>&1
rc=$?; echo "trace rc=$rc"; if [ $rc -ne 0 ]; then tail -5 gpurun_out/$tag/trace.log; exit $rc; fi
python3 tools/summarize_rocprof.py gpurun_out/$tag/trace > gpurun_out/${tag}_kernel_stats.md 2>&1 || true
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD TA_BUSY_avr" \
           "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  if [ $i -gt ${PMC_PASSES:-4} ]; then break; fi
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/$tag/pass$i -- python3 tools/profile_attn_stream.py $n 3 stream $dv > gpurun_out/$tag/pass$i.log 2>&1
  rc=$?; echo "pass$i rc=$rc"; if [ $rc -ne 0 ]; then tail -5 gpurun_out/$tag/pass$i.log; exit $rc; fi
done
for k in fwd16_stream_kernel bwddst16_stream_kernel bwdsrc16_stream_kernel; do python3 tools/summarize_pmc.py gpurun_out/$tag $k; done > gpurun_out/${tag}_pmc.txt
rm -rf gpurun_out/$tag/pass*/ gpurun_out/$tag/trace
cat gpurun_out/${tag}_kernel_stats.md | head -30; cat gpurun_out/${tag}_pmc.txt
