#!/bin/bash
# Round-4 evidence in one gpurun call: the default bench line, rocprofv3 kernel stats of the same command, the Netlib step's
# kernel stats, PMC passes (separate --pmc runs, no trace domains) of the streamed SpMM and of the lane-per-row layer-1
# kernels, the AngleModel step's kernel stats, the forced-distributed single-rank bench.  Summaries land in gpurun_out/final/.
mkdir -p gpurun_out/final; export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2; timeout -k 10 "$secs" "$@" > "gpurun_out/final/$name.log" 2>&1; local rc=$?; echo "[$name] rc=$rc"; tail -n 2 "gpurun_out/final/$name.log" | cut -c1-300; if grep -q "Memory access fault" "gpurun_out/final/$name.log"; then exit 9; fi; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
part=${1:-all}
if [ $part != b ]; then
step bench_plain 900 python3 bench.py
step bench_prof 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_bench -- python3 bench.py --no-cpu-baseline
python3 tools/summarize_rocprof.py gpurun_out/final/prof_bench gpurun_out/final/kernel_stats.md > /dev/null; rm -rf gpurun_out/final/prof_bench
step netlib_prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_netlib -- python3 tools/profile_step.py netlib 30
python3 tools/summarize_rocprof.py gpurun_out/final/prof_netlib gpurun_out/final/netlib_kernel_stats.md > /dev/null; rm -rf gpurun_out/final/prof_netlib
step syn_prof 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_syn -- python3 tools/profile_step.py synthetic 4 256
python3 tools/summarize_rocprof.py gpurun_out/final/prof_syn gpurun_out/final/synthetic256_kernel_stats.md > /dev/null; rm -rf gpurun_out/final/prof_syn
step angle_prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_angle -- python3 tools/bench_angle.py 25fv47 256 10
python3 tools/summarize_rocprof.py gpurun_out/final/prof_angle gpurun_out/final/angle_kernel_stats.md > /dev/null; rm -rf gpurun_out/final/prof_angle
step angle_bench 200 python3 tools/bench_angle.py 25fv47 256 20
fi
if [ $part = a ]; then exit 0; fi
step pmc_fetch 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pmc_fetch -- python3 tools/profile_stream.py 256 3
step pmc_write 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/pmc_write -- python3 tools/profile_stream.py 256 3
for k in fetch write; do python3 tools/summarize_pmc.py gpurun_out/final/pmc_$k spmm_stream > gpurun_out/final/pmc_$k.txt; rm -rf gpurun_out/final/pmc_$k; done
step lane_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/lane_trace -- python3 tools/profile_lane.py 256 3
python3 tools/summarize_rocprof.py gpurun_out/final/lane_trace gpurun_out/final/lane_kernel_stats.md > /dev/null; rm -rf gpurun_out/final/lane_trace
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD TA_BUSY_avr"; do
  i=$((i+1))
  step lane_pmc$i 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/final/lane_pmc/pass$i -- python3 tools/profile_lane.py 256 3
done
python3 tools/summarize_pmc.py gpurun_out/final/lane_pmc lane1_kernel > gpurun_out/final/lane_pmc.txt; rm -rf gpurun_out/final/lane_pmc
export MLLP_BENCH_FORCE_DIST=1; step bench_dist1 600 python3 bench.py --steps 20 --no-synthetic --no-cpu-baseline
