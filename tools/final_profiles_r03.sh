#!/bin/bash
# Round-3 evidence in one gpurun call: the default bench line, rocprofv3 kernel stats of the same command, PMC traffic
# passes (separate --pmc runs) and SQ / LDS counters of the streamed SpMM, its in-kernel stamps (timing build).
mkdir -p gpurun_out/final; export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2; timeout -k 10 "$secs" "$@" > "gpurun_out/final/$name.log" 2>&1; local rc=$?; echo "[$name] rc=$rc"; tail -n 2 "gpurun_out/final/$name.log" | cut -c1-300; if grep -q "Memory access fault" "gpurun_out/final/$name.log"; then exit 9; fi; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
part=${1:-all}
if [ $part != b ]; then
step bench_plain 900 python3 bench.py
step bench_prof 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_bench -- python3 bench.py --no-cpu-baseline
# the raw kernel trace (every torch kernel of the tiled-copy builder included) exceeds what gpurun copies back: keep the summary
python3 tools/summarize_rocprof.py gpurun_out/final/prof_bench gpurun_out/final/kernel_stats.md > /dev/null; rm -rf gpurun_out/final/prof_bench
step netlib_prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_netlib -- python3 tools/profile_step.py netlib 30
python3 tools/summarize_rocprof.py gpurun_out/final/prof_netlib gpurun_out/final/netlib_kernel_stats.md > /dev/null; rm -rf gpurun_out/final/prof_netlib
fi
if [ $part = a ]; then exit 0; fi
step pmc_fetch 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pmc_fetch -- python3 tools/profile_stream.py 256 3
step pmc_write 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/pmc_write -- python3 tools/profile_stream.py 256 3
step pmc_l2 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/final/pmc_l2 -- python3 tools/profile_stream.py 256 3
for k in fetch write l2; do python3 tools/summarize_pmc.py gpurun_out/final/pmc_$k spmm_stream > gpurun_out/final/pmc_$k.txt; rm -rf gpurun_out/final/pmc_$k; done
step pmc_sq 600 bash tools/pmc_stream.sh 64
rm -rf gpurun_out/pmc_stream
step stamps 300 python3 tools/stream_cycles.py 64 both
export MLLP_BENCH_FORCE_DIST=1; step bench_dist1 600 python3 bench.py --steps 20 --no-synthetic --no-cpu-baseline
