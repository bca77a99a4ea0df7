#!/bin/bash
# Round-end evidence: rocprofv3 kernel stats of the default bench command + PMC traffic passes for the tiled SpMM.
mkdir -p gpurun_out/final; export TMPDIR=/tmp
step() { local name=$1 secs=$2; shift 2; timeout -k 10 "$secs" "$@" > "gpurun_out/final/$name.log" 2>&1; local rc=$?; echo "[$name] rc=$rc"; tail -n 2 "gpurun_out/final/$name.log" | cut -c1-300; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step bench_plain 900 python3 bench.py
step bench_prof 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_bench -- python3 bench.py --no-cpu-baseline
# the raw kernel trace (every torch kernel of the tiled-copy builder included) exceeds what gpurun copies back: keep the summary
python3 tools/summarize_rocprof.py gpurun_out/final/prof_bench gpurun_out/final/kernel_stats.md > /dev/null; rm -rf gpurun_out/final/prof_bench
step pmc_fetch 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pmc_fetch -- python3 tools/bench_spmm.py 256 3
step pmc_write 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/pmc_write -- python3 tools/bench_spmm.py 256 3
for k in fetch write; do python3 tools/summarize_pmc.py gpurun_out/final/pmc_$k spmm_tiled > gpurun_out/final/pmc_$k.txt; rm -rf gpurun_out/final/pmc_$k; done
step pmc_sq 600 bash tools/pmc_ws.sh 64
rm -rf gpurun_out/pmc_ws
step phase 300 python3 tools/phase_cycles.py 64
export MLLP_BENCH_FORCE_DIST=1; step bench_dist1 600 python3 bench.py --steps 20 --no-synthetic --no-cpu-baseline
