#!/bin/bash
# rocprofv3 kernel stats of a few eager training steps on a synthetic batch: tools/prof_syn.sh [instances] [tag]
n=${1:-64}; tag=${2:-syn}; export TMPDIR=/tmp; mkdir -p gpurun_out/$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python3 tools/profile_step.py synthetic 3 $n > gpurun_out/$tag/run.log 2>&1 || { tail -5 gpurun_out/$tag/run.log; exit 1; }
python3 tools/summarize_rocprof.py gpurun_out/$tag/prof gpurun_out/$tag/stats.md | grep "sweep_kernel\|tiled\|bwd_pre\|param_stats\|node_qp" | head -14
