#!/bin/bash
# One gpurun call: GPU parity tests, then a short bench; stops if a step was killed by its timeout.
mkdir -p gpurun_out
step() {  # name, seconds, command...
  local name=$1 secs=$2; shift 2
  timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "[$name] rc=$rc"; tail -n 4 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed by timeout -> stop"; exit $rc; fi
  return 0
}
for s in "$@"; do
  case $s in
    tests) step pytest_gpu 900 python -m pytest tests -m gpu -x -q ;;
    diagfull) step diag_full 300 python tests/diag/gpu_diag.py full ;;
    bench) step bench 900 python bench.py --steps 50 --warmup 5 ;;
    benchquick) step benchquick 600 python bench.py --steps 20 --warmup 3 --synthetic-instances 32 ;;
    diag) step diag_subset 300 python tests/diag/gpu_diag.py subset ; step diag_subset_t 300 python tests/diag/gpu_diag.py subset 4 16 ;;
    prof) mkdir -p gpurun_out/prof_netlib gpurun_out/prof_syn; export TMPDIR=/tmp
          step prof_netlib 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_netlib -- python3 tools/profile_step.py netlib 10
          step prof_syn 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_syn -- python3 tools/profile_step.py synthetic 5 32 ;;
    smoke) step smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
  esac
done
