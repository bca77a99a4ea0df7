#!/bin/bash
# One gpurun call, several steps: `tools/gpu_steps.sh name:seconds:command ...` -- each step under `timeout -k 10`, output in
# gpurun_out/<name>.log; a step that was killed by its timeout stops the call (no further GPU step behind a hung one).
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; secs=${rest%%:*}; cmd=${rest#*:}
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "[$name] rc=$rc"; tail -n 12 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed by timeout -> stop"; exit $rc; fi
done
exit 0
