export TMPDIR=/tmp
for v in "" libmllp_var_scalarpart.so "" libmllp_var_scalarpart.so; do
  if [ -n "$v" ]; then export MLLP_LIB=$v; else unset MLLP_LIB; fi
  d=gpurun_out/ab_$RANDOM
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/profile_step.py netlib 60 > /dev/null 2>&1
  echo "== ${v:-product}"; python3 tools/summarize_rocprof.py $d | grep "fused_" | awk -F'|' '{printf "%s %s\n", $2, $4}' | head -7
  rm -rf $d
done
