#!/bin/bash
# rocprofv3 kernel averages of the AngleModel attention kernels for the product library and the ablation variants
# (tools/variant_lib.sh ang<mask> angle.hip -DMLLP_ANGLE_ABL=<mask>): tools/angle_abl.sh "" ang1 ang2 ...
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "$@"; do
  d=gpurun_out/angle_abl_$v
  if [ -n "$v" ]; then export MLLP_LIB=libmllp_var_$v.so; else unset MLLP_LIB; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_angle.py 25fv47 256 6 > /dev/null 2>&1 || { echo "[$v] failed"; rm -rf $d; continue; }
  python3 tools/summarize_rocprof.py $d | grep "attn_kernel" | awk -v v="$v" -F'|' '{printf "[%s] %s avg %s us\n", v, $2, $4}'
  rm -rf $d
done
