#!/bin/bash
for m in 0 3 4 7; do echo "== mask $m (1=no H loads 2=no entry loads 4=no walk)"; MLLP_TILED_ABLATION=$m timeout -k 10 200 python tools/bench_spmm.py ${1:-64} 10 2>&1 | grep "tiled  "; done
