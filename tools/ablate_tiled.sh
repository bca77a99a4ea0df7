#!/bin/bash
# Timing-only ablations of the FIRST tiled SpMM kernel (all waves load and walk; MLLP_TILED_SPMM=v1).  The shipped
# wave-specialised kernel has per-phase cycle counters instead: tools/phase_cycles.py.
for m in 0 3 4 7; do echo "== mask $m (1=no H loads 2=no entry loads 4=no walk)"; MLLP_TILED_SPMM=v1 MLLP_TILED_ABLATION=$m timeout -k 10 200 python tools/bench_spmm.py ${1:-64} 10 2>&1 | grep "tiled  "; done
