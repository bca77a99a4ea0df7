#!/bin/bash
# Experiment builds of the lane-per-row layer-1 kernels (lane_stream.hip): tools/lane_variants.sh name "-DMLLP_L1_AHEAD=8" ...
# -> mllp_amd/csrc/libmllp_var_<name>.so (the product objects + lane_stream / stream_api rebuilt with the defines);
# run with MLLP_LIB=libmllp_var_<name>.so python3 tools/bench_lane.py 256 5
set -e
cd "$(dirname "$0")/../mllp_amd/csrc"
name=$1; shift
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-inline-asm -ffp-contract=off"
mkdir -p var
hipcc $F "$@" -c lane_stream.hip -o var/lane_stream_$name.o
hipcc $F "$@" -c stream_api.cpp -o var/stream_api_$name.o
OBJS=$(make -pn 2>/dev/null | sed -n 's/^OBJS = //p' | head -1)
[ -n "$OBJS" ] || OBJS="graph.o host_graph.o host_stream.o mps_reader.o api.o stream_api.o sweep_kernels.o node_kernels.o tiled_kernels.o stream_spmm.o stream_attn.o lane_stream.o stream_build.o tiled_build.o transpose.o fused_kernels.o angle.o"
OBJS=$(echo $OBJS | sed "s/\\\\//g; s/lane_stream.o/var\/lane_stream_$name.o/; s/stream_api.o/var\/stream_api_$name.o/")
hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -ldl -o libmllp_var_$name.so
echo built libmllp_var_$name.so
