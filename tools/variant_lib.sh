#!/bin/bash
# Experiment build of ONE translation unit: tools/variant_lib.sh <name> <file.hip|file.cpp> [-Dflags ...]
# -> mllp_amd/csrc/libmllp_var_<name>.so = the product objects with that unit rebuilt with the defines (results may be
# wrong: timing only).  Run a tool with MLLP_LIB=libmllp_var_<name>.so.
set -e
cd "$(dirname "$0")/../mllp_amd/csrc"
name=$1; src=$2; shift 2
base=${src%.*}
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-inline-asm -ffp-contract=off"
mkdir -p var
hipcc $F "$@" -c $src -o var/${base}_$name.o
OBJS="graph.o host_graph.o host_stream.o mps_reader.o api.o stream_api.o sweep_kernels.o node_kernels.o tiled_kernels.o stream_spmm.o stream_attn.o lane_stream.o stream_build.o tiled_build.o transpose.o fused_kernels.o angle.o"
OBJS=$(echo $OBJS | sed "s/\b$base\.o/var\/${base}_$name.o/")
hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -ldl -o libmllp_var_$name.so
echo built libmllp_var_$name.so
