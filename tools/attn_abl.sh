# usage: bash tools/attn_abl.sh "<list of ablation numbers>"   (variant libraries: make -C mllp_amd/csrc abl ABL=n)
MLLP_LIB=libmllp_hip_timing.so python tools/attn_stream_cycles.py 64 0 fwd 2>&1 | grep -v amdgpu.ids
for a in $1; do
  MLLP_LIB=libmllp_hip_abl$a.so python tools/attn_stream_cycles.py 64 0 fwd 2>&1 | grep -v amdgpu.ids
done
