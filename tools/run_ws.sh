set -e
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "tiled or spmm" 2>&1 | tail -3
timeout -k 10 300 python tools/phase_cycles.py 64 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/bench_spmm.py 64 5 2>&1 | grep "tiled  "
timeout -k 10 300 python tools/bench_spmm.py 256 5 2>&1 | grep "tiled  "
