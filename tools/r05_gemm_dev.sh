# dev loop for gemm_big8 (library from tools/build_gemm_abl.sh): bit parity, then timings and phase clocks
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -k "gemm_big" 2>&1 | tail -3
timeout 600 python tools/probe_gemm_abl.py ${1:-0} 2>&1 | cat
