# round-5 GEMM work: bit parity of the 256 x 256 eight-wave GEMM and of its users, then timings with and without it
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_nn_gpu.py tests/test_llm_gpu.py -x -q -k "gemm_big or linear or whisper or qwen2 or encoder or speecht5" 2>&1 | tail -5
for v in 0 1; do
  echo "== IFH_GEMM_BIG8=$v"
  IFH_GEMM_BIG8=$v timeout 300 python tools/probe_igemm_enc.py 192000 2>&1 | tail -8
  IFH_GEMM_BIG8=$v timeout 300 python tools/probe_encoder.py 128 whisper_base 2>&1 | tail -1
  IFH_GEMM_BIG8=$v timeout 300 python tools/probe_llm.py 2>&1 | tail -6
done
