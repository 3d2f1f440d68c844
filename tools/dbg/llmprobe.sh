mkdir -p gpurun_out
echo "plain"; IFH_GEMM_DMA=0 timeout -k 5 280 python3 tools/probe_llm.py 64 192 8 2>&1 | grep "prefill per" | cut -c1-300
echo "remap"; IFH_IGEMM_REMAP=1 IFH_GEMM_DMA=0 timeout -k 5 280 python3 tools/probe_llm.py 64 192 8 2>&1 | grep "prefill per" | cut -c1-300
IFH_IGEMM_REMAP=1 IFH_GEMM_DMA=0 timeout -k 5 600 python -m pytest tests/test_nn_gpu.py -x -q -k "conv_kernel_matches" 2>&1 | tail -3
