for v in "" 1 "" 1; do echo "NO_DMA=$v"; IFH_NO_GEMM_DMA=$v timeout -k 5 280 python3 tools/probe_llm.py 64 192 8 2>&1 | grep "prefill" | cut -c1-120; done
for v in "" 1; do echo "NO_DMA=$v"; IFH_NO_GEMM_DMA=$v timeout -k 5 280 python3 tools/probe_encoder.py 128 whisper_base 2>&1 | tail -3 | cut -c1-200; done
