echo "k_gemm_256 random"; timeout -k 5 120 python3 tools/dbg/gemm1.py
echo "k_igemm random"; IFH_NO_GEMM_DMA=1 timeout -k 5 120 python3 tools/dbg/gemm1.py
echo "k_gemm_256 zeros"; timeout -k 5 120 python3 tools/dbg/gemm1.py zeros
echo "k_igemm zeros"; IFH_NO_GEMM_DMA=1 timeout -k 5 120 python3 tools/dbg/gemm1.py zeros
