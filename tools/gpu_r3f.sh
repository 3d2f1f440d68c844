cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests/test_nn_gpu.py -q -m gpu -k "gemm_dec or skinny or small_grid" > gpurun_out/r3f/t_nn.log 2>&1; echo "nn rc=$?"; tail -n 8 gpurun_out/r3f/t_nn.log
timeout 300 python tools/probe_ragged_step.py 2>&1 | grep -v amdgpu.ids
IFH_GEMM_DEC_ROWS=100000 timeout 300 python tools/probe_ragged_step.py 2>&1 | grep -v amdgpu.ids | head -4
timeout 300 python tools/probe_beam.py 2>&1 | tail -8
IFH_GEMM_DEC_ROWS=100000 timeout 300 python tools/probe_beam.py 2>&1 | tail -8
