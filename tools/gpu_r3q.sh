cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nn_gpu.py -q -m gpu -k "conv_kernel or small_grid or transpose or whisper or hifigan_and or tts_pipe or tts_infer" 2>&1 | tail -3
timeout 300 python tools/probe_igemm_enc.py 2>&1 | grep -v amdgpu
timeout 300 python tools/probe_encoder.py 128 whisper_base 2>&1 | tail -1
