cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nn_gpu.py tests/test_dsp_gpu.py -q -m gpu -k "attention or whisper or tts_pipe or logmel" 2>&1 | tail -4
timeout 300 python tools/probe_encoder.py 128 whisper_base 2>&1 | tail -2
