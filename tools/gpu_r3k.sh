cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nn_gpu.py tests/test_dsp_gpu.py -q -m gpu -k "transpose or hifigan or logmel or amend" 2>&1 | tail -3
timeout 300 python tools/probe_chain.py 512 2>&1 | tail -6
timeout 300 python tools/probe_chain.py 1280 2>&1 | tail -3
