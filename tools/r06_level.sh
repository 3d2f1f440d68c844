#!/bin/bash
# round 6: the pipelined level kernel against the barrier form -- bits (pytest) and time (probe_level), a stress loop for the counter protocol
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_nn_gpu.py -q -m gpu -k "resblock_level" -x 2>&1 | tail -15
echo "--- pipelined"; timeout 300 python tools/probe_level.py 1280 2>&1 | tail -6
echo "--- barrier"; IFH_LEVEL_BARRIER=1 timeout 300 python tools/probe_level.py 1280 2>&1 | tail -6
echo "--- pipelined 2048"; timeout 300 python tools/probe_level.py 2048 2>&1 | tail -3
