#!/bin/bash
# the round-6 abort (a CUDAGraph destroyed by the collector while another thread captures): the pipeline file in ONE process, twice
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
  IFH_ISOLATED=inline timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -x -k "cycle_small or pipelined_tts or continuous_tts_schedule or block_ingest or paced_ticks" 2>&1 | tail -3
done
