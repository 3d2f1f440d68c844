mkdir -p gpurun_out
timeout -k 5 150 python3 tools/probe_beam.py whisper_base 128 5 > gpurun_out/probe_beam.log 2>&1
echo "rc=$?" >> gpurun_out/probe_beam.log
IFH_FOLD_MAX_ROWS=256 timeout -k 5 150 python3 tools/probe_beam.py whisper_base 128 5 > gpurun_out/probe_beam256.log 2>&1
head -2 gpurun_out/probe_beam.log | tail -1; head -2 gpurun_out/probe_beam256.log | tail -1
timeout -k 5 600 python -m pytest tests/test_beam_gpu.py tests/test_workers_gpu.py -x -q 2>&1 | tail -3
