mkdir -p gpurun_out
timeout -k 5 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
tail -4 gpurun_out/gpu_tests.log
timeout -k 5 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 900 gpurun_out/bench_full.json
tail -3 gpurun_out/bench_full.err
