set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests/test_continuous_tts_gpu.py -x -q -m gpu -s > gpurun_out/r3a/t_cont.log 2>&1; echo "cont rc=$?" >> gpurun_out/r3a/rc.log
timeout 1200 python -m pytest tests/test_nn_gpu.py -q -m gpu -k "skinny or small_grid or m64 or whisper" > gpurun_out/r3a/t_nn.log 2>&1; echo "nn rc=$?" >> gpurun_out/r3a/rc.log
timeout 900 python -m pytest tests/test_beam_gpu.py -q -m gpu -k "base_beam" -s > gpurun_out/r3a/t_beam.log 2>&1; echo "beam rc=$?" >> gpurun_out/r3a/rc.log
timeout 1200 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -k "continuous or as-benched" -s > gpurun_out/r3a/t_pipe.log 2>&1; echo "pipe rc=$?" >> gpurun_out/r3a/rc.log
timeout 900 python bench.py --steps 16 --warmup 3 --no-extra-configs --no-cpu-baseline --breakdown > gpurun_out/r3a/bench_cont.json 2> gpurun_out/r3a/bench_cont.err; echo "bench cont rc=$?" >> gpurun_out/r3a/rc.log
timeout 900 python bench.py --steps 16 --warmup 3 --no-extra-configs --no-cpu-baseline --tts-mode lanes > gpurun_out/r3a/bench_lanes.json 2> gpurun_out/r3a/bench_lanes.err; echo "bench lanes rc=$?" >> gpurun_out/r3a/rc.log
cat gpurun_out/r3a/rc.log
tail -5 gpurun_out/r3a/t_cont.log gpurun_out/r3a/t_nn.log gpurun_out/r3a/t_beam.log gpurun_out/r3a/t_pipe.log
tail -c 1500 gpurun_out/r3a/bench_cont.json; tail -5 gpurun_out/r3a/bench_cont.err
