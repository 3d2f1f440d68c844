set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_continuous_tts_gpu.py -x -q -m gpu > gpurun_out/r3b/t_cont.log 2>&1; echo "cont rc=$?" >> gpurun_out/r3b/rc.log
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -k "continuous" > gpurun_out/r3b/t_pipe.log 2>&1; echo "pipe rc=$?" >> gpurun_out/r3b/rc.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in continuous lanes; do
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3b/prof_$mode -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs --no-tick-probe --tts-mode $mode > $R/gpurun_out/r3b/prof_$mode.log 2>&1
f=$(find $R/gpurun_out/r3b/prof_$mode -name '*kernel_trace.csv' | head -1)
python3 $R/tools/trace_busy.py "$f" 0.55 > $R/gpurun_out/r3b/busy_$mode.txt 2>&1
rm -rf $R/gpurun_out/r3b/prof_$mode
done
cd $R
cat gpurun_out/r3b/rc.log; tail -n 3 gpurun_out/r3b/t_cont.log; tail -n 3 gpurun_out/r3b/t_pipe.log
cat gpurun_out/r3b/busy_continuous.txt; cat gpurun_out/r3b/busy_lanes.txt
