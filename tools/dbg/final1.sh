mkdir -p gpurun_out
timeout -k 5 500 python -m pytest tests/test_attendant_gpu.py tests/test_pipeline_gpu.py -x -q -k "attendant or config5" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_bench
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/prof_bench.log 2>&1
echo "rocprof rc=$?"
f=$(find $R/gpurun_out/prof_bench -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/r02_bench_kernel_stats.csv; head -8 $R/gpurun_out/r02_bench_kernel_stats.csv | cut -c1-150; fi
tail -1 $R/gpurun_out/prof_bench.log | cut -c1-200
find $R/gpurun_out -name '*kernel_trace.csv' -delete
