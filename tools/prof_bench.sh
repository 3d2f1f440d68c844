# per-kernel profile of the default bench command:  bash tools/prof_bench.sh [steps] [warmup]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps ${1:-6} --warmup ${2:-2} --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
f=$(find $R/gpurun_out/prof_bench -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/bench_kernel_stats.csv
tail -1 $R/gpurun_out/prof_bench.log | cut -c1-300
head -32 "$f" | cut -c1-170
find $R/gpurun_out -name '*kernel_trace.csv' -delete   # the traces are large; only the stats travel back
