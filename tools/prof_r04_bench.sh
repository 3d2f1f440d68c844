# Round-4: kernel trace + stats of the default bench command, timed region bracketed by marker kernels -> gpurun_out/r04/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
IFH_TRACE_MARK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs --no-tick-probe $BENCH_EXTRA > $O/prof_bench.log 2>&1
cp "$(find $O/prof_bench -name '*kernel_stats.csv' | head -1)" $O/bench_kernel_stats.csv
python3 $R/tools/trace_busy.py "$(find $O/prof_bench -name '*kernel_trace.csv' | head -1)" > $O/bench_busy.txt 2>&1
tail -1 $O/prof_bench.log | cut -c1-200
cat $O/bench_busy.txt
find $O -name '*kernel_trace.csv' -delete
find $O -name '*agent_info.csv' -delete
