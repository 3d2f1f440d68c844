# part (1) of tools/prof_r05.sh alone: kernel trace + stats of the default bench command, timed region bracketed by marker kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
mkdir -p $O
IFH_TRACE_MARK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs --no-tick-probe > $O/prof_bench.log 2>&1
cp "$(find $O/prof_bench -name '*kernel_stats.csv' | head -1)" $O/bench_kernel_stats.csv
python3 $R/tools/trace_busy.py "$(find $O/prof_bench -name '*kernel_trace.csv' | head -1)" > $O/bench_busy.txt 2>&1
tail -1 $O/prof_bench.log | cut -c1-300
head -12 $O/bench_busy.txt
find $O -name '*kernel_trace.csv' -delete
find $O -name '*agent_info.csv' -delete
