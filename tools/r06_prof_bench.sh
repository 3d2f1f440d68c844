#!/bin/bash
# the timed region of the default bench under rocprofv3 (kernel trace + stats), as tools/prof_r06.sh step (1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
mkdir -p $O; rm -rf $O/prof_bench
IFH_TRACE_MARK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs --no-tick-probe $BENCH_EXTRA > $O/prof_bench.log 2>&1
echo "rc=$?"
f="$(find $O/prof_bench -name '*kernel_stats.csv' | head -1)"
[ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv
t="$(find $O/prof_bench -name '*kernel_trace.csv' | head -1)"
[ -n "$t" ] && python3 $R/tools/trace_busy.py "$t" > $O/bench_busy.txt 2>&1
tail -1 $O/prof_bench.log | cut -c1-300
head -12 $O/bench_busy.txt
[ -n "$t" ] && python3 $R/tools/trace_gaps.py "$t" > $O/bench_gaps.txt 2>&1; head -40 $O/bench_gaps.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name '*agent_info.csv' -delete
