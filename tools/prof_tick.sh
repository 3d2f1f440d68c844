# kernel trace of the default bench WITH the tick probe -> what ran while the slowest tick waited (tools/trace_tick_stall.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof_tick -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/prof_tick.log 2>&1
python3 $R/tools/trace_tick_stall.py "$(find $O/prof_tick -name '*kernel_trace.csv' | head -1)" > $O/tick_stall.txt 2>&1
cat $O/tick_stall.txt
grep '^{' $O/prof_tick.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['p99_tick_latency_ms'], d.get('worst_tick_split'))"
find $O -name '*kernel_trace.csv' -delete
