cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3d
i=0
for cfg in "--tts-mode continuous" "--tts-mode lanes"; do
i=$((i+1))
IFH_TRACE_MARK=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3d/prof_$i -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs --no-tick-probe $cfg > $R/gpurun_out/r3d/prof_$i.log 2>&1
f=$(find $R/gpurun_out/r3d/prof_$i -name '*kernel_trace.csv' | head -1)
echo "== $cfg" > $R/gpurun_out/r3d/busy_$i.txt
python3 $R/tools/trace_busy.py "$f" >> $R/gpurun_out/r3d/busy_$i.txt 2>&1
rm -rf $R/gpurun_out/r3d/prof_$i
cat $R/gpurun_out/r3d/busy_$i.txt
done
cd $R
for cfg in "--tts-lanes 4 --front-lanes 3" "--tts-lanes 6 --front-lanes 3" "--tts-lanes 6 --front-lanes 4" "--tts-lanes 7 --front-lanes 2" "--tts-lanes 5 --front-lanes 4 --stt-beam 1"; do
python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extra-configs $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$cfg', 'value', d['value'], 'ms/step', d['ms_per_step'], 'launches', d.get('launches_per_cycle'), 'rows/step', d['config'].get('tts_rows_per_decode_step'), 'p50/p99', d['p50_tick_latency_ms'], d['p99_tick_latency_ms'], 'voc', d['roofline']['frac'])
"
done
