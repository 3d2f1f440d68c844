set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3c
export IFH_TRACE_MARK=1
i=0
for cfg in "--tts-mode continuous" "--tts-mode lanes" "--tts-mode continuous --stt-beam 1" "--tts-mode continuous --tts-lanes 6 --front-lanes 4"; do
i=$((i+1))
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3c/prof_$i -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs --no-tick-probe $cfg > $R/gpurun_out/r3c/prof_$i.log 2>&1
f=$(find $R/gpurun_out/r3c/prof_$i -name '*kernel_trace.csv' | head -1)
echo "== $cfg" > $R/gpurun_out/r3c/busy_$i.txt
python3 $R/tools/trace_busy.py "$f" >> $R/gpurun_out/r3c/busy_$i.txt 2>&1
rm -rf $R/gpurun_out/r3c/prof_$i
python3 - <<PY >> $R/gpurun_out/r3c/busy_$i.txt
import json
d=json.loads([l for l in open('$R/gpurun_out/r3c/prof_$i.log') if l.startswith('{')][-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'launches', d.get('launches_per_cycle'), 'rows/step', d['config'].get('tts_rows_per_decode_step'), d['sequential_stage_ms'])
PY
head -4 $R/gpurun_out/r3c/busy_$i.txt; tail -1 $R/gpurun_out/r3c/busy_$i.txt
done
