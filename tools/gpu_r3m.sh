cd $GRAFT_REPO_ROOT
run() {
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', '| value', d['value'], 'ms/step', d['ms_per_step'], 'rows/step', d['config'].get('tts_rows_per_decode_step'), 'p50/p99', d['p50_tick_latency_ms'], d['p99_tick_latency_ms'])
"
}
run
GPU_MAX_HW_QUEUES=8 run
GPU_MAX_HW_QUEUES=6 run
IFH_TTS_PRIO=0 GPU_MAX_HW_QUEUES=8 run
run --front-lanes 2
