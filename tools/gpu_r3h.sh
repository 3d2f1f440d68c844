cd $GRAFT_REPO_ROOT
run() {
python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extra-configs "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', '| value', d['value'], 'ms/step', d['ms_per_step'], 'launches', d.get('launches_per_cycle'), 'rows/step', d['config'].get('tts_rows_per_decode_step'), 'p50/p99', d['p50_tick_latency_ms'], d['p99_tick_latency_ms'])
"
}
run --tts-lanes 4 --front-lanes 3
IFH_TTS_PRIO=0 run --tts-lanes 4 --front-lanes 3
run --tts-lanes 6 --front-lanes 4
run --tts-lanes 3 --front-lanes 3
run --tts-lanes 4 --front-lanes 4
