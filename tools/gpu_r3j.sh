cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_nn_gpu.py -q -m gpu -k "gemm_dec or skinny" 2>&1 | tail -2
timeout 300 python tools/probe_ragged_step.py 2>&1 | grep -v amdgpu.ids | head -4
run() {
python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extra-configs "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', '| value', d['value'], 'ms/step', d['ms_per_step'], 'launches', d.get('launches_per_cycle'), 'rows/step', d['config'].get('tts_rows_per_decode_step'), 'p50/p99', d['p50_tick_latency_ms'], d['p99_tick_latency_ms'], 'logmel', d['roofline_logmel']['frac'], d['sequential_stage_ms'])
"
}
run --tts-lanes 3 --front-lanes 3
run --tts-lanes 4 --front-lanes 3
run --tts-mode lanes
run --tts-lanes 3 --front-lanes 3 --stt-beam 1
