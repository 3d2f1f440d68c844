#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06s
run() { n=$1; shift
  python bench.py --no-cpu-baseline --no-extra-configs --steps 24 --warmup 3 "$@" > gpurun_out/r06s/$n.json 2> gpurun_out/r06s/$n.err
  python - gpurun_out/r06s/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s=d.get('tts_decode_step_ms') or {}
    print('%-10s value %8.1f  ms/cycle %6.2f  tick p50 %.2f p99 %.2f  step in-pipe %.2f idle %.2f rows %s steps/cycle %.1f' % (sys.argv[2], d['value'], d['ms_per_step'], d.get('p50_tick_latency_ms',0), d.get('p99_tick_latency_ms',0), s.get('in_pipeline',0), s.get('idle_gpu',0), d['config'].get('tts_rows_per_decode_step'), s.get('steps_per_cycle',0)))
except Exception as e:
    print(sys.argv[2], 'failed', e, open(sys.argv[1].replace('.json','.err')).read()[-300:])
PY
}
run l5a
run l6 --tts-lanes 6
run l7 --tts-lanes 7
run l5b
run l8 --tts-lanes 8
run l6f5 --tts-lanes 6 --front-lanes 5
