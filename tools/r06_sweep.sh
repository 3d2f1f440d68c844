#!/bin/bash
# default bench (C3) against single-parameter variations, alternating with the default so that box drift shows:  value / ms per cycle
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06s
run() { # name args...
  n=$1; shift
  python bench.py --no-cpu-baseline --no-extra-configs --steps 24 --warmup 3 "$@" > gpurun_out/r06s/$n.json 2> gpurun_out/r06s/$n.err
  python - gpurun_out/r06s/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%-14s value %8.1f  ms/cycle %6.2f  launches %6.0f  tick p50 %.2f p99 %.2f  step_ms %s' % (sys.argv[2], d['value'], d['ms_per_step'], d['launches_per_cycle'], d.get('p50_tick_latency_ms',0), d.get('p99_tick_latency_ms',0), d.get('tts_decode_step_ms')))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
run base1
run lanes6 --tts-lanes 6
run lanes7 --tts-lanes 7
run base2
run front3 --front-lanes 3
run front5 --front-lanes 5
run res64 --cu-reserve 64
run res128 --cu-reserve 128
run base3
