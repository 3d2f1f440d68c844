#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06s
run() { # name envs -- args...
  n=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-extra-configs --steps 24 --warmup 3 > gpurun_out/r06s/$n.json 2> gpurun_out/r06s/$n.err
  python - gpurun_out/r06s/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%-14s value %8.1f  ms/cycle %6.2f  launches %6.0f  tick p50 %.2f p99 %.2f  step_ms %s' % (sys.argv[2], d['value'], d['ms_per_step'], d['launches_per_cycle'], d.get('p50_tick_latency_ms',0), d.get('p99_tick_latency_ms',0), d.get('tts_decode_step_ms')))
except Exception as e:
    print(sys.argv[2], 'failed', e, open(sys.argv[1].replace('.json','.err')).read()[-300:])
PY
}
run base1 A=1
run big160 IFH_BIG_CUS=160
run big192 IFH_BIG_CUS=192
run base2 A=1
run big128 IFH_BIG_CUS=128
