#!/bin/bash
# the CU reserve of the persistent kernels on the final build: 12 cycles each, two rounds
cd "$GRAFT_REPO_ROOT" || exit 1
run() { name=$1; shift; env "$@" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra-configs > gpurun_out/rs_$name.json 2> gpurun_out/rs_$name.err; python - <<P
import json
try:
    d=json.loads(open("gpurun_out/rs_$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["ms_per_step"], d["tts_decode_step_ms"]["in_pipeline"], d["p99_tick_latency_ms"])
except Exception as e: print("$name failed", e)
P
}
for r in 1 2; do
  for c in 96 64 80 112 128; do run c${c}_$r IFH_CU_RESERVE=$c; done
done
