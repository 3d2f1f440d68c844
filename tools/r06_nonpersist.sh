#!/bin/bash
# EXPERIMENT: the chip-filling kernels (vocoder, encoder GEMMs) as one tile per workgroup instead of persistent workgroups on 160 CUs
cd "$GRAFT_REPO_ROOT" || exit 1
run() { name=$1; shift; env "$@" python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra-configs > gpurun_out/np_$name.json 2> gpurun_out/np_$name.err; python - <<P
import json
try:
    d=json.loads(open("gpurun_out/np_$name.json").read().strip().splitlines()[-1]); print("$name", d["value"], d["ms_per_step"], d["tts_decode_step_ms"]["in_pipeline"], d["p99_tick_latency_ms"])
except Exception as e: print("$name failed", e); print(open("gpurun_out/np_$name.err").read()[-1500:])
P
}
run base_1 X=1
run np_1 IFH_CU_RESERVE=-1 IFH_EXP_NONPERSIST=1
run base_2 X=1
run np_2 IFH_CU_RESERVE=-1 IFH_EXP_NONPERSIST=1
run res0 IFH_CU_RESERVE=0
