# bench.py under environment / flag variations, one line per run: bash tools/exp_bench_env.sh
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --no-extra-configs --no-cpu-baseline $EXTRA > gpurun_out/exp_$name.json 2> gpurun_out/exp_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/exp_$name.json")); print("$name", d["value"], d["ms_per_step"], d["config"].get("tts_rows_per_decode_step"), d["p50_tick_latency_ms"], d["p99_tick_latency_ms"])
except Exception as e:
    print("$name failed", e)
PY
}
for rep in 1 2; do
EXTRA="--tts-lanes 5" run l5_$rep A=1
EXTRA="--tts-lanes 6" run l6_$rep A=1
EXTRA="--tts-lanes 7" run l7_$rep A=1
EXTRA="--tts-lanes 6" run l6_tick96_$rep IFH_TICK_CUS=160,96
EXTRA="--tts-lanes 6" run l6_tick256_$rep IFH_TICK_CUS=0,256
done
