# bench.py under environment / flag variations, one line per run: bash tools/exp_bench_env.sh
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --no-extra-configs --no-cpu-baseline $EXTRA > gpurun_out/exp_$name.json 2> gpurun_out/exp_$name.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/exp_$name.json")); print("$name", d["value"], d["ms_per_step"], d["config"].get("tts_rows_per_decode_step"), d["p50_tick_latency_ms"], d["p99_tick_latency_ms"], d.get("worst_tick_ms"), d.get("worst_tick_split"))
except Exception as e:
    print("$name failed", e)
PY
}
for rep in 1 2 3; do
run gb2_$rep A=1
run gb1_$rep IFH_GEMM_BIG_LDS=90000
EXTRA="--cu-reserve 64" run gb1_res64_$rep IFH_GEMM_BIG_LDS=90000
done
