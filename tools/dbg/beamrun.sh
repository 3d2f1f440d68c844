mkdir -p gpurun_out
timeout -k 5 600 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extra-configs --stt-beam 5 2>/dev/null | tail -1 > gpurun_out/bench_beam5.json
timeout -k 5 600 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extra-configs --stt-beam 1 2>/dev/null | tail -1 > gpurun_out/bench_beam1.json
python - <<'PY'
import json
for n in ('bench_beam5','bench_beam1'):
    d=json.load(open('gpurun_out/%s.json'%n)); print(n, d['value'], d['ms_per_step'], d.get('tick_latency'), d['roofline']['frac'])
PY
