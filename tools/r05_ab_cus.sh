# A/B of the vocoder's CU reservation (which k_gemm_big8 now follows) inside the C3 bench: alternating 12-step runs
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for r in 96 64 128 32; do
    timeout 600 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extra-configs --cu-reserve $r 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cu_reserve $r rep $rep: value %.0f  ms %.1f  p50 tick %.2f p99 %.2f' % (d['value'], d['ms_per_step'], d.get('p50_tick_latency_ms', -1), d.get('p99_tick_latency_ms', -1)))"
  done
done
