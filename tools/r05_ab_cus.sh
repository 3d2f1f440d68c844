cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for cus in 160 256; do
    IFH_GEMM_BIG8_CUS=$cus timeout 600 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extra-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('big8 cus $cus rep $rep: value %.0f  ms %.1f  p50 tick %.2f p99 %.2f' % (d['value'], d['ms_per_step'], d.get('p50_tick_latency_ms', -1), d.get('p99_tick_latency_ms', -1)))"
  done
done
