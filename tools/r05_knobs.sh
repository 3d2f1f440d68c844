# bench knobs after the round-5 kernels: the 256 x 256 GEMM's workgroup count against the vocoder's CU reservation
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cus in 0 256 208; do
  for rep in 1 2; do
    IFH_GEMM_BIG8_CUS=$cus timeout 600 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extra-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('big8 cus $cus rep $rep: value %.0f  ms %.1f  p50 tick %.2f p99 %.2f  vocoder cus %s' % (d['value'], d['ms_per_step'], d.get('p50_tick_latency_ms', -1), d.get('p99_tick_latency_ms', -1), d['config'].get('vocoder_cus_in_timed_region')))"
  done
done
