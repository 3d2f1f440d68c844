# bench knobs after the round-5 kernels: ticket-based tile order of the 256 x 256 GEMM against the static round-robin
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_nn_gpu.py -x -q -k "gemm_big" 2>&1 | tail -2
for rep in 1 2 3; do
  for tk in 0 1; do
    IFH_GEMM_BIG8_TICKETS=$tk timeout 600 python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extra-configs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tickets $tk rep $rep: value %.0f  ms %.1f  p50 tick %.2f p99 %.2f' % (d['value'], d['ms_per_step'], d.get('p50_tick_latency_ms', -1), d.get('p99_tick_latency_ms', -1)))"
  done
done
IFH_GEMM_BIG8_TICKETS=1 timeout 300 python tools/probe_igemm_enc.py 192000 2>&1 | tail -8
