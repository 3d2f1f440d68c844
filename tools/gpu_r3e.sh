cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_nn_gpu.py -q -m gpu -k "gemm_dec or skinny or small_grid or whisper" > gpurun_out/r3e/t_nn.log 2>&1; echo "nn rc=$?"; tail -n 15 gpurun_out/r3e/t_nn.log
timeout 900 python -m pytest tests/test_continuous_tts_gpu.py tests/test_beam_gpu.py -x -q -m gpu > gpurun_out/r3e/t_cont.log 2>&1; echo "cont+beam rc=$?"; tail -n 5 gpurun_out/r3e/t_cont.log
timeout 300 python tools/probe_ragged_step.py 2>&1 | grep -v amdgpu.ids
IFH_GEMM_DEC_ROWS=100000 timeout 300 python tools/probe_ragged_step.py 2>&1 | grep -v amdgpu.ids
for cfg in "--tts-lanes 4 --front-lanes 3" "--tts-lanes 6 --front-lanes 3" "--tts-lanes 6 --front-lanes 4" "--tts-mode lanes"; do
python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extra-configs $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$cfg', 'value', d['value'], 'ms/step', d['ms_per_step'], 'launches', d.get('launches_per_cycle'), 'rows/step', d['config'].get('tts_rows_per_decode_step'), 'p50/p99', d['p50_tick_latency_ms'], d['p99_tick_latency_ms'], 'voc', d['roofline']['frac'], d['sequential_stage_ms'])
"
done
