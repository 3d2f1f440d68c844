R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_nn_gpu.py -x -q -k "resblock or hifigan or conv_ring or vocoder" > $O/voc_tests.txt 2>&1; tail -5 $O/voc_tests.txt
python3 tools/probe_seq.py 1280 64,256 2>&1 | grep "C="
for c in 64 128 256 32; do python3 tools/probe_voc_level.py $c 1280 5 2>&1 | grep level; done
python3 - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from infernos_amd import _lib
from infernos_amd.engines.vocoder import HifiGan
from infernos_amd.weights import synth_state_dict
dev = _lib.require_device('cuda:0')
voc = HifiGan(synth_state_dict('hifigan', 0), dev)
for n in (1280, 2048, 512):
    x = torch.randn(n, 12, 80, device=dev).to(torch.bfloat16)
    for _ in range(3): voc(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): voc(x)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print('vocoder pass %d chunks: %.3f ms = %.0f TF/s (%.1f %% of 2.5 PF)' % (n, t * 1e3, n * 3.28 / t / 1e3, n * 3.28 / t / 1e3 / 25))
PY
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
