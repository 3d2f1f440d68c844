#!/bin/bash
# A/B on one box: the residual-block kernels (chain.hip, seq.hip, level.hip) compiled without packed-f32 vector instructions
# (-Xclang -target-feature -Xclang -packed-fp32-ops: v_pk_add/mul_f32 behind an MFMA pay 12-20 extra clocks, tools/mb/mfma_valu.hip)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06n
mkdir -p $O
stats() { # name
  rm -rf $O/prof_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -- python3 $R/tools/probe_vocoder.py 10 1280 > $O/prof_$1.log 2>&1
  cp "$(find $O/prof_$1 -name '*kernel_stats.csv' | head -1)" $O/$1_kernel_stats.csv
  python3 - $O/$1_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows if 'ifh::' in r['Name'])
print('  pass %.1f us' % (tot/10/1e3))
for r in rows:
    if 'k_resblock' in r['Name'] or 'k_gemm_big8' in r['Name']: print('   %8.1f us  %s' % (float(r['AverageNs'])/1e3, r['Name'][:80]))
PY
  find $O/prof_$1 -name '*.csv' -delete
}
hash_audio() {
  python3 - <<PY
import sys, hashlib
sys.path.insert(0, "$R")
import torch
from infernos_amd import _lib
from infernos_amd.engines.vocoder import HifiGan
from infernos_amd.weights import synth_state_dict
dev = _lib.require_device('cuda:0')
voc = HifiGan(synth_state_dict('hifigan', 0), dev)
torch.manual_seed(3)
for n in (5, 300, 1280):
    x = torch.randn(n, 12, 80, device=dev).to(torch.bfloat16)
    y = voc(x).clone(); torch.cuda.synchronize()
    print('  audio', n, hashlib.sha1(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16])
PY
}
echo "=== A (committed flags)"; stats A; hash_audio
cd $R
python3 - <<'PY'
import re
p='infernos_amd/build.py'; s=open(p).read()
s=s.replace("EXTRA_FLAGS = {'attn.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}",
 "NOPK = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']\nEXTRA_FLAGS = {'attn.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form'], 'chain.hip': NOPK, 'seq.hip': NOPK, 'level.hip': NOPK}")
open(p,'w').write(s)
PY
touch infernos_amd/csrc/chain.hip infernos_amd/csrc/seq.hip infernos_amd/csrc/level.hip
s=$(date +%s); python3 -c "from infernos_amd import build as b; b.build()" > $O/rebuild.log 2>&1; echo "rebuild rc=$? $(( $(date +%s) - s )) s"; tail -2 $O/rebuild.log
cd /tmp
echo "=== B (no packed f32 in chain / seq / level)"; stats B; hash_audio
echo "=== A again is not possible without a second rebuild: compare the k_gemm_big8 line (unchanged code) for the box's drift"
