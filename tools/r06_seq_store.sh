#!/bin/bash
# the sequence kernels' last epilogue with a row's pieces stored back to back: bits (pytest), WRITE_SIZE + write requests per kernel, time per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
mkdir -p $O
(cd $R && timeout 600 python -m pytest tests/test_nn_gpu.py -q -m gpu -x -k "seq or hifigan" 2>&1 | tail -3)
for c in "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $c | tr ' ' '_'); rm -rf $O/pmcs_$n
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmcs_$n -- python3 $R/tools/probe_vocoder.py 2 1280 > $O/pmcs_$n.log 2>&1
  f="$(find $O/pmcs_$n -name '*counter_collection.csv' | head -1)"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][:60]
    if 'seq' not in k and 'chain<64' not in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in acc.items():
    print('%-62s' % k, '  '.join('%s %.4e' % (c, x/max(1,cnt[(k,c)])) for c,x in v.items()))
PY
done
rm -rf $O/prof_vocs; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vocs -- python3 $R/tools/probe_vocoder.py 10 1280 > $O/prof_vocs.log 2>&1
f="$(find $O/prof_vocs -name '*kernel_stats.csv' | head -1)"; grep "seq\|level" "$f" | cut -d, -f1-4 | cut -c1-150
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
