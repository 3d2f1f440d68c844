#!/bin/bash
# write requests L2 -> memory of a vocoder pass, per kernel: how many, and how many of them full 64-byte ones (the residual-block
# kernels' 8-byte-per-lane stores: k_resblock_seq writes 1.8-2.1 x its output by WRITE_SIZE, k_resblock_chain 1.0 x)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_WRREQ[A-Za-z0-9_]*\|TCC_EA_WRREQ[A-Za-z0-9_]*\|TCC_WRITE[A-Za-z0-9_]*\|TCC_EA0_WR_UNCACHED[A-Za-z0-9_]*" | sort -u | head -30 > $O/wr_counters.txt
cat $O/wr_counters.txt
for c in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_WRITE_sum TCC_WRITEBACK_sum"; do
  n=$(echo $c | tr ' ' '_')
  rm -rf $O/pmcw_$n
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmcw_$n -- python3 $R/tools/probe_vocoder.py 2 1280 > $O/pmcw_$n.log 2>&1
  f="$(find $O/pmcw_$n -name '*counter_collection.csv' | head -1)"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][:70]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    cnt[(k,r['Counter_Name'])]+=1
for k,v in acc.items():
    if 'ifh::' not in k: continue
    print('%-72s' % k, '  '.join('%s %.3e (n=%d)' % (c, x/max(1,cnt[(k,c)]), cnt[(k,c)]) for c,x in v.items()))
PY
done
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
