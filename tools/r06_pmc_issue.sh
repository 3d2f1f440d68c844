#!/bin/bash
# per kernel of a 1 280-chunk vocoder pass: how busy the matrix pipe and the vector issue are, and how much of it together
# (separate --pmc passes; SQ_* are summed over the chip's SQs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06i
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_VALU_MFMA[A-Za-z0-9_]*\|SQ_ACTIVE_INST_[A-Z]*\|SQ_BUSY_CU_CYCLES\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_WAIT_INST_ANY\|SQ_WAIT_ANY\|SQ_INSTS_VALU[A-Za-z0-9_]*\|SQ_INSTS_MFMA\|SQ_INST_CYCLES_VMEM\|GRBM_GUI_ACTIVE\|SQ_WAIT_INST_LDS\|SQ_INSTS_LDS\|SQ_LDS_BANK_CONFLICT\|SQ_LDS_IDX_ACTIVE\|SQ_CYCLES" | sort -u > $O/counters.txt
cat $O/counters.txt | tr '\n' ' '; echo
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rm -rf $O/p$i
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/probe_vocoder.py 2 1280 > $O/p$i.log 2>&1
  f="$(find $O/p$i -name '*counter_collection.csv' | head -1)"
  [ -n "$f" ] && python3 - "$f" >> $O/issue_raw.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][:80]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    cnt[(k,r['Counter_Name'])]+=1
for k,v in acc.items():
    if 'ifh::' not in k: continue
    for c,x in v.items():
        print('%s\t%s\t%.6e\t%d' % (k, c, x/max(1,cnt[(k,c)]), cnt[(k,c)]))
PY
done
cat $O/issue_raw.txt | sort | head -150
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*counter_collection.csv' -delete
