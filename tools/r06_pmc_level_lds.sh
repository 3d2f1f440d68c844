#!/bin/bash
# where the C = 32 level's LDS bank conflicts come from: the k = 11 block of the stand-alone harness, full / no fragment reads (ABL 4) /
# no reads and no MFMAs (ABL 12: the epilogue's ds_write_b64 and the weight-fragment copies remain)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06l
mkdir -p $O
for abl in 0 4 12; do
  rm -rf $O/a$abl
  IFH_LEVEL_BARRIER=1 IFH_LEVEL_ABL=$abl rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/a$abl -- $R/tools/mb/bin/level_bench_base 1280 11 3 > $O/a$abl.log 2>&1
  f="$(find $O/a$abl -name '*counter_collection.csv' | head -1)"
  echo "--- ABL $abl"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0][:60]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,v in acc.items():
    print(k, '  '.join('%s %.4e' % (c, x/max(1,cnt[(k,c)])) for c,x in sorted(v.items())))
PY
done
find $O -name '*.csv' -delete
