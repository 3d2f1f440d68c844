# SQ counter sets per vocoder level (1x MI355X): bash tools/r05_pmc_levels.sh "<levels>" [chunks]    -> gpurun_out/r05/pmc_levels.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
LV="${1:-32 64 128 256}"; N=${2:-1280}
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
      "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" \
      "GRBM_GUI_ACTIVE")
: > $O/pmc_levels.txt
for c in $LV; do
  python3 $R/tools/probe_voc_level.py $c $N 5 >> $O/pmc_levels.txt 2>&1
  i=0
  for s in "${SETS[@]}"; do
    rm -rf $O/pmc_tmp
    rocprofv3 --pmc $s --kernel-trace --output-format csv -d $O/pmc_tmp -- python3 $R/tools/probe_voc_level.py $c $N 2 > $O/pmc_tmp.log 2>&1
    python3 - "$c" "$(find $O/pmc_tmp -name '*counter_collection.csv' | head -1)" >> $O/pmc_levels.txt <<'PY'
import csv, sys, collections
c, f = sys.argv[1], sys.argv[2]
s = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    if 'ifh::' not in k: continue
    k = k.replace('void ifh::', '')[:70]
    e = s[k][r['Counter_Name']]; e[0] += float(r['Counter_Value']); e[1] += 1
for k, d in s.items():
    for cn, (v, n) in d.items():
        print('C=%-4s %-70s %-26s launches %4d  per launch %.5g' % (c, k, cn, n, v / n))
PY
    i=$((i+1))
  done
done
rm -rf $O/pmc_tmp
cat $O/pmc_levels.txt
