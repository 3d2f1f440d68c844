# matrix-pipe / LDS counters of the log-mel kernel (separate --pmc passes; kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lm_$tag -- python3 $R/tools/probe_logmel_diff.py /tmp/x.npy > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmc_lm_$tag -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_logmel_dft2' in r['Kernel_Name']]
by={}
for r in rows: by.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for k,v in by.items(): print(k, 'per launch', sum(v)/len(v), 'launches', len(v))
PY
done
