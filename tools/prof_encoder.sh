cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_enc -- python3 $R/tools/probe_encoder.py 64 > $R/gpurun_out/prof_enc.log 2>&1
f=$(find $R/gpurun_out/prof_enc -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), '%9.1f us' % (float(r['AverageNs'])/1e3), '%7.2f ms' % (int(r['TotalDurationNs'])/1e6))
PY
find $R/gpurun_out -name '*kernel_trace.csv' -delete
