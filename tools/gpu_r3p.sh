cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc -- python3 $R/tools/probe_encoder.py 128 whisper_base > /tmp/prof_enc.log 2>&1
tail -1 /tmp/prof_enc.log
f=$(find /tmp/prof_enc -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:12]:
    print('%-70s calls %5s total %8.2f ms avg %8.1f us %5.1f%%' % (r['Name'][:70], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, 100*int(r['TotalDurationNs'])/tot))
PY
