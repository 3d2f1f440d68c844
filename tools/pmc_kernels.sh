# Per-kernel PMC sums of a probe:  bash tools/pmc_kernels.sh "<counter> <counter> ..." <script.py> [args]   (one rocprofv3 pass per counter,
# kernel-trace only, the program directly behind "--")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
CTRS="$1"; shift
for c in $CTRS; do
  rm -rf $O/$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python3 $R/"$@" > $O/$c.log 2>&1
  python3 - "$c" "$(find $O/$c -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
c, f = sys.argv[1], sys.argv[2]
s = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] != c: continue
    k = r['Kernel_Name'].split('(')[0][:60]
    s[k][0] += float(r['Counter_Value']); s[k][1] += 1
for k, (v, n) in sorted(s.items(), key=lambda kv: -kv[1][0])[:6]:
    print('%-14s %-60s launches %5d  sum %.4g  per launch %.4g' % (c, k, n, v, v / n))
PY
  find $O/$c -name '*.csv' -delete
done
