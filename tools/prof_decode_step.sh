# per-kernel time inside the captured decoder step at 64 vs 192 rows:  bash tools/prof_decode_step.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for g in 1 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_step$g -- python3 $R/tools/probe_graph_launch.py $g > /dev/null 2>&1
  f=$(find $R/gpurun_out/prof_step$g -name '*kernel_stats.csv' | head -1)
  echo "== group $g"; head -9 "$f" | cut -d, -f1-4 | cut -c1-110
done
find $R/gpurun_out -name '*kernel_trace.csv' -delete   # the traces are large; only the stats travel back
