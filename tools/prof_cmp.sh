cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export IFH_FOLD_LN=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fold$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > $GRAFT_REPO_ROOT/gpurun_out/prof_fold$v.log 2>&1
  grep -o '"value": [0-9.]*' $GRAFT_REPO_ROOT/gpurun_out/prof_fold$v.log
done
