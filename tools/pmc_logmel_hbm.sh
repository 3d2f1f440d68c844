# HBM traffic of one log-mel launch (64 x 30 s windows) from the L2 fabric counters, separate --pmc passes, kernel-trace only
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmclm_$c -- python3 $GRAFT_REPO_ROOT/tools/probe_logmel.py 3 > $GRAFT_REPO_ROOT/gpurun_out/pmclm_$c.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/pmclm_$c.log
done
find $GRAFT_REPO_ROOT/gpurun_out -name '*kernel_trace.csv' -delete
