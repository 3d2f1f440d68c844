# HBM traffic of one vocoder pass from the L2 fabric counters (MI355X_MICROARCH.md, HBM section):
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE; kernel-trace only.
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -- python3 $GRAFT_REPO_ROOT/tools/probe_vocoder.py 3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log
done
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_FETCH_SIZE/*/
find $GRAFT_REPO_ROOT/gpurun_out -name '*kernel_trace.csv' -delete
