#!/bin/bash
# the GPU suite file by file (a crash in one file does not hide the others), with the slowest tests of each
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
t0=$(date +%s)
for f in tests/test_*.py; do
  grep -q "mark.gpu\|pytestmark" $f || continue
  s=$(date +%s)
  timeout 900 python -m pytest $f -q -m gpu --durations=25 -p no:cacheprovider > gpurun_out/r06_t_$(basename $f .py).txt 2>&1
  rc=$?
  echo "$f rc=$rc $(( $(date +%s) - s )) s: $(tail -1 gpurun_out/r06_t_$(basename $f .py).txt)"
done
echo "total $(( $(date +%s) - t0 )) s"
