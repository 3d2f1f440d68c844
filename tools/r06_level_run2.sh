#!/bin/bash
# quick variant comparison: pipelined form only, per block and the level
cd "$GRAFT_REPO_ROOT" || exit 1
for b in tools/mb/bin/level_bench_*; do
  echo "=== $b"
  for taps in 3 7 11 0; do timeout 60 $b 1280 $taps 10 | grep -v "two launches\|inf/nan"; done
done
