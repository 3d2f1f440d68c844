#!/bin/bash
# run every tools/mb/bin/level_bench_* variant: time per block and per level, both forms, then the phase clocks
cd "$GRAFT_REPO_ROOT" || exit 1
for b in tools/mb/bin/level_bench_*; do
  echo "=== $b"
  for taps in 3 7 11 0; do
    timeout 60 $b 1280 $taps 10
    [ -z "$SKIP_BARRIER" ] && IFH_LEVEL_BARRIER=1 timeout 60 $b 1280 $taps 10
  done
  for taps in 3 7 11; do
    IFH_LEVEL_ABL=16 timeout 60 $b 1280 $taps 10
    [ -z "$SKIP_BARRIER" ] && IFH_LEVEL_BARRIER=1 IFH_LEVEL_ABL=16 timeout 60 $b 1280 $taps 10
  done
done
