#!/bin/bash
# barrier-form variants of csrc/level.hip: time per block and per level, then phase clocks
cd "$GRAFT_REPO_ROOT" || exit 1
export IFH_LEVEL_BARRIER=1
for b in tools/mb/bin/level_bench_*; do
  echo "=== $b"
  for taps in 3 7 11 0; do timeout 60 $b 1280 $taps 10 | grep -v "inf/nan"; done
  for taps in 3 11; do IFH_LEVEL_ABL=16 timeout 60 $b 1280 $taps 10 | grep "per conv"; done
done
