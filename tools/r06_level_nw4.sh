#!/bin/bash
# the C = 32 level with FOUR waves per workgroup (one per SIMD, 14 row blocks each) against the product's eight: per block and per level,
# harness checksums must agree
cd "$GRAFT_REPO_ROOT" || exit 1
for v in ${VARIANTS:-base nw4 nw4pf5}; do
  b=tools/mb/bin/level_bench_$v
  echo "=== $v"
  for taps in 3 7 11 0; do
    IFH_LEVEL_BARRIER=1 timeout 60 $b 1280 $taps 10
  done
  for taps in 3 7 11; do
    IFH_LEVEL_BARRIER=1 IFH_LEVEL_ABL=16 timeout 60 $b 1280 $taps 10
  done
done
