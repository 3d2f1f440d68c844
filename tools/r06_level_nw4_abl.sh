#!/bin/bash
# ablations of the k = 11 block (wrong results by design): 4 = no activation fragment reads, 8 = no MFMAs, 12 = neither (the epilogue
# stream alone), 1 = no weight-fragment reloads
cd "$GRAFT_REPO_ROOT" || exit 1
for v in ${VARIANTS:-base nw4}; do
  b=tools/mb/bin/level_bench_$v
  echo "=== $v"
  for abl in 0 1 4 12 13; do
    echo "--- abl $abl"
    IFH_LEVEL_BARRIER=1 IFH_LEVEL_ABL=$abl timeout 60 $b 1280 11 10 | grep barrier
  done
done
