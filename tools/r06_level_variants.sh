#!/bin/bash
# build tools/mb/bin/level_bench_<name> in compile variants of csrc/level.hip: name=flags ...  (run in this container; the binaries
# travel to the GPU box).  The harness and the software-pipelined experiment (tools/exp/level_pipe.hip) are compiled once.
cd "$(dirname "$0")/.."
BASE="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000 -DLV_DEV_ABL -DLV_WITH_PIPE -fPIC"
mkdir -p tools/mb/bin tools/mb/obj
H=tools/mb/obj/harness.o; L=tools/mb/obj/pipe.o
[ tools/mb/level_bench.hip -nt $H ] && ( /opt/rocm/bin/hipcc $BASE -c tools/mb/level_bench.hip -o $H 2>&1 | grep -v "hip-link\|warning generated" ) &
[ tools/exp/level_pipe.hip -nt $L -o infernos_amd/csrc/level.h -nt $L ] && ( /opt/rocm/bin/hipcc $BASE -c tools/exp/level_pipe.hip -o $L 2>&1 | grep -v "hip-link" ) &
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  [ "$name" = "$spec" ] && flags=""
  ( /opt/rocm/bin/hipcc $BASE $flags -c infernos_amd/csrc/level.hip -o tools/mb/obj/level_$name.o 2>&1 | grep -v "hip-link" ; echo "compiled $name" ) &
done
wait
for spec in "$@"; do
  name="${spec%%=*}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $H $L tools/mb/obj/level_$name.o -o tools/mb/bin/level_bench_$name && echo "linked $name"
done
