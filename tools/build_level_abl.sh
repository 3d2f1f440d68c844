# dev build: level.hip with its ablation instantiations (IFH_LEVEL_ABL), relinked into libinfernos_hip.so
cd "$(dirname "$0")/.."
python -c "from infernos_amd import build as b; b.build(verbose=False)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000 ${LV_EXTRA} -DLV_DEV_ABL -c infernos_amd/csrc/level.hip -o infernos_amd/build/level.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o infernos_amd/libinfernos_hip.so infernos_amd/build/*.o
