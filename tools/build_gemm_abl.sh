# dev build: gemm_big8.hip / gemm_big.hip with their ablation switches (IFH_GEMM_BIG_ABL), relinked into libinfernos_hip.so
cd "$(dirname "$0")/.."
python -c "from infernos_amd import build as b; b.build(verbose=False)"
for f in gemm_big8 gemm_big; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000 ${GB_EXTRA} -DGB_DEV_ABL -c infernos_amd/csrc/$f.hip -o infernos_amd/build/$f.hip.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o infernos_amd/libinfernos_hip.so infernos_amd/build/*.o
