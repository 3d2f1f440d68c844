# round-5 baseline characterisation of the vocoder kernels (1x MI355X): bash tools/r05_baseline.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
PROF=1 timeout 600 python3 tools/probe_chain.py 1280 > $O/base_chain.txt 2>&1
tail -30 $O/base_chain.txt
cp infernos_amd/libinfernos_hip.so /tmp/lib_keep.so
timeout 900 bash tools/build_level_abl.sh > $O/base_level_build.txt 2>&1
timeout 600 python3 tools/probe_level_abl.py 1280 > $O/base_level_abl.txt 2>&1
cat $O/base_level_abl.txt
cp /tmp/lib_keep.so infernos_amd/libinfernos_hip.so
