R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
timeout 3300 python3 -m pytest tests -q -m gpu --durations=8 > $O/gpu_tests.txt 2>&1; tail -22 $O/gpu_tests.txt
