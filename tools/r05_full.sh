R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
timeout 3000 python3 -m pytest tests -x -q -m gpu --durations=15 > $O/gpu_tests.txt 2>&1; tail -25 $O/gpu_tests.txt
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json | head -c 2500; tail -3 $O/bench.err
