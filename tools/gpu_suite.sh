cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/full/gputest.log 2>&1; echo "rc=$?"; tail -n 6 gpurun_out/full/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
