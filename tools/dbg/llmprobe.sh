mkdir -p gpurun_out
timeout -k 5 280 python3 tools/probe_llm.py 64 192 64 > gpurun_out/probe_llm.log 2>&1
echo "rc=$?" >> gpurun_out/probe_llm.log
tail -6 gpurun_out/probe_llm.log
