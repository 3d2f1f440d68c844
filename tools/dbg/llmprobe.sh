mkdir -p gpurun_out
for n in 100000 128; do
echo "NW8 from $n"; IFH_GQA_NW8=$n timeout -k 5 200 python3 tools/probe_llm.py 64 192 64 2>&1 | grep "decode\|per launch" | sed 's/.*decode/decode/' | cut -c1-200
done
