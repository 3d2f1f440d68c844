# A/B of the beams' cross-attention kernel inside the pipeline: bash tools/r05_ab_few.sh  (IFH_ATTN_FEW = 0: k_attn_prefill, 1: k_attn_prefill_few)
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do
  for f in 0 1; do
    echo "few=$f: $(IFH_ATTN_FEW=$f python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extra-configs --no-tick-probe 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"
  done
done
