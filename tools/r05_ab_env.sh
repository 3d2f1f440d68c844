# A/B of one tuning switch inside the pipeline: bash tools/r05_ab_env.sh NAME v1 v2 ...   (two alternating 12-step C3 runs per value)
R=$GRAFT_REPO_ROOT; cd $R
NAME=$1; shift
for i in 1 2; do
  for v in "$@"; do
    echo "$NAME=$v: $(env $NAME=$v python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extra-configs --no-tick-probe 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"
  done
done
