cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
(for i in $(seq 1 400); do rocm-smi --showpower --showclocks --showuse --json 2>/dev/null | head -c 1200; echo; sleep 0.25; done) > gpurun_out/r3l/smi.log 2>&1 &
SMI=$!
python3 bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-extra-configs 2>/dev/null | tail -1 | cut -c1-300
kill $SMI
python3 - <<'PY'
import json
rows=[]
for l in open('gpurun_out/r3l/smi.log'):
    l=l.strip()
    if not l.startswith('{'): continue
    try: d=json.loads(l)
    except Exception: continue
    c=d.get('card0',{})
    rows.append(c)
print('samples', len(rows))
if rows:
    print('keys', list(rows[0].keys())[:20])
    for r in rows[::8]:
        print({k:v for k,v in r.items() if any(s in k.lower() for s in ('power','sclk','mclk','use','fclk'))})
PY
