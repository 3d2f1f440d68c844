# Round-6 profile set (1x MI355X):  bash tools/prof_r06.sh [vocoder chunks] [bench-line group size] [log-mel windows]
#   (1) kernel trace + stats of the default bench command (C3, continuous TTS, LSTM VAD), timed region bracketed by marker kernels
#       -> gpurun_out/r06p/bench_kernel_stats.csv, bench_busy.txt (GPU-busy union / kernels resident / top kernels, tools/trace_busy.py)
#   (2) per-kernel stats of 10 vocoder passes at 1 280 chunks, at the bench line's render-group size and at 512 chunks
#   (3) HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel-trace only) of a vocoder pass and a log-mel launch
#   (4) the log-mel kernel's issue counters (SQ_*: one pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
mkdir -p $O
NCH=${1:-1280}; NGRP=${2:-2048}; NW=${3:-128}
IFH_TRACE_MARK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs --no-tick-probe > $O/prof_bench.log 2>&1
cp "$(find $O/prof_bench -name '*kernel_stats.csv' | head -1)" $O/bench_kernel_stats.csv
python3 $R/tools/trace_busy.py "$(find $O/prof_bench -name '*kernel_trace.csv' | head -1)" > $O/bench_busy.txt 2>&1
tail -1 $O/prof_bench.log | cut -c1-300
head -12 $O/bench_busy.txt
for n in $NCH $NGRP 512; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_voc$n -- python3 $R/tools/probe_vocoder.py 10 $n > $O/prof_voc$n.log 2>&1
  cp "$(find $O/prof_voc$n -name '*kernel_stats.csv' | head -1)" $O/voc${n}_kernel_stats.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lm -- python3 $R/tools/probe_logmel.py 10 $NW > $O/prof_lm.log 2>&1
cp "$(find $O/prof_lm -name '*kernel_stats.csv' | head -1)" $O/logmel_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c $O/pmclm_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/tools/probe_vocoder.py 3 $NCH > $O/pmc_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmclm_$c -- python3 $R/tools/probe_logmel.py 3 $NW > $O/pmclm_$c.log 2>&1
done
rm -rf $O/pmclm_sq
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmclm_sq -- python3 $R/tools/probe_logmel.py 3 $NW > $O/pmclm_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmclm_sq2 -- python3 $R/tools/probe_logmel.py 3 $NW > $O/pmclm_sq2.log 2>&1
python3 - $O <<'PY'
import csv, glob, json, os, sys, collections
O = sys.argv[1]
res = collections.defaultdict(float); n = collections.defaultdict(int)
for d in ('pmclm_sq', 'pmclm_sq2'):
    fs = glob.glob(os.path.join(O, d, '**', '*counter_collection.csv'), recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        if 'k_logmel_fft' in r['Kernel_Name']:
            res[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
json.dump({k: res[k] / n[k] for k in res}, open(os.path.join(O, 'logmel_sq_counters.json'), 'w'), indent=1)
print(json.dumps({k: res[k] / n[k] for k in res}))
PY
bash $R/tools/prof_encoder.sh > $O/prof_encoder.out 2>&1; tail -3 $O/prof_encoder.out
find $O -name '*kernel_trace.csv' -delete   # the traces are large; only stats and counters travel back
find $O -name '*agent_info.csv' -delete
du -sh $O
