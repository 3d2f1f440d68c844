# Round-4 profile set (1x MI355X):  bash tools/prof_r04.sh [vocoder chunks] [log-mel windows]
#   (1) kernel trace + stats of the default bench command (C3, continuous TTS), timed region bracketed by marker kernels
#       -> gpurun_out/r04/bench_kernel_stats.csv, bench_busy.txt (GPU-busy union / kernels resident / top kernels, tools/trace_busy.py)
#   (2) per-kernel stats of 10 vocoder passes at the bench's launch-group size -> voc_kernel_stats.csv
#   (3) HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel-trace only) of a vocoder pass and a log-mel launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
NCH=${1:-1280}; NW=${2:-128}
IFH_TRACE_MARK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs --no-tick-probe > $O/prof_bench.log 2>&1
cp "$(find $O/prof_bench -name '*kernel_stats.csv' | head -1)" $O/bench_kernel_stats.csv
python3 $R/tools/trace_busy.py "$(find $O/prof_bench -name '*kernel_trace.csv' | head -1)" > $O/bench_busy.txt 2>&1
tail -1 $O/prof_bench.log | cut -c1-300
head -12 $O/bench_busy.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_voc -- python3 $R/tools/probe_vocoder.py 10 $NCH > $O/prof_voc.log 2>&1
cp "$(find $O/prof_voc -name '*kernel_stats.csv' | head -1)" $O/voc_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_voc512 -- python3 $R/tools/probe_vocoder.py 10 512 > $O/prof_voc512.log 2>&1
cp "$(find $O/prof_voc512 -name '*kernel_stats.csv' | head -1)" $O/voc512_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lm -- python3 $R/tools/probe_logmel.py 10 $NW > $O/prof_lm.log 2>&1
cp "$(find $O/prof_lm -name '*kernel_stats.csv' | head -1)" $O/logmel_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c $O/pmclm_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/tools/probe_vocoder.py 3 $NCH > $O/pmc_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmclm_$c -- python3 $R/tools/probe_logmel.py 3 $NW > $O/pmclm_$c.log 2>&1
done
find $O -name '*kernel_trace.csv' -delete   # the traces are large; only stats and counters travel back
find $O -name '*agent_info.csv' -delete
du -sh $O
