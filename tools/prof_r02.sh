# Round-2 profile set (1x MI355X):  bash tools/prof_r02.sh
#   (1) per-kernel stats of the default bench command (C3)           -> gpurun_out/r02_bench_kernel_stats.csv
#   (2) per-kernel stats of 10 vocoder passes at the bench's group size (512 chunks = 4 x 128 calls) -> r02_voc_kernel_stats.csv
#   (3) HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel-trace only) of a vocoder pass and a log-mel launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NCH=${1:-512}; NW=${2:-128}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $R/gpurun_out/prof_bench.log 2>&1
cp "$(find $R/gpurun_out/prof_bench -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/r02_bench_kernel_stats.csv
tail -1 $R/gpurun_out/prof_bench.log | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_voc -- python3 $R/tools/probe_vocoder.py 10 $NCH > $R/gpurun_out/prof_voc.log 2>&1
cp "$(find $R/gpurun_out/prof_voc -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/r02_voc_kernel_stats.csv
head -20 $R/gpurun_out/r02_voc_kernel_stats.csv | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c $R/gpurun_out/pmclm_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/tools/probe_vocoder.py 3 $NCH > $R/gpurun_out/pmc_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmclm_$c -- python3 $R/tools/probe_logmel.py 3 $NW > $R/gpurun_out/pmclm_$c.log 2>&1
done
find $R/gpurun_out -name '*kernel_trace.csv' -delete   # the traces are large; only stats and counters travel back
