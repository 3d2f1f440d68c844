# per-kernel time of the HiFi-GAN vocoder pass and its HBM traffic (separate --pmc passes):  bash tools/prof_vocoder.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_voc -- python3 $R/tools/probe_vocoder.py 10 > $R/gpurun_out/prof_voc.log 2>&1
f=$(find $R/gpurun_out/prof_voc -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/voc_kernel_stats.csv
head -25 "$f" | cut -c1-200
find $R/gpurun_out -name '*kernel_trace.csv' -delete   # the traces are large; only the stats travel back
