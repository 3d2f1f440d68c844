# per-kernel stats of the Whisper-base encoder at 128 windows:  bash tools/prof_encoder.sh  -> gpurun_out/r05p/encoder_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
mkdir -p $O
rm -rf $O/prof_enc
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_enc -- python3 $R/tools/probe_encoder.py 128 whisper_base > $O/prof_enc.log 2>&1
cp "$(find $O/prof_enc -name '*kernel_stats.csv' | head -1)" $O/encoder_kernel_stats.csv
tail -1 $O/prof_enc.log
head -20 $O/encoder_kernel_stats.csv | cut -c1-200
