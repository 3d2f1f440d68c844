# per-kernel stats of the Whisper-base encoder at 128 windows:  bash tools/prof_encoder.sh  -> gpurun_out/r06p/encoder_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${PROF_DIR:-r06p}
mkdir -p $O
rm -rf $O/prof_enc
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_enc -- python3 $R/tools/probe_encoder.py 128 whisper_base > $O/prof_enc.log 2>&1
cp "$(find $O/prof_enc -name '*kernel_stats.csv' | head -1)" $O/encoder_kernel_stats.csv
grep 'encoder B=' $O/prof_enc.log
# and unprofiled, three times (the figure the round's target is stated on)
for i in 1 2 3; do python3 $R/tools/probe_encoder.py 128 whisper_base; done 2>/dev/null | grep 'encoder B=' > $O/encoder_probe.txt; cat $O/encoder_probe.txt
python3 $R/tools/probe_attn_prefill.py 128 8 1500 2>/dev/null | grep -E 'form|identical' >> $O/encoder_probe.txt
python3 $R/tools/probe_layernorm.py 2>/dev/null | grep -E 'rows per wave|copy' >> $O/encoder_probe.txt
head -20 $O/encoder_kernel_stats.csv | cut -c1-200
