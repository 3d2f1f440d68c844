R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_nn_gpu.py -x -q -k "resblock_seq" > $O/seq_test.txt 2>&1; tail -15 $O/seq_test.txt
for lv in "64" "64,128" "64,128,256" ""; do
  for c in 64 128 256; do IFH_SEQ_LEVELS=$lv timeout 300 python3 tools/probe_voc_level.py $c 1280 5 2>&1 | grep level | sed "s/^/SEQ=[$lv] /"; done
done | tee $O/seq_levels.txt
timeout 600 python3 -m pytest tests/test_pipeline_gpu.py -x -q -k "block_ingest" > $O/vad_test.txt 2>&1; tail -5 $O/vad_test.txt
