R=$GRAFT_REPO_ROOT; cd $R
python3 tools/probe_seq.py 1280 256,64 2>&1 | grep "C="
for lv in "64,256" "" "256" "64"; do for c in 256 64; do IFH_SEQ_LEVELS=$lv python3 tools/probe_voc_level.py $c 1280 5 2>&1 | grep level | sed "s/^/SEQ=[$lv] /"; done; done
