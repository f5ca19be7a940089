cd /tmp; export TMPDIR=/tmp
for cfg in "base" "prio0:DGSCT_COMPUTE_PRIORITY=0" "noaux:DGSCT_NO_AUX=1"; do
  tag=${cfg%%:*}; envs=${cfg#*:}; [ "$envs" = "$cfg" ] && envs="X=1"
  rm -rf /tmp/po_$tag
  env $envs timeout 200 rocprofv3 --kernel-trace -d /tmp/po_$tag -o p -- python $GRAFT_REPO_ROOT/tools/trace_stage.py 2 16 4 2>&1 | grep "stage 2"
  python $GRAFT_REPO_ROOT/tools/pair_overlap.py $(find /tmp/po_$tag -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/pair_overlap_$tag.txt
  tail -3 $GRAFT_REPO_ROOT/gpurun_out/pair_overlap_$tag.txt
done
