cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl; DGSCT_NO_AUX=1 timeout 120 rocprofv3 --kernel-trace -d /tmp/tl -o p -- python $GRAFT_REPO_ROOT/tools/trace_adapter.py 4096 96 2304 128 160 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $(find /tmp/tl -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/timeline_st0a_noaux.txt
