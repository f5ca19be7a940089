cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 > /tmp/b.json 2>/dev/null
cat /tmp/b.json | tail -1 | cut -c1-200
db=$(find /tmp/tr -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/tools/idle_hist.py $db 0.6 2 > $GRAFT_REPO_ROOT/gpurun_out/r3_idle_hist.txt
