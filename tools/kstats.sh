# per-kernel time of a short traced bench run -> gpurun_out/kstats.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kst && rocprofv3 --kernel-trace --stats -d /tmp/kst -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/b.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kst -name "*.db" | head -1) 60 > $GRAFT_REPO_ROOT/gpurun_out/kstats.txt
