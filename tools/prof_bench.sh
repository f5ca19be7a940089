# usage (on the GPU box): bash tools/prof_bench.sh <tag> [bench args]   -> gpurun_out/<tag>_kernel_stats.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst_$tag
rocprofv3 --kernel-trace --stats -d /tmp/kst_$tag -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline "$@" > /tmp/b_$tag.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kst_$tag -name "*.db" | head -1) 60 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.txt
grep "^{" /tmp/b_$tag.log | cut -c1-160
cd $GRAFT_REPO_ROOT
