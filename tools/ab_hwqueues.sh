cd $GRAFT_REPO_ROOT
echo plain; bash tools/ab_bench.sh 2
echo force-dp; bash tools/ab_bench.sh 2 --force-dp
echo "force-dp, 8 hw queues"; GPU_MAX_HW_QUEUES=8 bash tools/ab_bench.sh 2 --force-dp
echo "plain, 8 hw queues"; GPU_MAX_HW_QUEUES=8 bash tools/ab_bench.sh 2
