cd $GRAFT_REPO_ROOT
cp dg-sct_amd/libdgsct.so /tmp/new.so
for r in 1 2; do
  cp /tmp/new.so dg-sct_amd/libdgsct.so; echo new; bash tools/ab_bench.sh 2
  cp dg-sct_amd/libdgsct_old.so dg-sct_amd/libdgsct.so; echo old; bash tools/ab_bench.sh 2
done
