#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) over ONE forward+backward of each of the 8
# distinct adapter shapes of the AVE Swin-V2-B stack.  A whole bench step under --pmc serialises ~12000 dispatches and
# takes > 40 min; per-shape passes take seconds and are scaled by the schedule (tools/pmc_traffic.py --stack).
set -u
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
SHAPES="2304,128,4096,96 4096,96,2304,128 576,256,1024,192 1024,192,576,256 144,512,256,384 256,384,144,512 36,1024,64,768 64,768,36,1024"
for s in $SHAPES; do
  IFS=, read N C No Co <<< "$s"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/${N}_${C}_${No}_${Co}_$ctr -o p -- \
      python $GRAFT_REPO_ROOT/tools/trace_adapter.py $N $C $No $Co 160 > /dev/null 2>&1
  done
done
ls $OUT | wc -l
