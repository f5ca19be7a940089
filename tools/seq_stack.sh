#!/bin/bash
# Kernel sequence (durations, grids) of ONE forward+backward of each of the 8 adapter shapes of the AVE Swin-V2-B stack.
set -u
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/seq
mkdir -p $OUT
SHAPES="${SHAPES:-2304,128,4096,96 4096,96,2304,128 576,256,1024,192 1024,192,576,256 144,512,256,384 256,384,144,512 36,1024,64,768 64,768,36,1024}"
for s in $SHAPES; do
  IFS=, read N C No Co <<< "$s"
  rm -rf /tmp/seq_$N
  timeout 120 rocprofv3 --kernel-trace -d /tmp/seq_$N -o p -- python $GRAFT_REPO_ROOT/tools/trace_adapter.py $N $C $No $Co 160 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_seq.py $(find /tmp/seq_$N -name "*.db" | head -1) > $OUT/seq_${N}_${C}_${No}_${Co}.txt
  head -1 $OUT/seq_${N}_${C}_${No}_${Co}.txt
done
