#!/bin/bash
# PMC passes over ONE forward+backward (x2 identical iterations) of each of the 8 distinct adapter shapes of the AVE
# Swin-V2-B stack, one counter group per run (kernel-trace + pmc only, as the pool requires):
#   FETCH_SIZE | WRITE_SIZE | SQ (MFMA busy, LDS bank conflicts, wave cycles)
# A whole bench step under --pmc serialises ~9000 dispatches (tens of minutes); per-shape passes take seconds and are
# scaled by the schedule (tools/pmc_stack_summary.py), whose launch counts are checked against the bench's own.
set -u
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
SHAPES="${SHAPES:-2304,128,4096,96 4096,96,2304,128 576,256,1024,192 1024,192,576,256 144,512,256,384 256,384,144,512 36,1024,64,768 64,768,36,1024}"
for s in $SHAPES; do
  IFS=, read N C No Co <<< "$s"
  for grp in FETCH_SIZE WRITE_SIZE SQ; do
    ctrs=$grp
    [ $grp = SQ ] && ctrs="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
    timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d $OUT/${N}_${C}_${No}_${Co}_$grp -o p -- \
      python $GRAFT_REPO_ROOT/tools/trace_adapter.py $N $C $No $Co 160 > $OUT/log_${N}_$grp.txt 2>&1
  done
done
ls $OUT | wc -l
