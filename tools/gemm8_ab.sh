#!/bin/bash
# A/B the 8-wave deep-product kernel (gemm8.hip) against the tiled engine on the big shapes of tools/gemm_bench.py,
# tile-order sweep (DGSCT_GEMM8_GM), then one SQ counter pass over the same launches.
cd $GRAFT_REPO_ROOT
echo "== tiled engine"; DGSCT_GEMM8=0 BIG=1 python tools/gemm_bench.py child 2>/dev/null | grep RESULT
for gm in ${GMS:-16 4 2}; do echo "== gemm8 gm=$gm"; DGSCT_GEMM8_GM=$gm BIG=1 python tools/gemm_bench.py child 2>/dev/null | grep RESULT; done
if [ -n "${PMC:-}" ]; then
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/g8pmc
  BIG=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS -d /tmp/g8pmc -o p -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py child > /tmp/g8pmc.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find /tmp/g8pmc -name "*.db" | head -1) gemm
fi
