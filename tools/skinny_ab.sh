#!/bin/bash
# the [BT, C] gate-MLP products on the tiled engine (DGSCT_GEMM_SKINNY=0) and on the K-split kernel (1), then the whole step
cd $GRAFT_REPO_ROOT
for m in 0 1; do echo "DGSCT_GEMM_SKINNY=$m"; DGSCT_GEMM_SKINNY=$m python - <<'PY'
src=open("tools/gemm_bench.py").read().split('if __name__ == "__main__":')[0]
ns={"__file__":"tools/gemm_bench.py"}; exec(src, ns)
for s in [(160,1024,1024,1,1,1,1,0,0,1,0),(160,512,1024,1,1,1,1,0,0,1,0),(160,512,512,1,1,1,1,0,0,1,0),(160,256,512,1,1,1,1,0,0,1,0),(160,512,256,1,1,1,1,0,0,0,0),(160,384,192,1,1,1,1,0,0,1,0),(160,128,128,1,1,1,1,0,0,1,0),(160,1536,1536,1,1,1,1,0,0,1,0)]:
    print(s[:3], round(ns["run"](s, 30),1))
PY
DGSCT_GEMM_SKINNY=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"ms_per_step\"])"
done
