#!/bin/bash
# every shape of tools/gemm_bench.py under the tiled engine (DGSCT_GEMM8=0), the default gates (1) and "every eligible shape" (2)
cd $GRAFT_REPO_ROOT
for m in 0 1 2; do DGSCT_GEMM8=$m python tools/gemm_bench.py child 2>/dev/null | grep RESULT | sed "s/RESULT/RESULT$m/"; done > /tmp/ab_all.txt
python - <<'PY'
import json, re
src = open("tools/gemm_bench.py").read()
ns = {}; exec("SHAPES = [" + src.split("SHAPES = [", 1)[1].split("\n]\n", 1)[0] + "\n]", ns)
rows = {l[6]: json.loads(l.split(" ", 1)[1]) for l in open("/tmp/ab_all.txt")}
print("shape (M,N,K,KB,batch,ak,bk,shared,atomic)".ljust(52) + "   tiled   gates     all   (us)")
for i, s in enumerate(ns["SHAPES"]):
    print(str(s[:9]).ljust(52) + "".join(f"{rows[m][i]:8.1f}" for m in "012"))
PY
