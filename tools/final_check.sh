cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -i -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r03_bench.json
python bench.py --steps 10 --warmup 3 --backbone swinv2_large --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r03_bench_swinL.json
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kst_f && rocprofv3 --kernel-trace --stats -d /tmp/kst_f -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /tmp/b_f.log 2>&1; python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kst_f -name "*.db" | head -1) 60 > $GRAFT_REPO_ROOT/gpurun_out/r03_bench_kernel_stats.txt )
python - <<'PY'
import json
for f in ("bench","bench_swinL"):
    d=json.load(open(f"gpurun_out/r03_{f}.json")); r=d.get("roofline") or {}
    print(f, d["ms_per_step"], d["value"], r.get("frac"), r.get("achieved"), r.get("gemm_ms_per_step"))
PY
