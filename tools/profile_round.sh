# Regenerates the measured artefacts of a round on the GPU box:  bash tools/profile_round.sh r02
tag=${1:-r04}
python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${tag}_bench.json
bash tools/pmc_stack.sh 2>&1 | tail -1
python tools/pmc_stack_summary.py $tag gpurun_out/${tag}_bench.json
cp profiles/${tag}_pmc_traffic.* profiles/${tag}_sq_counters.txt gpurun_out/ 2>/dev/null; rm -rf gpurun_out/pmc
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kst_$tag && rocprofv3 --kernel-trace --stats -d /tmp/kst_$tag -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/b_$tag.log 2>&1; python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kst_$tag -name "*.db" | head -1) 70 $GRAFT_REPO_ROOT/profiles/${tag}_gemm_in_step.json 7 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_kernel_stats.txt; cp $GRAFT_REPO_ROOT/profiles/${tag}_gemm_in_step.json $GRAFT_REPO_ROOT/gpurun_out/ )
python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${tag}_bench.json        # again: now with this round's PMC traffic
python bench.py --steps 10 --warmup 3 --dtype fp32 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${tag}_bench_fp32.json
python bench.py --steps 10 --warmup 3 --backbone swinv2_large --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${tag}_bench_swinL.json
python tools/attn_bench.py > gpurun_out/${tag}_attn_bench.txt 2>&1
bash tools/seq_stack.sh > /dev/null 2>&1; python tools/seq_summary.py gpurun_out/seq > gpurun_out/${tag}_seq_summary.txt
for f in bench bench_fp32 bench_swinL; do python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_$f.json")); r=d.get("roofline") or {}
print("$f", d["ms_per_step"], "ms/step", d["value"], "clips/s  gemm frac", r.get("frac"), " step", r.get("step"))
PY
done
