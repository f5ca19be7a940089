bash tools/pmc_stack.sh 2>&1 | tail -1
python tools/pmc_stack_summary.py | head -4
cp profiles/r01_pmc_traffic.* gpurun_out/
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kst -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /tmp/b.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kst -name "*.db" | head -1) 45 > $GRAFT_REPO_ROOT/gpurun_out/r01_bench_kernel_stats.txt
cd $GRAFT_REPO_ROOT
python bench.py 2>&1 | grep "^{" | tail -1 > gpurun_out/bench_final.json
cut -c1-200 gpurun_out/bench_final.json
python -c "
import json
d=json.load(open('gpurun_out/bench_final.json')); print(json.dumps(d['roofline'])[:1000]); print(d['cpu_baseline']['value'])"
