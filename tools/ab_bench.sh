# bench.py (no CPU baseline / roofline legs) under two git states of the working tree is not possible on the box; this runs the
# current tree N times: bash tools/ab_bench.sh [runs] [extra bench flags]
cd $GRAFT_REPO_ROOT
R=${1:-3}; shift
for i in $(seq $R); do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done
