# bench.py (no CPU baseline / roofline legs) N times on the current tree: bash tools/ab_bench.sh [runs] [extra bench flags]
cd $GRAFT_REPO_ROOT
R=${1:-3}; shift
for i in $(seq $R); do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python tools/last_json.py; done
