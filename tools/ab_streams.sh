run() { echo -n "$1  "; env $1 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
run X=0; run DGSCT_NO_AUX=1; run DGSCT_AUX_PRIORITY=0; run DGSCT_AUX_PRIORITY=-1; run DGSCT_COMPUTE_PRIORITY=0; run DGSCT_COMPUTE_PRIORITY=1; run X=0
