# usage: bash tools/ab_env.sh VAR   -> bench with and without VAR=1
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export $1=1; else unset $1; fi
  echo -n "$1=$v  "; python bench.py --steps 15 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
done
