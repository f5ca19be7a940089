for v in -2 -1 0 1; do
  if [ $v = -1 ]; then unset DGSCT_GEMM_NFAST; else export DGSCT_GEMM_NFAST=$v; fi
  echo "NFAST=$v"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
done
unset DGSCT_GEMM_NFAST
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
