# non-DP step vs the 1-rank RCCL DP step (overlap default) vs --no-overlap, same box
j() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('host_ms_fwdbwd_allreduce_optim'))"; }
echo -n "no DP        "; python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | j
echo -n "DP overlap   "; python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 15 --warmup 4 --force-dp --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | j
echo -n "DP no-overlap"; python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 15 --warmup 4 --force-dp --no-overlap --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | j
echo -n "no DP        "; python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | j
