"""Wall time of fwd+bwd of ONE stage of the AVE stack (its layers only), real AdapterStack scheduling, no profiler.
usage: python tools/stage_wall.py [B]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgsct_amd  # noqa
from dgsct_amd import AdapterStack, ave_stage_shapes
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0"); dt = torch.bfloat16
tot = 0.0
for i, st in enumerate(ave_stage_shapes("swinv2_base")):
    torch.manual_seed(0)
    stack = AdapterStack([st], compute_dtype=dt).to(dev); stack.flatten_parameters()
    with torch.no_grad():
        for n, p in stack.named_parameters():
            if n.endswith("gate") or n.endswith("gate_av"): p.fill_(0.5)
    feats, cots, mcots = bench.make_inputs([st], B * 10, dt, dev, 1)
    def step():
        outs, maps = stack(feats)
        tensors = [t for pair in outs for t in pair] + [maps[0], maps[1]]
        grads = [g for pair in cots for g in pair] + [mcots[0], mcots[1]]
        torch.autograd.backward(tensors, grads)
        for fv, fa in feats: fv.grad = None; fa.grad = None
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 10
    for _ in range(K): step()
    host = (time.perf_counter() - t0) / K * 1e3
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / K * 1e3
    pairs = st["layers"] * 2
    print(f"stage {i}: N={st['Nv']}/{st['Na']}  {ms:7.2f} ms  host {host:6.2f} ms  ({ms/pairs*1e3:7.0f} us per adapter pair, {pairs} pairs)")
    tot += ms
print(f"sum {tot:.2f} ms")
