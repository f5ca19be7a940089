"""Steady-state cost of one layer (2 adapter pairs) of each stage: full AVE stack vs the stack with 2 extra layers of
that stage; no profiler.  usage: python tools/stage_delta.py [B]"""
import os, sys, time, copy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgsct_amd  # noqa
from dgsct_amd import AdapterStack, ave_stage_shapes
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0"); dt = torch.bfloat16

def run(stages):
    torch.manual_seed(0)
    stack = AdapterStack(stages, compute_dtype=dt).to(dev); stack.flatten_parameters()
    with torch.no_grad():
        for n, p in stack.named_parameters():
            if n.endswith("gate") or n.endswith("gate_av"): p.fill_(0.5)
    feats, cots, mcots = bench.make_inputs(stages, B * 10, dt, dev, 1)
    def step():
        outs, maps = stack(feats)
        tensors = [t for pair in outs for t in pair] + [maps[0], maps[1]]
        grads = [g for pair in cots for g in pair] + [mcots[0], mcots[1]]
        torch.autograd.backward(tensors, grads)
        for fv, fa in feats: fv.grad = None; fa.grad = None
    for _ in range(3): step()
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): step()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 8 * 1e3)
    del stack
    return best

base_st = ave_stage_shapes("swinv2_base")
base = run(base_st)
print(f"full stack fwd+bwd: {base:.2f} ms")
tot = 0
for i in range(4):
    st = copy.deepcopy(base_st); st[i]["layers"] += 2
    t = run(st)
    per_layer = (t - base) / 2
    tot += per_layer * base_st[i]["layers"]
    print(f"stage {i}: +2 layers -> {t:.2f} ms; per layer {per_layer:.2f} ms; stage total {per_layer*base_st[i]['layers']:.2f} ms")
print(f"sum of stage totals {tot:.2f} ms")
