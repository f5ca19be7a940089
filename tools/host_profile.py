"""Where does the host time of an eager step go?  (ctypes library calls vs. the rest of Python/PyTorch)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgsct_amd
from dgsct_amd import _lib
import bench
acc = {"forward": 0.0, "backward": 0.0, "prepare": 0.0, "query": 0.0}
cnt = {k: 0 for k in acc}
for name in acc:
    orig = getattr(_lib.Lib, name)
    def wrap(self, *a, _o=orig, _n=name, **k):
        t = time.perf_counter(); r = _o(self, *a, **k); acc[_n] += time.perf_counter() - t; cnt[_n] += 1; return r
    setattr(_lib.Lib, name, wrap)
dev = torch.device("cuda:0")
stages, stack = bench.build_stack("swinv2_base", torch.bfloat16, dev, concurrent="--serial" not in sys.argv)
stack.train()
BT = int(os.environ.get("BT", "160"))
feats, cots, mcots = bench.make_inputs(stages, BT, torch.bfloat16, dev, 1)
params = [p for p in stack.parameters()]
def step():
    outs, maps = stack(feats)
    tensors = [t for pair in outs for t in pair] + [maps[0], maps[1]]
    grads = [g for pair in cots for g in pair] + [mcots[0], mcots[1]]
    torch.autograd.backward(tensors, grads)
    for p in params: p.grad = None
    for fv, fa in feats: fv.grad = None; fa.grad = None
for _ in range(2): step()
torch.cuda.synchronize()
for k in acc: acc[k] = 0.0; cnt[k] = 0
t0 = time.perf_counter()
n = 3
for _ in range(n): step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"per step: wall {tot/n*1e3:.1f} ms, host enqueue {host/n*1e3:.1f} ms")
for k in acc: print(f"  lib.{k:9s} {acc[k]/n*1e3:7.2f} ms/step  ({cnt[k]//n} calls/step, {acc[k]/max(cnt[k],1)*1e6:.0f} us/call)")
print(f"  other python/torch {(host-sum(acc.values()))/n*1e3:.1f} ms/step")
