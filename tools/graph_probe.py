"""Is HIP-graph replay of the whole step host-cheap on this ROCm?  Times replay() (host) vs wall."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
serial = "--serial" in sys.argv
stages, stack = bench.build_stack("swinv2_base", torch.bfloat16, dev, concurrent=not serial)
stack.train()
feats, cots, mcots = bench.make_inputs(stages, 160, torch.bfloat16, dev, 1)
params = [p for p in stack.parameters()]
def fwd_bwd():
    outs, maps = stack(feats)
    tensors = [t for pair in outs for t in pair] + [maps[0], maps[1]]
    grads = [g for pair in cots for g in pair] + [mcots[0], mcots[1]]
    torch.autograd.backward(tensors, grads)
    for p in params: p.grad = None
    for fv, fa in feats: fv.grad = None; fa.grad = None
s = torch.cuda.Stream(device=dev)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): fwd_bwd()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): fwd_bwd()
torch.cuda.synchronize()
for _ in range(2): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter(); g.replay(); h = time.perf_counter() - t0; torch.cuda.synchronize(); w = time.perf_counter() - t0
print(f"serial={serial}: one replay: host {h*1e3:.2f} ms, wall {w*1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(5): g.replay()
h = time.perf_counter() - t0; torch.cuda.synchronize(); w = time.perf_counter() - t0
print(f"5 replays: host {h/5*1e3:.2f} ms each, wall {w/5*1e3:.2f} ms each")
