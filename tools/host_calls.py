"""Host-side time line of the library calls of one eager step: when each dgsct_adapter_forward / _backward call is entered and
left (perf_counter on the calling thread), grouped by adapter shape.  Answers: is a late-stage pair bound by the host enqueue of
its two calls (one thread, one after the other) or by the GPU chains?
usage: python tools/host_calls.py [BT]"""
import os, sys, time, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgsct_amd
from dgsct_amd import _lib
import bench
log = []
for name in ("forward", "backward"):
    orig = getattr(_lib.Lib, name)
    def wrap(self, desc, *a, _o=orig, _n=name, **k):
        t = time.perf_counter(); r = _o(self, desc, *a, **k); t1 = time.perf_counter()
        log.append((_n, int(desc.N), int(desc.C), t, t1, threading.get_ident())); return r
    setattr(_lib.Lib, name, wrap)
dev = torch.device("cuda:0")
stages, stack = bench.build_stack("swinv2_base", torch.bfloat16, dev, concurrent=True)
stack.train()
BT = int(sys.argv[1]) if len(sys.argv) > 1 else 160
feats, cots, mcots = bench.make_inputs(stages, BT, torch.bfloat16, dev, 1)
params = [p for p in stack.parameters()]
def step():
    outs, maps = stack(feats)
    tensors = [t for pair in outs for t in pair] + [maps[0], maps[1]]
    grads = [g for pair in cots for g in pair] + [mcots[0], mcots[1]]
    torch.autograd.backward(tensors, grads)
    for p in params: p.grad = None
    for fv, fa in feats: fv.grad = None; fa.grad = None
for _ in range(3): step()
torch.cuda.synchronize()
log.clear()
t0 = time.perf_counter()
step()
th = time.perf_counter()
torch.cuda.synchronize()
te = time.perf_counter()
print(f"step: host {1e3*(th-t0):.2f} ms, wall {1e3*(te-t0):.2f} ms, {len(log)} library calls, threads {len(set(l[5] for l in log))}")
agg = {}
for i, (n, N, C_, a, b, tid) in enumerate(log):
    gap = (a - log[i - 1][4]) if i else 0.0
    k = (n, N, C_)
    e = agg.setdefault(k, [0, 0.0, 0.0]); e[0] += 1; e[1] += b - a; e[2] += gap
print(f"{'call':9s} {'N':>5s} {'C':>5s} {'calls':>5s} {'in-call us':>11s} {'gap before us':>14s}")
for (n, N, C_), (c, d, g) in agg.items():
    print(f"{n:9s} {N:5d} {C_:5d} {c:5d} {1e6*d/c:11.1f} {1e6*g/c:14.1f}")
tot_in = sum(b - a for (_, _, _, a, b, _) in log)
print(f"in library calls {1e3*tot_in:.2f} ms; between them (python / torch / autograd) {1e3*((th-t0)-tot_in):.2f} ms")
