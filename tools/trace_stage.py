"""fwd+bwd of ONE stage of the AVE stack under the real AdapterStack schedule (for rocprofv3 --kernel-trace).
usage: python tools/trace_stage.py <stage 0..3> [B] [steps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgsct_amd  # noqa
from dgsct_amd import AdapterStack, ave_stage_shapes
import bench
si = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 16; K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0"); dt = torch.bfloat16
st = ave_stage_shapes("swinv2_base")[si]
torch.manual_seed(0)
stack = AdapterStack([st], compute_dtype=dt).to(dev); stack.flatten_parameters()
with torch.no_grad():
    for n, p in stack.named_parameters():
        if n.endswith("gate") or n.endswith("gate_av"): p.fill_(0.5)
feats, cots, mcots = bench.make_inputs([st], B * 10, dt, dev, 1)
params = list(stack.parameters())
def step():
    outs, maps = stack(feats)
    tensors = [t for pair in outs for t in pair] + [maps[0], maps[1]]
    grads = [g for pair in cots for g in pair] + [mcots[0], mcots[1]]
    torch.autograd.backward(tensors, grads)
    for p in params: p.grad = None
    for fv, fa in feats: fv.grad = None; fa.grad = None
import time
for i in range(K):
    if i == K - 1: torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
torch.cuda.synchronize()
print(f"stage {si}: last step wall {1e3*(time.perf_counter()-t0):.2f} ms")
