"""GPU diagnostic: bf16 errors (relative L2) of every output / gradient at a real shape."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_adapter_gpu import _real_case, _l2
shape = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else [144, 512, 256, 384]
r = _real_case(*shape, BT=10, dtype=torch.bfloat16)
for k in ("out", "map", "dX", "dY"):
    print(f"{k:32s} l2 {_l2(*r[k]):.4f}")
for k, (g, go) in r["grads"].items():
    print(f"{k:32s} l2 {_l2(g, go.reshape(-1)):.4f}   |ref| {go.norm().item():.3e}")
