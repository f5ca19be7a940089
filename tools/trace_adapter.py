"""Run ONE adapter forward+backward (twice) at a given shape; meant to be wrapped by rocprofv3 --kernel-trace.
usage: python tools/trace_adapter.py N C No Co [BT] [dtype]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dgsct_amd  # noqa
from dgsct_amd import ops
from dgsct_amd._lib import default_lib
from helpers import param_table, spec_of
from oracle import dgsct_oracle as O

N, C, No, Co = [int(x) for x in sys.argv[1:5]]
BT = int(sys.argv[5]) if len(sys.argv) > 5 else 160
dtype = torch.float32 if len(sys.argv) > 6 and sys.argv[6] == "fp32" else torch.bfloat16
dev = torch.device("cuda:0")
cfg = O.AdapterConfig(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2)
p = O.random_params(cfg, "ave", seed=0, scale=0.577)
spec = spec_of(cfg)
params = param_table(p, spec, dev)
lib = default_lib()
X = torch.randn(BT, N, C, device=dev).to(dtype)
Y = torch.randn(BT, No, Co, device=dev).to(dtype)
dOut = torch.randn(BT, N, C, device=dev).to(dtype)
dMap = torch.randn(BT, N, device=dev)
for it in range(2):
    prep = ops.prepare(lib, spec, params, dtype, dev)
    out, amap, _, saved, d = ops.raw_forward(lib, spec, params, prep, X, Y, True)
    dX, dY, grads = ops.raw_backward(lib, spec, d, params, prep, X, Y, saved, dOut, dMap, None)
    torch.cuda.synchronize()
print("done")
