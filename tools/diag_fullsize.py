import os
"""diagnostic (not a test): margins of test_full_size_bf16_path_matches_fp32_path over repeated runs"""
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import param_table, spec_of
from dgsct_amd import ops
from dgsct_amd._lib import PARAM_NAMES, default_lib
from oracle import dgsct_oracle as O

DEV = torch.device("cuda", 0)


def l2(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def run(shape, dt, p, X, Y, g, m):
    N, C, No, Co = shape
    cfg = O.AdapterConfig(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2)
    spec = spec_of(cfg)
    lib = default_lib()
    params = param_table({k: v.clone() for k, v in p.items()}, spec, DEV)
    prep = ops.prepare(lib, spec, params, dt, DEV)
    Xd, Yd = X.to(DEV, dt), Y.to(DEV, dt)
    out, amap, _, saved, d = ops.raw_forward(lib, spec, params, prep, Xd, Yd, True)
    dX, dY, grads = ops.raw_backward(lib, spec, d, params, prep, Xd, Yd, saved, g.to(DEV, dt), m.to(DEV), None)
    torch.cuda.synchronize()
    return out.float().cpu(), amap.cpu(), dX.float().cpu(), dY.float().cpu(), [x.cpu() if x is not None else None for x in grads]


for shape in [(144, 512, 256, 384), (2304, 128, 4096, 96)]:
    N, C, No, Co = shape
    cfg = O.AdapterConfig(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2)
    p = O.random_params(cfg, "ave", seed=31, scale=0.577)
    gen = torch.Generator().manual_seed(32)
    X = torch.randn(160, N, C, generator=gen).bfloat16().float()
    Y = torch.randn(160, No, Co, generator=gen).bfloat16().float()
    g = torch.randn(160, N, C, generator=gen).bfloat16().float()
    m = torch.randn(160, N, generator=gen)
    f = run(shape, torch.float32, p, X, Y, g, m)
    for rep in range(4):
        h = run(shape, torch.bfloat16, p, X, Y, g, m)
        worst = max((l2(a, b), PARAM_NAMES[i]) for i, (a, b) in enumerate(zip(h[4], f[4])) if b is not None and b.dim() > 0 and b.numel() > 4096)
        print(shape, "out %.4f map %.4f dX %.4f dY %.4f worst-W %.4f %s" % (l2(h[0], f[0]), l2(h[1], f[1]), l2(h[2], f[2]), l2(h[3], f[3]), worst[0], worst[1]), flush=True)
