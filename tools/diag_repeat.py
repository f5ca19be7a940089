import os
"""diagnostic (not a test): repeat one full-size forward+backward many times and report the largest run-to-run deviation
(atomic summation order alone gives ~1e-6; anything larger is a race)"""
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import param_table, spec_of
from dgsct_amd import ops
from dgsct_amd._lib import PARAM_NAMES, default_lib
from oracle import dgsct_oracle as O

DEV = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
DT = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "fp32" else torch.bfloat16
for shape in [(144, 512, 256, 384), (36, 1024, 64, 768), (2304, 128, 4096, 96)]:
    N, C, No, Co = shape
    cfg = O.AdapterConfig(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2)
    p = O.random_params(cfg, "ave", seed=31, scale=0.577)
    spec = spec_of(cfg)
    lib = default_lib()
    gen = torch.Generator().manual_seed(32)
    dt = DT
    X = torch.randn(160, N, C, generator=gen).to(DEV, dt)
    Y = torch.randn(160, No, Co, generator=gen).to(DEV, dt)
    g = torch.randn(160, N, C, generator=gen).to(DEV, dt)
    m = torch.randn(160, N, generator=gen).to(DEV)
    params = param_table(p, spec, DEV)
    first = None
    worst = 0.0
    bad, nbad = {}, {}
    for rep in range(reps):
        prep = ops.prepare(lib, spec, params, dt, DEV)
        out, amap, _, saved, d = ops.raw_forward(lib, spec, params, prep, X, Y, True)
        dX, dY, grads = ops.raw_backward(lib, spec, d, params, prep, X, Y, saved, g, m, None)
        torch.cuda.synchronize()
        cur = [out.float(), amap, dX.float(), dY.float()] + [x for x in grads if x is not None and x.numel() > 64]
        if first is None:
            first = [c.clone() for c in cur]
            continue
        for i, (a, b) in enumerate(zip(cur, first)):
            dev = ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
            if dev > worst:
                worst = dev
            if dev > (1e-3 if DT == torch.bfloat16 else 1e-5) or dev != dev:
                bad[i] = max(bad.get(i, 0.0), dev); nbad[i] = nbad.get(i, 0) + 1
    print(shape, str(DT), "worst run-to-run deviation over", reps, "reps: %.3e" % worst, "| tensors over threshold (idx: count, max):",
          {i: (nbad[i], "%.1e" % bad[i]) for i in sorted(bad)}, flush=True)
