"""Per-shape timing of the fused latent-token attention kernels at the AVE stack's 8 adapter shapes (B = 160 frames).
usage: python tools/attn_bench.py [swinv2_base|swinv2_large] [bf16|fp32]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dgsct_amd  # noqa: E402
from attn_ref import AttnCall  # noqa: E402
from dgsct_amd import ave_stage_shapes  # noqa: E402
from dgsct_amd._lib import default_lib  # noqa: E402

backbone = sys.argv[1] if len(sys.argv) > 1 else "swinv2_base"
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "fp32") else torch.bfloat16
es = 2 if dtype == torch.bfloat16 else 4
dev = torch.device("cuda:0")
lib = default_lib()
if os.environ.get("TFS8") is not None:          # A/B: 0 = the 4-wave short-frame tokattn_fwd kernel
    lib.test_tune("tfs8", int(os.environ["TFS8"]))
names = ["tokattn_fwd(+combine,pack)", "xattn_fwd", "xattn_bwd", "tokattn_bwd(+pack)"]
units = [1, 2, 3, 2]            # algorithmic passes over a [B][N][C] activation: read Yp | X->X1 | X,dX1->dX | Yp->dYp
tot = [0.0] * 4
print(f"{'N':>5} {'C':>5} " + " ".join(f"{n:>22}" for n in names))
for s in ave_stage_shapes(backbone):
    for (N, C) in ((s["Nv"], s["Cv"]), (s["Na"], s["Ca"])):
        c = AttnCall(lib, dtype, 160, N, C, 32, dev)
        c.run(4)
        row = []
        for op in range(4):
            for _ in range(3):
                c.run(op)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                c.run(op)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            gbs = units[op] * 160 * N * C * es / (us * 1e-6) / 1e9
            row.append(f"{us:9.1f} us {gbs:7.0f} GB/s")
            tot[op] += us * 2 * s["layers"]
        print(f"{N:5d} {C:5d} " + " ".join(f"{r:>22}" for r in row))
print("per step (48 adapter calls), ms: " + "  ".join(f"{n} {t / 1e3:.2f}" for n, t in zip(names, tot)) + f"   total {sum(tot) / 1e3:.2f}")
