"""Per-shape timing of the frozen blocks' kernels at the AVE stack's geometries (B = 160 frames, bf16): the fused window attention
(csrc/wattn.hip) forward / backward and the LayerNorm (+ residual) row kernels (dgsct_layer_norm_*), against their algorithmic bytes.
  window attention forward : qkv read once + O written          = B L (3C + C) 2 bytes
  window attention backward: qkv, O, dO read + dqkv written     = B L (3C + C + C + 3C) 2 bytes
  LayerNorm forward (+res) : x (+ residual) read, out written   = rows C (2 | 3) 2 bytes;   backward: dout, x read, dx written = rows C 3 x 2
usage: python tools/wattn_bench.py [swinv2_base|swinv2_large]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgsct_amd  # noqa: E402,F401
from dgsct_amd import ave_stage_shapes, ops  # noqa: E402
from dgsct_amd.backbone import _HTSAT_HEADS, _SWIN_HEADS  # noqa: E402

backbone = sys.argv[1] if len(sys.argv) > 1 else "swinv2_base"
dev = torch.device("cuda:0")
B = 160


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


print(f"{'block':>8} {'map':>5} {'C':>5} {'ws':>3} {'heads':>5} {'cos':>3}   {'wattn fwd':>20}   {'wattn bwd':>20}   {'LN fwd (+res)':>20}   {'LN bwd':>20}")
tot = [0.0] * 4
for s in ave_stage_shapes(backbone):
    for kind, N, C, ws0, heads in (("swin-v2", s["Nv"], s["Cv"], 12, _SWIN_HEADS[s["Cv"]]), ("hts-at", s["Na"], s["Ca"], 8, _HTSAT_HEADS[s["Ca"]])):
        R = int(round(math.sqrt(N)))
        ws = min(ws0, R)
        shift = ws // 2 if (R > ws and os.environ.get("NOSHIFT") is None) else 0        # NOSHIFT=1: the un-shifted blocks (one bias table for all windows)
        n, nW = ws * ws, (R // ws) ** 2
        cosine = kind == "swin-v2"
        gen = torch.Generator().manual_seed(1)
        qkv = torch.randn(B, N, 3 * C, generator=gen).to(dev, torch.bfloat16).requires_grad_(True)
        bm = torch.randn(nW if shift else 1, heads, n, n, generator=gen).to(dev)
        scale = (torch.rand(heads, generator=gen) + 0.5).to(dev)
        dout = torch.randn(B, N, C, generator=gen).to(dev, torch.bfloat16)
        o = ops.window_attention(qkv, bm, scale, R, R, ws, shift, heads, None, cosine)
        t_f = timed(lambda: ops.window_attention(qkv.detach(), bm, scale, R, R, ws, shift, heads, None, cosine))
        t_b = timed(lambda: torch.autograd.grad(o, qkv, dout, retain_graph=True))
        x = torch.randn(B * N, C, generator=gen).to(dev, torch.bfloat16).requires_grad_(True)
        res = torch.randn(B * N, C, generator=gen).to(dev, torch.bfloat16) if cosine else None      # post-norm (Swin-V2): residual fused
        w, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        y = ops.layer_norm(x, w, b_, 1e-5, res)
        g = torch.randn_like(y)
        t_lf = timed(lambda: ops.layer_norm(x.detach(), w, b_, 1e-5, res))
        t_lb = timed(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
        by = [B * N * 4 * C * 2, B * N * 8 * C * 2, B * N * C * 2 * (3 if cosine else 2), B * N * C * 2 * 3]
        cells = [f"{t:8.1f} us {bb / t / 1e3:6.0f} GB/s" for t, bb in zip((t_f, t_b, t_lf, t_lb), by)]
        print(f"{kind:>8} {R:3d}^2 {C:5d} {ws:3d} {heads:5d} {int(cosine):3d}   " + "   ".join(f"{c:>20}" for c in cells))
        for i, t in enumerate((t_f, t_b, t_lf, t_lb)):
            tot[i] += t * s["layers"] * (1 if i < 2 else 2)           # one attention, two LayerNorms per block
print("per step (24 blocks), ms: wattn fwd %.2f  bwd %.2f   LayerNorm fwd %.2f  bwd %.2f   (HBM roof: 8000 GB/s; backward timings include one autograd node's "
      "host launch)" % tuple(t / 1e3 for t in tot))
