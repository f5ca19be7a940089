"""SURVEY.md 8(f) row f4 (round 5, first step): the frozen blocks either side of the adapter calls (dg-sct_amd/backbone.py).

* `HTSATBlock` against the reference class (`DG-SCT/AVE/nets/htsat.py:135-251`) through the committed fixture
  `tests/golden/htsat_block.pt` (oracle/make_golden_backbone.py: the restatement matches the imported reference bit for bit on CPU).
* `SwinV2Block` restates timm 0.6.12 (not vendored, not installed): parity UNPINNED.  What can be checked without timm is checked:
  the window partition / cyclic shift / shift mask / relative-offset plumbing against a direct per-token evaluation of the published
  formulas over the whole map (every token against every token, membership and offsets computed from coordinates).
* On the GPU: the blocks as `AdapterStack(vis_block=, aud_block=)` callables in bf16 -- gradients flow through the frozen blocks to
  the adapters below, their own parameters get none."""
import math
import os

import pytest
import torch

from helpers import ROOT
from dgsct_amd.backbone import FrozenBlocks, HTSATBlock, SwinV2Block


def _cases():
    return torch.load(os.path.join(ROOT, "tests", "golden", "htsat_block.pt"), weights_only=False)


@pytest.mark.parametrize("name", ["plain", "shifted", "one_window"])
def test_htsat_block_matches_reference_fixture(name):
    fx = _cases()[name]
    dim, res, heads, ws, shift = fx["cfg"]
    blk = HTSATBlock(dim, (res, res), heads, window_size=ws, shift_size=shift).eval()
    blk.load_state_dict(fx["state"])
    x = fx["x"].clone().requires_grad_(True)
    y, attn = blk(x)
    y.backward(fx["cot"])
    assert (y - fx["y"]).abs().max() < 1e-5 and (x.grad - fx["dx"]).abs().max() < 1e-5
    assert attn.shape == (3 * (res // blk.window_size) ** 2, heads, blk.window_size ** 2, blk.window_size ** 2)


def _naive_swinv2_attn(blk: SwinV2Block, x: torch.Tensor) -> torch.Tensor:
    """`blk._attn(x)` evaluated token against token over the whole map: two tokens interact iff the cyclic shift puts them into the same
    window; inside a window, tokens that the shift brought together from different sides of the map border get -100 on their logit;
    the position bias is 16 sigmoid(cpb_mlp(log-spaced (dy, dx))) of their offset in the shifted frame (Liu et al. 2022, eqs. 3-4)."""
    H, W = blk.input_resolution
    ws, s, h = blk.window_size, blk.shift_size, blk.num_heads
    a = blk.attn
    B, L, C = x.shape
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    sy, sx = ((ys - s) % H).reshape(-1), ((xs - s) % W).reshape(-1)            # coordinates after the roll by -s
    same_win = (sy[:, None] // ws == sy[None, :] // ws) & (sx[:, None] // ws == sx[None, :] // ws)
    reg = lambda c, n: torch.where(c < n - ws, 0, torch.where(c < n - s, 1, 2)) if s else torch.zeros_like(c)
    region = reg(sy, H) * 3 + reg(sx, W)
    masked = region[:, None] != region[None, :]
    bias3 = torch.cat([a.q_bias, torch.zeros_like(a.v_bias), a.v_bias])
    q, k, v = torch.nn.functional.linear(x, a.qkv.weight, bias3).reshape(B, L, 3, h, C // h).permute(2, 0, 3, 1, 4)
    qn, kn = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12), k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    logits = (qn @ kn.transpose(-2, -1)) * torch.clamp(a.logit_scale, max=math.log(100.0)).exp()
    dy, dx = (sy[:, None] - sy[None, :]).float(), (sx[:, None] - sx[None, :]).float()
    off = torch.stack([dy, dx], -1) / (ws - 1) * 8
    off = torch.sign(off) * torch.log2(off.abs() + 1) / math.log2(8)
    bias = 16 * torch.sigmoid(a.cpb_mlp(off)).permute(2, 0, 1)                  # [h, L, L]
    logits = logits + bias[None] + torch.where(masked, -100.0, 0.0)[None, None]
    logits = logits.masked_fill(~same_win[None, None], float("-inf"))
    return a.proj((logits.softmax(-1) @ v).transpose(1, 2).reshape(B, L, C))


@pytest.mark.parametrize("res,ws,shift", [(24, 12, 0), (24, 12, 6), (12, 12, 6), (16, 8, 4)])
def test_swinv2_block_windowing_against_a_per_token_evaluation(res, ws, shift):
    torch.manual_seed(5)
    blk = SwinV2Block(32, (res, res), 2, window_size=ws, shift_size=shift).eval()
    with torch.no_grad():
        for p in blk.parameters():
            p.add_(0.2 * torch.randn_like(p))
    x = torch.randn(2, res * res, 32)
    with torch.no_grad():
        got, ref = blk._attn(x), _naive_swinv2_attn(blk, x)
    assert (got - ref).abs().max() < 2e-5, (got - ref).abs().max()
    if res <= ws:
        assert blk.shift_size == 0 and blk.attn_mask is None                 # a window as large as the map: no shift (stages 2-3 at 192^2)


def test_frozen_blocks_are_frozen_and_shaped_for_the_ave_stack():
    from dgsct_amd import ave_stage_shapes
    fb = FrozenBlocks(ave_stage_shapes("swinv2_base"), dtype=torch.float32)
    assert len(fb.vis) == len(fb.aud) == 12 and not any(p.requires_grad for p in fb.parameters())
    assert [b.window_size for b in fb.vis] == [12, 12, 12, 12] + [12] * 6 + [6, 6]
    assert [b.shift_size for b in fb.vis][:4] == [0, 6, 0, 6] and all(b.shift_size == 0 for b in fb.vis[4:])     # stages 2-3: one window
    assert [b.shift_size for b in fb.aud] == [0, 4, 0, 4, 0, 4, 0, 4, 0, 4, 0, 0]                               # 8 x 8 map at stage 3
    f_a = torch.randn(2, 64, 768, requires_grad=True)
    y = fb.aud_block(11, f_a)
    y.sum().backward()
    assert f_a.grad is not None and y.shape == f_a.shape


@pytest.mark.gpu
def test_adapter_stack_with_frozen_blocks_on_gpu():
    """harness B of SURVEY.md 8(d): two stages of the AVE stack with the frozen half-blocks / blocks in the loop, bf16, fwd + bwd"""
    from dgsct_amd import AdapterStack
    dev = torch.device("cuda:0")
    stages = [dict(layers=1, Nv=576, Cv=256, Na=1024, Ca=192), dict(layers=2, Nv=144, Cv=512, Na=256, Ca=384)]
    torch.manual_seed(0)
    st = AdapterStack(stages, compute_dtype=torch.bfloat16).to(dev)
    with torch.no_grad():
        for n, p in st.named_parameters():
            if n.endswith("gate") or n.endswith("gate_av"):
                p.fill_(0.5)
    st.flatten_parameters().train()
    fb = FrozenBlocks(stages, dtype=torch.bfloat16).to(dev)
    feats = [(torch.randn(4, s["Nv"], s["Cv"], device=dev, dtype=torch.bfloat16, requires_grad=True),
              torch.randn(4, s["Na"], s["Ca"], device=dev, dtype=torch.bfloat16, requires_grad=True)) for s in stages]
    outs, maps = st(feats, vis_block=fb.vis_block, aud_block=fb.aud_block)
    plain, _ = st([(a.detach(), b.detach()) for a, b in feats])
    torch.autograd.backward([t for pr in outs for t in pr], [torch.randn_like(t) for pr in outs for t in pr])
    torch.cuda.synchronize()
    for (fv, fa), (pv, pa) in zip(outs, plain):
        assert torch.isfinite(fv.float()).all() and torch.isfinite(fa.float()).all()
        assert (fv.float() - pv.float()).abs().max() > 1e-2                  # the blocks are in the loop
    assert all(torch.isfinite(f.grad.float()).all() and f.grad.abs().sum() > 0 for pr in feats for f in pr)
    assert all(p.grad is None for p in fb.parameters())
    assert all(torch.isfinite(m.flat_param.grad).all() for m in st.modules() if hasattr(m, "flat_param"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["plain", "shifted"])
def test_htsat_block_bf16_on_gpu(name):
    fx = _cases()[name]
    dim, res, heads, ws, shift = fx["cfg"]
    dev = torch.device("cuda:0")
    blk = HTSATBlock(dim, (res, res), heads, window_size=ws, shift_size=shift).eval()
    blk.load_state_dict(fx["state"])
    blk = blk.to(dev, torch.bfloat16)
    y, _ = blk(fx["x"].to(dev, torch.bfloat16))
    assert ((y.float().cpu() - fx["y"]).norm() / fx["y"].norm()) < 2e-2
