"""Frozen backbone blocks either side of the adapter calls.

The AVE layer loop (reference ``DG-SCT/AVE/nets/net_trans.py:880-916``) interleaves each adapter pair with

* the two halves of a timm Swin-V2 block: ``f_v = f_v + drop_path1(norm1(blk._attn(f_v)))`` (``:894``) and
  ``f_v = f_v + drop_path2(norm2(blk.mlp(f_v)))`` (``:903``) -- cosine window attention with a learned logit scale and a
  continuous relative position bias (an MLP over log-spaced offsets), POST-norm;
* one HTS-AT block ``f_a, _ = blk_a(f_a)`` (``:897``; class ``SwinTransformerBlock`` of ``DG-SCT/AVE/nets/htsat.py:135-251``):
  Swin-v1 -- scaled dot-product window attention with a relative-position bias TABLE, PRE-norm, both halves in one call.

Both are FROZEN in the reference (``main_trans.py:211-256``): gradients flow through them to the adapters below, their own
parameters get none.  Here they are plain PyTorch-ROCm modules (bf16 matmul / softmax / LayerNorm / GELU from ATen: device
memory, streams and autograd are PyTorch's job; none of this is on the graded adapter path) with the reference's attribute names,
so a checkpoint loads by name, and with the signatures ``AdapterStack.forward(vis_block=, aud_block=)`` expects.

Parity: ``HTSATBlock`` is pinned against the reference class (``oracle/make_golden_backbone.py`` -> ``tests/golden/htsat_block.pt``).
``SwinV2Block`` restates timm==0.6.12's ``SwinTransformerV2Block`` (``requirements.txt:39``; un-vendored, not installed here, no
reference test pins it): **parity against timm unpinned**; ``tests/test_backbone.py`` checks its windowing / shift / mask plumbing against a
direct per-token evaluation of the same published formulas, and the whole block (output, input gradient, 1e-5 in fp32) against an INDEPENDENT
implementation of the same published block that is importable here -- Hugging Face transformers' ``Swinv2Layer``, parameters renamed to timm's
layout (``oracle/make_golden_swinv2.py`` -> ``tests/golden/swinv2_block.pt``).

On the GPU in bf16 (``fused=None``) the window attention of both blocks is one HIP kernel each way (``csrc/wattn.hip``; Swin-V2's q / k
normalisation inside) and their LayerNorms (+ the post-norm residual) run on the adapter tail's row kernels (``dgsct_layer_norm_*``); the
qkv / proj / MLP Linears and GELU stay ATen.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


def _to_windows(x: torch.Tensor, ws: int) -> torch.Tensor:
    """[B, H, W, C] -> [B * nW, ws * ws, C], windows in row-major order"""
    B, H, W, C = x.shape
    x = x.reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws * ws, C)


def _from_windows(w: torch.Tensor, ws: int, H: int, W: int) -> torch.Tensor:
    """inverse of _to_windows: [B * nW, ws * ws, C] -> [B, H, W, C]"""
    C = w.shape[-1]
    B = w.shape[0] // ((H // ws) * (W // ws))
    x = w.reshape(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, H, W, C)


def _pair_index(ws: int) -> torch.Tensor:
    """index of the relative offset (dy, dx) of every token pair of a ws x ws window into a (2 ws - 1)^2 table"""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    pos = torch.stack([ys.reshape(-1), xs.reshape(-1)])                       # [2, ws*ws]
    rel = pos[:, :, None] - pos[:, None, :] + (ws - 1)                        # [2, n, n] in 0 .. 2 ws - 2
    return rel[0] * (2 * ws - 1) + rel[1]


def _shift_mask(H: int, W: int, ws: int, shift: int) -> Optional[torch.Tensor]:
    """additive mask [nW, n, n] (0 / -100) that keeps the tokens a cyclic shift brought together from attending to each other"""
    if shift == 0:
        return None
    region = torch.zeros(1, H, W, 1)
    cuts = (slice(0, -ws), slice(-ws, -shift), slice(-shift, None))
    k = 0
    for hs in cuts:
        for wsl in cuts:
            region[:, hs, wsl, :] = k
            k += 1
    r = _to_windows(region, ws).squeeze(-1)                                    # [nW, n]
    diff = r[:, None, :] - r[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def _ln(norm: nn.LayerNorm, x: torch.Tensor, residual: Optional[torch.Tensor] = None, lib=None) -> torch.Tensor:
    """`norm(x)` (+ residual) on the library's row kernels (dgsct_layer_norm_*: one pass forward, one backward, the residual add of a
    post-norm half-block folded in) instead of ATen's LayerNorm + add.  Frozen norms (the reference freezes both backbones,
    main_trans.py:211-256) use an fp32 copy of weight / bias made once; trainable ones go through a differentiable cast."""
    from . import ops
    w, b = norm.weight, norm.bias
    if torch.is_grad_enabled() and (w.requires_grad or b.requires_grad):
        return ops.layer_norm(x, w.float(), b.float(), norm.eps, residual, lib)
    key = (w._version, b._version, w.device)
    c = norm.__dict__.get("_f32")
    if c is None or c[0] != key:
        with torch.no_grad():
            c = norm.__dict__["_f32"] = (key, w.detach().float().contiguous(), b.detach().float().contiguous())
    return ops.layer_norm(x, c[1], c[2], norm.eps, residual, lib)


class _Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _WindowAttentionV1(nn.Module):
    """htsat.py:50-132: softmax(q k^T / sqrt(d) + bias_table[pair] (+ mask)) v, then proj"""

    def __init__(self, dim: int, ws: int, heads: int):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, (ws, ws), heads
        self.scale = (dim // heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.register_buffer("relative_position_index", _pair_index(ws))
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def _tables(self, mask):
        """frozen per-block tables of the fused kernel (csrc/wattn.hip): bm [1 | nW, heads, n, n] = relative-position bias (+ shift mask),
        scale [heads]; fp32, rebuilt only when the bias table changes (it is frozen in the reference: main_trans.py:211-256)"""
        tab = self.relative_position_bias_table
        key = (tab._version, tab.device, mask is None)
        if getattr(self, "_tab_cache", (None,))[0] != key:
            n, h = self.window_size[0] * self.window_size[1], self.num_heads
            with torch.no_grad():
                bm = tab[self.relative_position_index.reshape(-1)].reshape(n, n, h).permute(2, 0, 1).float()[None]
                if mask is not None:
                    bm = bm + mask.float()[:, None]
                self._tab_cache = (key, bm.contiguous(), torch.full((h,), float(self.scale), dtype=torch.float32, device=tab.device))
        return self._tab_cache[1], self._tab_cache[2]

    def forward_map(self, y, H, W, shift, mask=None, lib=None):
        """the same attention on the UN-partitioned map y [B, H*W, C]: qkv projection of the map, then one fused kernel that does the
        window partition / cyclic shift by address arithmetic (no roll / partition copies, no [windows, heads, n, n] tensor), then proj"""
        from . import ops
        bm, scale = self._tables(mask)
        o = ops.window_attention(self.qkv(y), bm, scale, H, W, self.window_size[0], shift, self.num_heads, lib)
        return self.proj(o)

    def forward(self, x, mask=None):
        Bw, n, C = x.shape
        h = self.num_heads
        q, k, v = self.qkv(x).reshape(Bw, n, 3, h, C // h).permute(2, 0, 3, 1, 4)
        logits = (q * self.scale) @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index.reshape(-1)].reshape(n, n, h).permute(2, 0, 1)
        logits = logits + bias.to(logits.dtype)[None]
        if mask is not None:
            nW = mask.shape[0]
            logits = (logits.reshape(Bw // nW, nW, h, n, n) + mask.to(logits.dtype)[None, :, None]).reshape(Bw, h, n, n)
        attn = logits.softmax(-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(Bw, n, C)), attn


class HTSATBlock(nn.Module):
    """``SwinTransformerBlock`` of htsat.py:135-251 (LayerNorm before the MLP, no drop-path / dropout: the frozen model runs them at
    0): ``forward(x [B, H*W, C]) -> (x, attn)`` like the reference call ``f_a, _ = blk_a(f_a)``."""

    def __init__(self, dim: int, input_resolution: Tuple[int, int], num_heads: int, window_size: int = 8, shift_size: int = 0,
                 mlp_ratio: float = 4.0, fused: Optional[bool] = None, lib=None):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, tuple(input_resolution), num_heads
        # fused: None = the HIP window-attention kernel whenever it applies (bf16 on the GPU), False = the ATen formulation (the one pinned
        # bit-exactly to the reference class on the CPU); lib: a C-ABI instance other than the default (the host emulation of the CPU tests)
        self.fused, self._lib = fused, lib
        if min(self.input_resolution) <= window_size:                          # a window as large as the map: one window, no shift
            shift_size, window_size = 0, min(self.input_resolution)
        self.window_size, self.shift_size = window_size, shift_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _WindowAttentionV1(dim, window_size, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.register_buffer("attn_mask", _shift_mask(*self.input_resolution, window_size, shift_size))

    def _use_fused(self, x) -> bool:
        if self.fused is False or x.dtype != torch.bfloat16 or not (x.is_cuda or self._lib is not None):
            return False
        from . import ops
        return ops.window_attention_supported(x.new_empty(1, 1, 3 * self.dim), self.window_size, self.num_heads)

    def forward(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        if self._use_fused(x):
            # fused window attention (csrc/wattn.hip): the attention probabilities are never materialised, so the second result of the
            # reference call `f_a, _ = blk_a(f_a)` (net_trans.py:897, discarded there) is None on this path
            x = x + self.attn.forward_map(_ln(self.norm1, x, None, self._lib), H, W, self.shift_size, self.attn_mask, self._lib)
            return x + self.mlp(_ln(self.norm2, x, None, self._lib)), None
        y = self.norm1(x).reshape(B, H, W, C)
        s = self.shift_size
        if s:
            y = torch.roll(y, shifts=(-s, -s), dims=(1, 2))
        yw, attn = self.attn(_to_windows(y, self.window_size), self.attn_mask)
        y = _from_windows(yw, self.window_size, H, W)
        if s:
            y = torch.roll(y, shifts=(s, s), dims=(1, 2))
        x = x + y.reshape(B, L, C)
        return x + self.mlp(self.norm2(x)), attn


class _WindowAttentionV2(nn.Module):
    """Swin-V2 (Liu et al. 2022, section 3.2-3.3; timm 0.6.12 ``swin_transformer_v2.WindowAttention``): cosine attention
    ``cos(q, k) * exp(min(logit_scale, log 100))`` + ``16 sigmoid(cpb_mlp(log-spaced offsets))[pair]`` (+ mask); q / v biases, no k bias"""

    def __init__(self, dim: int, ws: int, heads: int, pretrained_ws: int = 0):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, (ws, ws), heads
        self.logit_scale = nn.Parameter(torch.log(10 * torch.ones(heads, 1, 1)))
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512), nn.ReLU(inplace=True), nn.Linear(512, heads, bias=False))
        r = torch.arange(-(ws - 1), ws, dtype=torch.float32)
        tab = torch.stack(torch.meshgrid(r, r, indexing="ij"), dim=-1)[None]   # [1, 2ws-1, 2ws-1, 2]
        tab = tab / float((pretrained_ws or ws) - 1) * 8
        tab = torch.sign(tab) * torch.log2(tab.abs() + 1.0) / math.log2(8)
        self.register_buffer("relative_coords_table", tab, persistent=False)
        self.register_buffer("relative_position_index", _pair_index(ws), persistent=False)
        self.qkv = nn.Linear(dim, 3 * dim, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.proj = nn.Linear(dim, dim)

    def _tables(self, mask):
        """frozen tables of the fused kernel: bm = 16 sigmoid(cpb_mlp(offsets))[pair] (+ shift mask), scale = exp(min(logit_scale, log 100))"""
        w = self.cpb_mlp[0].weight
        key = (w._version, self.cpb_mlp[2].weight._version, self.logit_scale._version, w.device, mask is None)
        if getattr(self, "_tab_cache", (None,))[0] != key:
            n, h = self.window_size[0] * self.window_size[1], self.num_heads
            with torch.no_grad():
                table = self.cpb_mlp(self.relative_coords_table.to(w.dtype)).reshape(-1, h).float()
                bm = (16 * torch.sigmoid(table[self.relative_position_index.reshape(-1)].reshape(n, n, h).permute(2, 0, 1)))[None]
                if mask is not None:
                    bm = bm + mask.float()[:, None]
                scale = torch.clamp(self.logit_scale.float(), max=math.log(100.0)).exp().reshape(h)
                self._tab_cache = (key, bm.contiguous(), scale.contiguous())
        return self._tab_cache[1], self._tab_cache[2]

    def forward_map(self, y, H, W, shift, mask=None, lib=None):
        """cosine window attention on the un-partitioned map (see _WindowAttentionV1.forward_map): the fused kernel (csrc/wattn.hip) normalises
        the q / k rows of a head in LDS (F.normalize), applies the per-head logit scale, the continuous position bias and the shift mask, and
        its backward returns the gradient of the raw projection"""
        from . import ops
        B, L, C = y.shape
        h = self.num_heads
        bias3 = torch.cat([self.q_bias, torch.zeros_like(self.v_bias), self.v_bias])
        qkv = F.linear(y, self.qkv.weight, bias3)
        bm, scale = self._tables(mask)
        return self.proj(ops.window_attention(qkv, bm, scale, H, W, self.window_size[0], shift, h, lib, cosine=True))

    def forward(self, x, mask=None):
        Bw, n, C = x.shape
        h = self.num_heads
        bias3 = torch.cat([self.q_bias, torch.zeros_like(self.v_bias), self.v_bias])
        q, k, v = F.linear(x, self.qkv.weight, bias3).reshape(Bw, n, 3, h, C // h).permute(2, 0, 3, 1, 4)
        logits = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
        logits = logits * torch.clamp(self.logit_scale, max=math.log(100.0)).exp().to(logits.dtype)
        table = self.cpb_mlp(self.relative_coords_table.to(self.cpb_mlp[0].weight.dtype)).reshape(-1, h)
        bias = 16 * torch.sigmoid(table[self.relative_position_index.reshape(-1)].reshape(n, n, h).permute(2, 0, 1))
        logits = logits + bias.to(logits.dtype)[None]
        if mask is not None:
            nW = mask.shape[0]
            logits = (logits.reshape(Bw // nW, nW, h, n, n) + mask.to(logits.dtype)[None, :, None]).reshape(Bw, h, n, n)
        return self.proj((logits.softmax(-1) @ v).transpose(1, 2).reshape(Bw, n, C))


class SwinV2Block(nn.Module):
    """timm 0.6.12 ``SwinTransformerV2Block`` with its two residual branches exposed the way ``net_trans.py:894, 903`` uses them:
    ``attn_branch(x) = norm1(_attn(x))`` and ``mlp_branch(x) = norm2(mlp(x))`` (post-norm; drop-path 0 in the frozen model)."""

    def __init__(self, dim: int, input_resolution: Tuple[int, int], num_heads: int, window_size: int = 12, shift_size: int = 0,
                 mlp_ratio: float = 4.0, fused: Optional[bool] = None, lib=None):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, tuple(input_resolution), num_heads
        self.fused, self._lib = fused, lib                                     # (as HTSATBlock)
        if min(self.input_resolution) <= window_size:
            shift_size, window_size = 0, min(self.input_resolution)
        self.window_size, self.shift_size = window_size, shift_size
        self.attn = _WindowAttentionV2(dim, window_size, num_heads)
        self.norm1 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.norm2 = nn.LayerNorm(dim)
        self.register_buffer("attn_mask", _shift_mask(*self.input_resolution, window_size, shift_size))

    _use_fused = HTSATBlock._use_fused

    # buffers a timm checkpoint may or may not carry depending on the release (registered persistent in some, not in others); they are
    # pure functions of the window size, rebuilt by the constructor, so a by-name load ignores them
    _DERIVED = ("relative_coords_table", "relative_position_index")

    def load_timm_state_dict(self, sd, strict: bool = True):
        """load a timm `SwinTransformerV2Block` state dict by name: derived index / offset tables are dropped, everything else must match"""
        sd = {k: v for k, v in sd.items() if not k.endswith(self._DERIVED)}
        return self.load_state_dict(sd, strict=strict)

    def _attn(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        if self._use_fused(x):
            return self.attn.forward_map(x, H, W, self.shift_size, self.attn_mask, self._lib)
        y = x.reshape(B, H, W, C)
        s = self.shift_size
        if s:
            y = torch.roll(y, shifts=(-s, -s), dims=(1, 2))
        y = _from_windows(self.attn(_to_windows(y, self.window_size), self.attn_mask), self.window_size, H, W)
        if s:
            y = torch.roll(y, shifts=(s, s), dims=(1, 2))
        return y.reshape(B, L, C)

    def attn_branch(self, x, add_residual: bool = False):
        """norm1(_attn(x)); add_residual: the whole line `x + norm1(_attn(x))` of net_trans.py:894 (one kernel with the fused LayerNorm)"""
        if self._use_fused(x):
            return _ln(self.norm1, self._attn(x), x if add_residual else None, self._lib)
        y = self.norm1(self._attn(x))
        return x + y if add_residual else y

    def mlp_branch(self, x, add_residual: bool = False):
        if self._use_fused(x):
            return _ln(self.norm2, self.mlp(x), x if add_residual else None, self._lib)
        y = self.norm2(self.mlp(x))
        return x + y if add_residual else y

    def forward(self, x):
        x = self.attn_branch(x, add_residual=True)
        return self.mlp_branch(x, add_residual=True)


# heads per stage: Swin-V2-B / -L (timm `swinv2_{base,large}_window12_192_22k`), HTS-AT (esc_config.py:67)
_SWIN_HEADS = {128: 4, 256: 8, 512: 16, 1024: 32, 192: 6, 384: 12, 768: 24, 1536: 48}
_HTSAT_HEADS = {96: 4, 192: 8, 384: 16, 768: 32}


class FrozenBlocks(nn.Module):
    """The frozen blocks that sit beside the adapter positions of an ``AdapterStack`` (one Swin-V2 block and one HTS-AT block per
    adapter layer; shifted windows on every second block of a stage, as the backbones alternate), randomly initialised -- no
    pretrained weights exist offline -- and frozen.  ``vis_block`` / ``aud_block`` are the callables ``AdapterStack.forward`` takes."""

    def __init__(self, stages: Sequence[Dict[str, int]], dtype: torch.dtype = torch.bfloat16, fused: Optional[bool] = None):
        super().__init__()
        vis, aud = [], []
        for s in stages:
            rv, ra = int(round(math.sqrt(s["Nv"]))), int(round(math.sqrt(s["Na"])))
            for i in range(s["layers"]):
                vis.append(SwinV2Block(s["Cv"], (rv, rv), _SWIN_HEADS[s["Cv"]], window_size=12, shift_size=0 if i % 2 == 0 else 6, fused=fused))
                aud.append(HTSATBlock(s["Ca"], (ra, ra), _HTSAT_HEADS[s["Ca"]], window_size=8, shift_size=0 if i % 2 == 0 else 4, fused=fused))
        self.vis, self.aud = nn.ModuleList(vis), nn.ModuleList(aud)
        self.to(dtype)
        for p in self.parameters():
            p.requires_grad_(False)

    def vis_block(self, idx: int, half: int, f_v: torch.Tensor) -> torch.Tensor:
        """the residual branch of Swin block `idx`: half 0 = window attention, half 1 = MLP (net_trans.py:894 / :903)"""
        blk = self.vis[idx]
        return blk.attn_branch(f_v) if half == 0 else blk.mlp_branch(f_v)

    def vis_block_map(self, idx: int, half: int, f_v: torch.Tensor) -> torch.Tensor:
        """the whole residual line `f_v + branch(f_v)`: what AdapterStack takes when the callable says `returns_map` (the add then sits in
        the fused LayerNorm kernel instead of a launch of its own)"""
        blk = self.vis[idx]
        return blk.attn_branch(f_v, True) if half == 0 else blk.mlp_branch(f_v, True)
    vis_block_map.returns_map = True

    def aud_block(self, idx: int, f_a: torch.Tensor) -> torch.Tensor:
        """the whole HTS-AT block `idx` (net_trans.py:897): returns the updated map"""
        return self.aud[idx](f_a)[0]
