"""The adapter call schedule of the AVE model (reference ``MMIL_Net``,
``DG-SCT/AVE/nets/net_trans.py``: per-layer dims ``:775-797``, the four ``ModuleList``s ``:807-845``,
the interleaving loop ``:880-916``), with the frozen Swin-V2 / HTS-AT blocks as pluggable callables
(identity stand-ins by default: the backbones are out of scope).

Used by bench.py (the graded workload) and by the identity-backbone stack fixtures.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .adapter import VisualAdapter

# (adapter layers in the stage, visual tokens, audio tokens); widths per backbone below.
# Swin-V2 @192^2, window 12: 48^2,24^2,12^2,6^2 tokens; HTS-AT spec 256, patch 4: 64^2,...,8^2 (esc_config.py:63-69).
# Stage 2: Swin has 18 blocks, HTS-AT 6 -> adapters at every 3rd Swin block (net_trans.py:885).
_STAGE_TOKENS = [(2, 2304, 4096), (2, 576, 1024), (6, 144, 256), (2, 36, 64)]
_WIDTHS = {
    "swinv2_large": [192, 384, 768, 1536],   # what the reference code builds (net_trans.py:693)
    "swinv2_base": [128, 256, 512, 1024],    # BASELINE.json config 2
}
_AUDIO_WIDTHS = [96, 192, 384, 768]


def ave_stage_shapes(backbone: str = "swinv2_base") -> List[Dict[str, int]]:
    return [dict(layers=l, Nv=nv, Cv=cv, Na=na, Ca=ca)
            for (l, nv, na), cv, ca in zip(_STAGE_TOKENS, _WIDTHS[backbone], _AUDIO_WIDTHS)]


def default_opt(**over) -> SimpleNamespace:
    """The adapter-relevant flags of DG-SCT/AVE/train.sh + base_options.py:158-178."""
    o = dict(is_multimodal=1, num_conv_group=2, is_before_layernorm=1, is_post_layernorm=1, num_tokens=32,
             Adapter_downsample=8, is_bn=1, is_gate=1, is_audio_adapter_p1=1, is_audio_adapter_p2=1)
    o.update(over)
    return SimpleNamespace(**o)


class AdapterStack(nn.Module):
    """4 x L adapters (audio/visual x p1/p2) in the reference's ModuleLists, plus the layer loop."""

    def __init__(self, stages: Sequence[Dict[str, int]], opt: Optional[SimpleNamespace] = None, flavour: str = "ave",
                 compute_dtype: Optional[torch.dtype] = None, lib=None, concurrent: bool = True, fuse_residual: bool = True,
                 fp8_projections: bool = False, pair_backward: Optional[bool] = None):
        super().__init__()
        self.concurrent = concurrent
        # both adapters of a position as one autograd node whose backward forms d f = dX(own) + dY(other) inside the dY products
        # (ops._PairFlatFn); None = on unless DGSCT_NO_PAIR=1.  Needs the fused residual, flat parameters and no frozen block between
        # a map and its adapter -- anything else takes the two-node path.
        self.pair_backward = pair_backward
        self.fuse_residual = fuse_residual        # `f = f + adapter(...)` inside the adapter's last kernel (8f row f2)
        self.opt = opt or default_opt()
        o = self.opt
        self.stages = [dict(s) for s in stages]
        hidden, hidden_a, conv, conv_a = [], [], [], []
        for s in self.stages:
            for _ in range(s["layers"]):
                hidden.append(s["Cv"]); hidden_a.append(s["Ca"]); conv.append(s["Nv"]); conv_a.append(s["Na"])
        kw = dict(flavour=flavour, compute_dtype=compute_dtype, lib=lib)
        if fp8_projections:                       # BASELINE configs[4]: e4m3 MFMA operands for fc / fc_affine_video_1 / fc_affine_video_2 (forward)
            kw["fp8_projections"] = True
        if flavour in ("ave", "avvp", "pretrain"):
            kw["num_tk"] = o.num_tokens

        def audio(i):
            return VisualAdapter(input_dim=hidden_a[i], output_dim=hidden_a[i], adapter_kind="bottleneck", dim_list=hidden_a,
                                 layer_idx=i, reduction_factor=o.Adapter_downsample, opt=o, use_bn=o.is_bn, use_gate=o.is_gate,
                                 conv_dim_in=conv[i], conv_dim_out=conv_a[i], linear_in=hidden[i], linear_out=hidden_a[i], **kw)

        def visual(i):
            return VisualAdapter(input_dim=hidden[i], output_dim=hidden[i], adapter_kind="bottleneck", dim_list=hidden,
                                 layer_idx=i, reduction_factor=o.Adapter_downsample, opt=o, use_bn=o.is_bn, use_gate=True,
                                 conv_dim_in=conv_a[i], conv_dim_out=conv[i], linear_in=hidden_a[i], linear_out=hidden[i], **kw)

        n = len(hidden)
        self.audio_adapter_blocks_p1 = nn.ModuleList([audio(i) for i in range(n)])
        self.vis_adapter_blocks_p1 = nn.ModuleList([visual(i) for i in range(n)])
        self.audio_adapter_blocks_p2 = nn.ModuleList([audio(i) for i in range(n)])
        self.vis_adapter_blocks_p2 = nn.ModuleList([visual(i) for i in range(n)])

    def flatten_parameters(self):
        """one flat fp32 parameter per adapter (see VisualAdapter.flatten_parameters); 48 tensors for the AVE stack"""
        for ml in (self.audio_adapter_blocks_p1, self.vis_adapter_blocks_p1, self.audio_adapter_blocks_p2,
                   self.vis_adapter_blocks_p2):
            for m in ml:
                m.flatten_parameters()
        return self

    @staticmethod
    def _view(f: torch.Tensor) -> torch.Tensor:
        return f.permute(0, 2, 1).unsqueeze(-1)          # the reference's [BT,C,N,1] view of a token-major map

    def forward(self, feats: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                vis_block: Optional[Callable] = None, aud_block: Optional[Callable] = None):
        """feats[s] = (f_v [BT,Nv,Cv], f_a [BT,Na,Ca]) entering stage s.  ``vis_block(layer, half, f_v)`` /
        ``aud_block(layer, f_a)`` return the frozen residual branches (None = identity stand-in); a ``vis_block`` with the attribute
        ``returns_map = True`` returns ``f_v + branch(f_v)`` itself.
        Returns ([(f_v, f_a) leaving each stage], (map_v, map_a) of the last p2 adapters)."""
        outs = []
        idx = 0
        maps = (None, None)
        # The audio and the visual adapter of a position read the same pre-block maps and are independent
        # (net_trans.py:891-892): run them on two HIP streams so the many small kernels of the late stages overlap.
        # autograd replays each backward on the stream of its forward, so backward overlaps the same way.
        dev = feats[0][0].device
        side = None
        if self.concurrent and dev.type == "cuda":
            from . import _lib, ops
            side = ops.side_stream(self.audio_adapter_blocks_p1[0]._lib or _lib.default_lib(), dev)

        fuse = self.fuse_residual

        def call(mod, f_own, f_other, f_res):
            """adapter on the pre-block maps; with fuse: returns f_res + adapter(...) token-major (f_res is f_own when no
            frozen block sits in between -> the fully fused skip), else the reference's [BT,C,N,1] result."""
            if not fuse:
                return mod(self._view(f_own), self._view(f_other))
            if f_res is f_own:
                r = mod(self._view(f_own), self._view(f_other), skip=True)
            else:
                r = mod(self._view(f_own), self._view(f_other), residual=self._view(f_res))
            return (r[0].squeeze(-1).permute(0, 2, 1),) + tuple(r[1:])

        from . import ops as _ops
        use_pair = _ops.PAIR_BACKWARD if self.pair_backward is None else bool(self.pair_backward)

        def pair_node(audio_mod, vis_mod, f_a, f_v):
            """the one-node path, or None when this position does not qualify"""
            for m in (audio_mod, vis_mod):
                if "_flat_views" not in m.__dict__ or m.flavour == "pretrain":
                    return None
            cd = audio_mod._compute_dtype_for(f_a.dtype)
            if (f_a.dtype != cd or f_v.dtype != cd or vis_mod._compute_dtype_for(f_v.dtype) != cd or f_a.device != f_v.device
                    or not f_a.is_contiguous() or not f_v.is_contiguous()):
                return None
            if not f_a.is_cuda and audio_mod._lib is None:
                return None
            la, sa, ta, pa, pla, fa = audio_mod._token_call(cd, f_a.device)
            lv, sv, tv, pv, plv, fv = vis_mod._token_call(cd, f_v.device)
            if la is not lv:
                raise RuntimeError("dg-sct_amd: the adapters of one position were built on different library instances")
            oa, ma, ov, mv = _ops.pair_apply(la, side, (sa, ta, pa, pla), (sv, tv, pv, plv), f_a, f_v, fa, fv)
            return (oa, ma.unsqueeze(1)), (ov, mv.unsqueeze(1))

        def pair(audio_mod, vis_mod, f_a, f_v, r_a, r_v):
            if use_pair and fuse and r_a is f_a and r_v is f_v:
                got = pair_node(audio_mod, vis_mod, f_a, f_v)
                if got is not None:
                    return got
            if side is None:
                return call(audio_mod, f_a, f_v, r_a), call(vis_mod, f_v, f_a, r_v)
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                a = call(audio_mod, f_a, f_v, r_a)
            v = call(vis_mod, f_v, f_a, r_v)
            main.wait_stream(side)
            for t in a:                                  # allocated on the side stream, consumed on the main one
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)
            return a, v

        def step(p_audio, p_vis, f_a, f_v, idx, half, with_aud_block):
            # frozen blocks first (they only read the pre-block maps), so their output can be the fused residual
            if vis_block is None:
                r_v = f_v
            elif getattr(vis_block, "returns_map", False):       # the callable returns f_v + branch(f_v) itself (FrozenBlocks.vis_block_map)
                r_v = vis_block(idx, half, f_v)
            else:
                r_v = f_v + vis_block(idx, half, f_v)
            r_a = f_a if (aud_block is None or not with_aud_block) else aud_block(idx, f_a)
            a, v = pair(p_audio, p_vis, f_a, f_v, r_a, r_v)
            if fuse:
                return a[0], v[0], a[1], v[1]
            return (r_a + a[0].squeeze(-1).permute(0, 2, 1), r_v + v[0].squeeze(-1).permute(0, 2, 1), a[1], v[1])

        # BatchNorm's num_batches_tracked (2 per adapter call): collected and bumped by ONE _foreach_add_ after the loop -- as 96
        # one-element kernels they sat on the adapter streams' dependency chains (~9 us per forward call)
        counters: List[torch.Tensor] = []
        mods = [m for ml in (self.audio_adapter_blocks_p1, self.vis_adapter_blocks_p1, self.audio_adapter_blocks_p2,
                             self.vis_adapter_blocks_p2) for m in ml]
        for m in mods:
            m.__dict__["_count_later"] = counters
        try:
            outs, maps = self._layers(feats, step, outs, maps)
        finally:
            for m in mods:
                m.__dict__.pop("_count_later", None)
        if counters:
            torch._foreach_add_(counters, 1)
        return outs, maps

    def _layers(self, feats, step, outs, maps):
        idx = 0
        for s, (f_v, f_a) in zip(self.stages, feats):
            for _ in range(s["layers"]):
                f_a, f_v, _, _ = step(self.audio_adapter_blocks_p1[idx], self.vis_adapter_blocks_p1[idx], f_a, f_v, idx, 0, True)
                f_a, f_v, a_map, v_map = step(self.audio_adapter_blocks_p2[idx], self.vis_adapter_blocks_p2[idx], f_a, f_v, idx, 1,
                                              False)
                maps = (v_map, a_map)
                idx += 1
            outs.append((f_v, f_a))
        return outs, maps
