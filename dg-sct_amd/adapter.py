"""Drop-in ``VisualAdapter(nn.Module)`` -- the reference's plugin interface for the hot path.

Mirrors reference ``DG-SCT/AVE/nets/net_trans.py:433-674`` (ctor :437, forward :552, returns :674):
same constructor arguments, same parameter/buffer names and shapes (so ``best_82.18.pt`` and any
reference ``state_dict`` load by name, ``main_trans.py:306``; freeze-by-name ``'adapter_blocks' in
name`` keeps working, ``main_trans.py:242``), same init distributions, same call convention
(``x``/``vis_token`` are the ``[BT,C,N,1]`` permuted views of token-major maps, ``:891-892``) and the
same error behaviour (``NotImplementedError`` for unknown ``adapter_kind``, ``:549-550``).

The other five copies of the class are flavours (``flavour=``): AVVP ``mgn.py:162-414``, AVS-S4/MS3
``PVT_AVSModel.py:90-316 / 90-300``, AVQA ``net_avst.py:27-218``, pretrain/few/zero-shot
``pretrain/nets/net_trans.py:343-600``.

The arithmetic runs in libdgsct.so (hand-written gfx950 kernels); there is no PyTorch or CPU
fallback: CPU tensors raise.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

import warnings

from . import _lib, ops
from ._lib import PARAM_NAMES

_WARNED = set()


def _warn_once(msg: str):
    if msg not in _WARNED:
        _WARNED.add(msg)
        warnings.warn(msg, stacklevel=3)

FLAVOUR_DEFAULTS = {
    #            remap      alpha beta  gamma temporal ln_before_ok gate_first T   tokens_init has(num_tk arg)
    "ave":      dict(remap="conv", alpha=0.3, beta=0.05, gamma=0.0, temporal=False, ln_before_ok=True, gate_first=False, T=10, tokens="rand"),
    "avvp":     dict(remap="conv", alpha=0.3, beta=0.05, gamma=0.0, temporal=False, ln_before_ok=True, gate_first=False, T=10, tokens="rand"),
    "avs_s4":   dict(remap="bicubic", alpha=0.3, beta=0.05, gamma=0.0, temporal=False, ln_before_ok=False, gate_first=True, T=5, tokens="zeros"),
    "avs_ms3":  dict(remap="conv", alpha=0.2, beta=0.1, gamma=0.0, temporal=False, ln_before_ok=False, gate_first=True, T=5, tokens="zeros"),
    "avqa":     dict(remap="conv", alpha=0.3, beta=0.05, gamma=0.0, temporal=False, ln_before_ok=True, gate_first=False, T=10, tokens="zeros"),
    "pretrain": dict(remap="conv", alpha=0.3, beta=0.01, gamma=0.05, temporal=True, ln_before_ok=True, gate_first=False, T=10, tokens="rand"),
}


def bicubic_matrix(No: int, N: int) -> torch.Tensor:
    """Dense [N, No] operator of F.interpolate(mode='bicubic') between square token grids
    (AVS-S4 remap, PVT_AVSModel.py:190-197): the resize is linear in the tokens, so it takes the
    place of conv_adapter.weight in the same MFMA GEMM."""
    hi, ho = int(math.isqrt(No)), int(math.isqrt(N))
    if hi * hi != No or ho * ho != N:
        raise ValueError("bicubic remap needs square token grids")
    eye = torch.eye(No, dtype=torch.float64).view(No, 1, hi, hi)
    out = F.interpolate(eye, size=[ho, ho], mode="bicubic")
    return out.view(No, N).t().contiguous().float()


class VisualAdapter(nn.Module):
    """Conventional bottleneck adapter with DG-SCT cross-modal gating (see module docstring)."""

    def __init__(self, input_dim, output_dim, adapter_kind, dim_list=None, layer_idx=0, reduction_factor=16, opt=None,
                 use_bn=True, use_gate=True, num_tk=None, conv_dim_in=0, conv_dim_out=0, linear_in=0, linear_out=0,
                 flavour: str = "ave", compute_dtype: Optional[torch.dtype] = None, lib: Optional[_lib.Lib] = None,
                 fp8_projections: bool = False):
        super().__init__()
        if flavour not in FLAVOUR_DEFAULTS:
            raise ValueError(f"unknown flavour {flavour!r}")
        fl = FLAVOUR_DEFAULTS[flavour]
        self.adapter_kind = adapter_kind
        self.use_bn = bool(use_bn)
        self.is_multimodal = bool(opt.is_multimodal)
        self.opt = opt
        self.flavour = flavour
        self.compute_dtype = compute_dtype
        self._lib = lib
        # AVE / AVVP / pretrain copies: `num_tk=87` is the constructor default (net_trans.py:437, mgn.py:166; every call site passes
        # opt.num_tokens); AVS / AVQA copies have no such argument and read opt.num_tokens (PVT_AVSModel.py:130, net_avst.py:60)
        if num_tk is None:
            num_tk = 87 if flavour in ("ave", "avvp", "pretrain") else opt.num_tokens
        self.num_tk = int(num_tk)
        if not (adapter_kind == "bottleneck" and self.is_multimodal):
            # "bottleneck" without is_multimodal and "basic" are never built by any reference launcher
            # (SURVEY 8a-1); anything else raises exactly like the reference (:549-550).
            raise NotImplementedError(f"adapter_kind={adapter_kind!r} with is_multimodal={self.is_multimodal} is not on the "
                                      f"DG-SCT hot path")
        if not 1 <= self.num_tk <= 1024:
            # num_tokens <= 32 (every reference launcher: AVE/AVVP train.sh 32, AVS 32, AVQA 2) runs on the fused attention kernels
            # (csrc/attn*.hip: one 32-row MFMA tile of latent tokens per frame); more -- the reference constructor's default is 87 --
            # on batched products + row softmax (csrc/attn_wide.cpp).  Fail here, with the reason, not inside dgsct_query.
            raise ValueError(f"dg-sct_amd: num_tokens={self.num_tk} is outside the supported range 1..1024; see INTEGRATION.md 'Limits'")
        if input_dim != output_dim or linear_out != input_dim:
            raise ValueError("DG-SCT adapters have input_dim == output_dim == linear_out")
        C, d_model = linear_out, linear_out // 2
        # --- parameter holders: stock modules give the reference names, shapes and init distributions
        self.conv_adapter = nn.Conv2d(conv_dim_in, conv_dim_out, kernel_size=1)
        self.fc = nn.Linear(linear_in, linear_out)
        self.conv_dim_out = conv_dim_out
        if flavour in ("avvp", "pretrain"):
            self.fc_caption = nn.Linear(512, 192)          # present in checkpoints, never on the used path
        self.fc_affine_audio_1 = nn.Linear(C, C)
        self.fc_affine_video_1 = nn.Linear(C, C)
        self.fc_affine_bottleneck = nn.Linear(C, d_model)
        self.fc_affine_video_2 = nn.Linear(C, d_model)
        self.fc_affine_audio_2 = nn.Linear(C, d_model)
        self.fc_affine_v_s_att = nn.Linear(d_model, 1)
        self.fc_affine_v_c_att = nn.Linear(d_model, C)
        if flavour in ("avvp", "avs_s4", "avs_ms3", "pretrain"):
            self.temporal_gated = nn.Sequential(nn.Linear(C, 1), nn.Sigmoid())
        self.gate = nn.Parameter(torch.zeros(1)) if use_gate else None
        self.down_sample_size = input_dim // reduction_factor
        if fl["tokens"] == "rand":
            self.my_tokens = nn.Parameter(torch.rand((self.num_tk, input_dim)))
        else:
            self.my_tokens = nn.Parameter(torch.zeros((self.num_tk, input_dim)))
        if flavour in ("ave", "avvp", "pretrain"):
            self.gate_tk = nn.Parameter(torch.ones(1))      # unused by the reference forward too (never gets a grad)
        self.gate_av = nn.Parameter(torch.zeros(1))
        g = int(opt.num_conv_group)
        self.down_sampler = nn.Conv2d(input_dim, self.down_sample_size, 1, groups=g, bias=False)
        self.up_sampler = nn.Conv2d(self.down_sample_size, output_dim, 1, groups=g, bias=False)
        if use_bn:
            self.bn1 = nn.BatchNorm2d(self.down_sample_size)
            self.bn2 = nn.BatchNorm2d(output_dim)
        if opt.is_before_layernorm:
            self.ln_before = nn.LayerNorm(output_dim)
            if not fl["ln_before_ok"]:
                # the AVS copies of the class build ln_before and never call it (PVT_AVSModel.py:239): mirrored, but say so
                _warn_once("VisualAdapter(flavour=%r): is_before_layernorm is set but this flavour's forward never applies "
                           "ln_before (as in the reference); the parameters exist for state_dict compatibility only" % flavour)
        if opt.is_post_layernorm:
            self.ln_post = nn.LayerNorm(output_dim)

        alpha = float(getattr(opt, "alpha", fl["alpha"])) if flavour == "pretrain" else fl["alpha"]
        beta = float(getattr(opt, "beta", fl["beta"])) if flavour == "pretrain" else fl["beta"]
        gamma = float(getattr(opt, "gamma", fl["gamma"])) if flavour == "pretrain" else fl["gamma"]
        self.spec = ops.AdapterSpec(
            N=int(conv_dim_out), C=int(C), No=int(conv_dim_in), Co=int(linear_in), tk=self.num_tk, r=int(reduction_factor), g=g,
            use_bn=bool(use_bn), use_gate=bool(use_gate), ln_before=bool(opt.is_before_layernorm) and fl["ln_before_ok"],
            ln_post=bool(opt.is_post_layernorm), gate_before_ln_post=fl["gate_first"], remap=fl["remap"],
            alpha=alpha, beta=beta, gamma=gamma, temporal=fl["temporal"], T=fl["T"], fp8=bool(fp8_projections))
        if fl["remap"] == "bicubic":
            self.register_buffer("_remap_op", bicubic_matrix(int(conv_dim_in), int(conv_dim_out)), persistent=False)
        self._prep_cache = None

    # ------------------------------------------------------------------
    def _lookup(self, name: str) -> Optional[torch.Tensor]:
        """Resolve a reference parameter / buffer name by ATTRIBUTE walk, not through named_parameters(): on an
        nn.DataParallel replica (torch.nn.parallel.replicate: reference AVS/AVQA call path, avs_s4/train.py:139,
        main_avst.py:236) ``_parameters`` is empty and the broadcast copies are plain tensor attributes."""
        if self.__dict__.get("_is_replica", False) and name in self.__dict__.get("_flat_layout", {}):
            return self._replica_flat_views()[name]
        obj = self
        for part in name.split("."):
            try:
                obj = getattr(obj, part)
            except AttributeError:
                return None
            if obj is None:
                return None
        return obj if isinstance(obj, torch.Tensor) else None

    def _replica_flat_views(self):
        """flat mode on a DataParallel replica: per-name views of THIS replica's copy of ``flat_param`` (the views the
        original module planted on its sub-modules point at the original's device)."""
        flat = self.flat_param
        c = self.__dict__.get("_rviews")
        if c is None or c[0] is not flat:
            d = flat.detach()
            c = (flat, {name: d[off:off + n].view(shape) for name, (off, n, shape) in self._flat_layout.items()})
            self.__dict__["_rviews"] = c
        return c[1]

    def _replicate_for_data_parallel(self):
        """nn.DataParallel replicas are shallow ``__dict__`` copies: give each its own caches (the parameter table holds
        tensors of the original's device; the prepared weights belong to the original's parameters)."""
        replica = super()._replicate_for_data_parallel()
        for k in ("_ptab", "_rviews"):
            replica.__dict__.pop(k, None)
        replica.__dict__["_prep_cache"] = None
        return replica

    def _param_list(self) -> List[Optional[torch.Tensor]]:
        """parameter table in C-ABI order; the name -> tensor resolution is done once per module object (Parameters keep
        their identity across .to()/.load_state_dict(); the 2-D views of the conv weights are re-made when their storage
        moves).  Replicas resolve on every call: their tensors are re-broadcast each forward."""
        replica = self.__dict__.get("_is_replica", False)
        cache = None if replica else self.__dict__.get("_ptab")
        if cache is None:
            cache = []
            for name in PARAM_NAMES:
                t = self._lookup(name)
                view = None
                if name == "conv_adapter.weight":
                    if self.spec.remap == "bicubic":
                        t = self._remap_op
                    else:
                        view = (self.spec.N, self.spec.No)
                elif name == "conv_adapter.bias" and self.spec.remap == "bicubic":
                    t = None
                elif name in ("down_sampler.weight", "up_sampler.weight"):
                    view = (t.shape[0], t.shape[1])
                elif name.startswith("temporal_gated") and not self.spec.temporal:
                    t = None
                elif name.startswith("ln_before") and not self.spec.ln_before:
                    t = None
                cache.append([t, view, None, 0])
            if not replica:
                self.__dict__["_ptab"] = cache
        out: List[Optional[torch.Tensor]] = []
        for ent in cache:
            t, view = ent[0], ent[1]
            if t is not None and view is not None:
                if ent[2] is None or ent[3] != t.data_ptr():
                    ent[2], ent[3] = t.view(*view), t.data_ptr()
                t = ent[2]
            out.append(t)
        return out

    def _apply(self, fn, *a, **k):
        self.__dict__.pop("_ptab", None)        # .to()/.cuda()/.float(): buffers may be replaced
        r = super()._apply(fn, *a, **k)
        if "_flat_views" in self.__dict__:
            self._rebuild_flat_views()
        return r

    # ------------------------------------------------------------------ flat parameters (opt-in)
    def flatten_parameters(self):
        """Move every trainable tensor of this adapter into ONE flat fp32 ``nn.Parameter`` (``flat_param``) laid out like
        the library's gradient buffer.  ``state_dict()`` / ``load_state_dict()`` keep the reference's names (hooks below);
        ``named_parameters()`` then lists ``flat_param`` instead of the ~30 individual tensors (its name still contains
        ``adapter_blocks`` inside the task models, so the reference's freeze-by-name rule is unaffected)."""
        if "_flat_views" in self.__dict__:
            return self
        lib = self._lib or _lib.default_lib()
        d = self.spec.desc(self.spec.T, torch.float32, True)
        lay = ops.grad_layout(lib, d)
        total = int(ops._sizes(lib, d).grad_floats)
        named = dict(self.named_parameters())
        dev = next(iter(named.values())).device
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._flat_layout = {}
        for i, name in enumerate(PARAM_NAMES):
            off, n = lay[i]
            if off < 0 or name not in named:
                continue
            p = named[name]
            flat[off:off + n].copy_(p.detach().reshape(-1))
            self._flat_layout[name] = (off, n, tuple(p.shape))
            owner, leaf = self, name
            if "." in name:
                path, leaf = name.rsplit(".", 1)
                owner = self.get_submodule(path)
            del owner._parameters[leaf]                   # the name stays reachable as a plain tensor view (set below)
        self.flat_param = nn.Parameter(flat)
        self.__dict__["_flat_views"] = {}
        self._rebuild_flat_views()
        self._register_state_dict_hook(VisualAdapter._flat_state_dict_hook)
        self._register_load_state_dict_pre_hook(self._flat_load_pre_hook)
        return self

    def _rebuild_flat_views(self):
        views = {}
        for name, (off, n, shape) in self._flat_layout.items():
            v = self.flat_param.data[off:off + n].view(shape)
            views[name] = v
            owner, leaf = self, name
            if "." in name:
                path, leaf = name.rsplit(".", 1)
                owner = self.get_submodule(path)
            object.__setattr__(owner, leaf, v)
        self.__dict__["_flat_views"] = views
        self.__dict__.pop("_ptab", None)

    @staticmethod
    def _flat_state_dict_hook(module, state_dict, prefix, local_metadata):
        state_dict.pop(prefix + "flat_param", None)
        for name, v in module.__dict__["_flat_views"].items():
            state_dict[prefix + name] = v.detach()
        return state_dict

    def _flat_load_pre_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for name, v in self.__dict__["_flat_views"].items():
            key = prefix + name
            if key in state_dict:
                with torch.no_grad():
                    v.copy_(state_dict.pop(key).reshape(v.shape))
            elif strict:
                missing_keys.append(key)
        state_dict[prefix + "flat_param"] = self.flat_param.detach()

    def _prepared(self, lib, params, dtype, device):
        """MFMA-operand weight copies + derived bias vectors (dgsct_prepare), re-made whenever a parameter changed.
        The key holds every tensor's (address, version counter).  In flat mode the per-name tensors are views of
        ``flat_param.data`` -- an alias with its OWN version counter that an optimizer step on ``flat_param`` never
        bumps -- so the flat parameter's own (address, version) is part of the key."""
        key = (dtype, device, tuple((p.data_ptr(), p._version) for p in params if p is not None))
        flat = self._parameters.get("flat_param") if "_flat_views" in self.__dict__ else None
        if flat is not None:
            key = key + ((flat.data_ptr(), flat._version),)
        if self._prep_cache is None or self._prep_cache[0] != key:
            self._prep_cache = (key, ops.prepare(lib, self.spec, params, dtype, device))
        return self._prep_cache[1]

    def invalidate_prep(self):
        """Force dgsct_prepare on the next forward (for parameter updates that bypass autograd's version counters,
        e.g. writes through raw pointers)."""
        self._prep_cache = None

    def _compute_dtype_for(self, in_dtype):
        return self.compute_dtype or (torch.bfloat16 if in_dtype == torch.bfloat16 else torch.float32)

    def _token_call(self, cd, device):
        """What one library call of this module needs besides the maps: (lib, spec, training, prep, params, flat parameter or None).
        Counts the call for BatchNorm's num_batches_tracked (training mode), as forward() always did."""
        lib = self._lib or _lib.default_lib()
        params = [ops.check_param(n, p, device) for n, p in zip(PARAM_NAMES, self._param_list())]
        prep = self._prepared(lib, params, cd, device)
        training = self.training
        flat = self.flat_param if "_flat_views" in self.__dict__ else None
        if training and self.use_bn:
            pend = self.__dict__.get("_count_later")
            if pend is not None:                         # AdapterStack: one _foreach_add_ per step instead of 96 one-element kernels
                pend.append(self.bn1.num_batches_tracked); pend.append(self.bn2.num_batches_tracked)
            else:
                self.bn1.num_batches_tracked += 1
                self.bn2.num_batches_tracked += 1
        return lib, self.spec, training, prep, params, flat

    def forward(self, x, vis_token=None, caption=None, is_temporal=False, residual=None, skip=False):
        """x [BT,C,N,1], vis_token [BT,Co,No,1] (views of token-major maps) ->
        (output [BT,C,N,1], spatial_att_maps [BT,1,N][, temporal_att_maps [BT/T,T,1,1]]).

        Extensions of the reference signature (the callers' `f = f + adapter(...)[0]` at
        net_trans.py:894-906): ``residual`` ([BT,C,N,1] view like x) -> output = residual + adapter(x, vis_token);
        ``skip=True`` -> output = x + adapter(x, vis_token) with the skip's share of d/dx fused into backward."""
        if caption is not None:
            raise NotImplementedError("caption prompts (AVVP mgn.py:306-308) are never passed by any reference launcher")
        if not x.is_cuda and self._lib is None:      # (tests inject the host-emulated library to check this plumbing)
            raise RuntimeError("dg-sct_amd.VisualAdapter runs on MI355X through libdgsct.so; there is no CPU path "
                               "(move the module and its inputs to a ROCm device)")
        X = x.squeeze(-1).permute(0, 2, 1)           # [BT,N,C]: contiguous when x is the reference's permuted view
        Y = vis_token.squeeze(-1).permute(0, 2, 1)
        in_dtype = x.dtype
        cd = self._compute_dtype_for(in_dtype)
        X = X.to(cd).contiguous()
        Y = Y.to(cd).contiguous()
        lib, spec, training, prep, params, flat = self._token_call(cd, X.device)
        res = None
        if residual is not None:
            res = residual.squeeze(-1).permute(0, 2, 1).to(cd).contiguous()
        out, amap, tmap = ops.adapter_apply(lib, spec, training, prep, X, Y, params, flat, residual=res, skip=skip)
        if out.dtype != in_dtype:
            out = out.to(in_dtype)
        output = out.permute(0, 2, 1).unsqueeze(-1)
        spatial = amap.unsqueeze(1)
        if self.flavour == "pretrain":
            T = self.spec.T
            return output, spatial, tmap.view(-1, T, 1, 1)
        return output, spatial
