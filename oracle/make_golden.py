"""Pin the oracle against the reference and emit golden vectors.  TEST INFRASTRUCTURE ONLY.

Runs ONLY in the build container (needs /root/reference).  For every adapter flavour it
  1. extracts the reference ``class VisualAdapter`` from the reference source file with ``ast`` and
     ``exec``s it in a namespace holding only torch / einops (nothing is copied into this repo),
  2. runs it on CPU (train-mode fwd+bwd with random cotangents, then an eval-mode forward),
  3. checks ``oracle.dgsct_oracle.forward/backward`` against it (asserts <= 1e-4 of max(1,|ref|max)), and
  4. writes inputs + expected outputs to ``tests/golden/<case>.pt`` (data only, < 200 KB each).

    python oracle/make_golden.py            # regenerate + validate everything
"""
from __future__ import annotations

import ast
import json
import math
import os
import sys
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dgsct_oracle as O  # noqa: E402

REF = "/root/reference"
REF_FILES = {
    "ave": "DG-SCT/AVE/nets/net_trans.py",
    "avvp": "DG-SCT/AVVP/nets/mgn.py",
    "avs_s4": "DG-SCT/AVS/avs_scripts/avs_s4/model/PVT_AVSModel.py",
    "avs_ms3": "DG-SCT/AVS/avs_scripts/avs_ms3/model/PVT_AVSModel.py",
    "avqa": "DG-SCT/AVQA/net_grd_avst/net_avst.py",
    "pretrain": "pretrain/nets/net_trans.py",
}


class _ContigGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


_BN_FORWARD = nn.BatchNorm2d.forward


def _bn_forward_workaround(self, x):
    """PyTorch 2.10 CPU bug workaround, applied to the REFERENCE module while vectors are generated:
    native_batch_norm_backward mis-computes dX when its cotangent is a permuted view and the saved
    input is NCHW-contiguous (the reference's bn2 -> permute -> ln_post chain hits exactly that).
    Stock autograd on the reference disagrees with float64 finite differences by O(1) (checked in
    ``fd_check`` below); with a contiguous input + contiguous cotangent it agrees to 1e-8."""
    return _ContigGrad.apply(_BN_FORWARD(self, x.contiguous()))


def load_reference_class(flavour: str):
    """ast-extract ``VisualAdapter`` from the reference file and exec it (SURVEY 8c recipe)."""
    from einops import rearrange, repeat
    path = os.path.join(REF, REF_FILES[flavour])
    src = open(path).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "VisualAdapter")
    mod = ast.Module(body=[node], type_ignores=[])
    ns = dict(torch=torch, nn=nn, F=F, math=math, rearrange=rearrange, repeat=repeat)
    exec(compile(mod, path, "exec"), ns)
    return ns["VisualAdapter"]


# case name -> (flavour, dims/flags)
CASES = {
    "ave_orderA": ("ave", dict(N=16, C=32, No=36, Co=16, tk=4, r=8, g=2)),
    "ave_orderB": ("ave", dict(N=36, C=16, No=16, Co=32, tk=4, r=4, g=2)),
    "ave_tk32": ("ave", dict(N=25, C=64, No=49, Co=48, tk=32, r=8, g=2)),
    "ave_audio_nogate": ("ave", dict(N=16, C=32, No=36, Co=16, tk=4, r=8, g=2, use_gate=False)),
    "ave_nobn_noln": ("ave", dict(N=16, C=32, No=36, Co=16, tk=4, r=8, g=2, use_bn=False, ln_before=False, ln_post=False)),
    "avvp": ("avvp", dict(N=16, C=32, No=36, Co=16, tk=4, r=8, g=2)),
    "avs_s4": ("avs_s4", dict(N=16, C=32, No=36, Co=16, tk=4, r=8, g=2)),
    "avs_s4_up": ("avs_s4", dict(N=36, C=16, No=16, Co=32, tk=4, r=4, g=2)),
    "avs_ms3": ("avs_ms3", dict(N=16, C=32, No=36, Co=16, tk=4, r=8, g=2)),
    "avqa": ("avqa", dict(N=16, C=32, No=36, Co=16, tk=2, r=8, g=4)),
    "avqa_audio_nogate": ("avqa", dict(N=36, C=16, No=16, Co=32, tk=2, r=4, g=4, use_gate=False)),
    "pretrain": ("pretrain", dict(N=16, C=32, No=36, Co=16, tk=4, r=8, g=2)),
    # num_tokens > 32 (more than one 32-row MFMA tile of latent tokens per frame: csrc/attn_wide.cpp); 87 is the reference
    # constructor's default (net_trans.py:437).  Appended: the seeds of the cases above (100 + index) do not move.
    "ave_tk40": ("ave", dict(N=36, C=32, No=16, Co=48, tk=40, r=8, g=2)),
    "ave_tk87": ("ave", dict(N=49, C=64, No=25, Co=32, tk=87, r=8, g=2)),
}
BATCH = 10   # BT = 1 clip x T=10 (2 clips x T=5 for AVS)


def build_reference(flavour: str, cfg: O.AdapterConfig):
    cls = load_reference_class(flavour)
    opt = SimpleNamespace(is_multimodal=1, num_conv_group=cfg.g, is_before_layernorm=int(cfg.ln_before or flavour.startswith("avs")),
                          is_post_layernorm=int(cfg.ln_post), num_tokens=cfg.tk, alpha=cfg.alpha, beta=cfg.beta, gamma=cfg.gamma)
    kw = dict(input_dim=cfg.C, output_dim=cfg.C, adapter_kind="bottleneck", dim_list=None, layer_idx=0,
              reduction_factor=cfg.r, opt=opt, use_bn=cfg.use_bn, use_gate=cfg.use_gate,
              conv_dim_in=cfg.No, conv_dim_out=cfg.N, linear_in=cfg.Co, linear_out=cfg.C)
    if flavour in ("ave", "avvp", "pretrain"):
        kw["num_tk"] = cfg.tk
    return cls(**kw)


def make_case(name: str, seed: int):
    flavour, dims = CASES[name]
    cfg = O.AdapterConfig(**{**O.FLAVOURS[flavour], **dims})
    torch.manual_seed(seed)
    ref = build_reference(flavour, cfg)
    # non-degenerate parameters: gate/gate_av != 0, tokens random, BN/LN affine perturbed
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, v in ref.named_parameters():
            if k == "gate":
                v.fill_(0.7)
            elif k == "gate_av":
                v.fill_(0.3)
            elif k == "my_tokens":
                v.copy_(torch.rand(v.shape, generator=gen))
            elif k.startswith(("bn", "ln_")):
                v.add_(0.1 * torch.randn(v.shape, generator=gen))
    state0 = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    B = BATCH
    X = torch.randn(B, cfg.N, cfg.C, generator=gen)
    Y = torch.randn(B, cfg.No, cfg.Co, generator=gen)
    dOut = torch.randn(B, cfg.N, cfg.C, generator=gen)
    dMap = torch.randn(B, cfg.N, generator=gen)
    dTmap = torch.randn(B, generator=gen) if cfg.temporal else None
    # ---- reference, train mode (BatchNorm CPU-backward workaround active, see above)
    nn.BatchNorm2d.forward = _bn_forward_workaround
    fd_err = fd_check(ref, cfg, X, Y, dOut, dMap, dTmap)
    ref.load_state_dict(state0)
    ref.train()
    Xr = X.clone().requires_grad_(True)
    Yr = Y.clone().requires_grad_(True)
    res = ref(Xr.permute(0, 2, 1).unsqueeze(-1), Yr.permute(0, 2, 1).unsqueeze(-1))
    out_r = res[0].squeeze(-1).permute(0, 2, 1)
    map_r = res[1].squeeze(1)
    loss = (out_r * dOut).sum() + (map_r * dMap).sum()
    tmap_r = None
    if cfg.temporal:
        tmap_r = res[2].reshape(B)
        loss = loss + (tmap_r * dTmap).sum()
    loss.backward()
    grads_r = {k: v.grad.detach().clone() for k, v in ref.named_parameters() if v.grad is not None}
    none_grads = sorted(k for k, v in ref.named_parameters() if v.grad is None)
    state1 = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    nn.BatchNorm2d.forward = _BN_FORWARD
    # ---- reference, eval mode (uses the just-updated running stats)
    ref.eval()
    with torch.no_grad():
        res_e = ref(X.permute(0, 2, 1).unsqueeze(-1), Y.permute(0, 2, 1).unsqueeze(-1))
    out_e = res_e[0].squeeze(-1).permute(0, 2, 1).contiguous()
    map_e = res_e[1].squeeze(1).contiguous()
    # ---- oracle
    p = {k: v.clone() for k, v in state0.items()}
    if cfg.remap == "bicubic":
        p["_bicubic"] = O.bicubic_matrix(cfg.No, cfg.N)
    out_o, map_o, tmap_o, saved = O.forward(p, X, Y, cfg, training=True)
    dX_o, dY_o, grads_o = O.backward(p, saved, cfg, dOut, dMap, dTmap, training=True)
    errs = {"fd_vs_reference_autograd(f64)": fd_err}

    def chk(tag, a, b):
        e = (a - b).abs().max().item() / max(1.0, b.abs().max().item())
        errs[tag] = e
        assert e <= 1e-4, f"{name}:{tag} mismatch {e}"

    chk("out", out_o, out_r.detach())
    chk("map", map_o, map_r.detach())
    if cfg.temporal:
        chk("tmap", tmap_o, tmap_r.detach())
    chk("dX", dX_o, Xr.grad)
    chk("dY", dY_o, Yr.grad)
    for k, gr in grads_r.items():
        assert k in grads_o, f"{name}: oracle lacks grad for {k}"
        chk("d" + k, grads_o[k].reshape(gr.shape), gr)
    extra = sorted(set(grads_o) - set(grads_r))
    assert not extra, f"{name}: oracle has grads the reference lacks: {extra}"
    for k in state1:
        if "running" in k or "num_batches" in k:
            chk("buf:" + k, p[k].float(), state1[k].float())
    pe = {k: v.clone() for k, v in state1.items()}
    if cfg.remap == "bicubic":
        pe["_bicubic"] = p["_bicubic"]
    out_oe, map_oe, _, _ = O.forward(pe, X, Y, cfg, training=False)
    chk("eval_out", out_oe, out_e)
    chk("eval_map", map_oe, map_e)
    # autograd port (used as the CPU baseline) must agree too
    pa = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in state0.items()}
    if cfg.remap == "bicubic":
        pa["_bicubic"] = p["_bicubic"]
    out_a, map_a, _ = O.forward_autograd(pa, X, Y, cfg, training=True)
    chk("port_out", out_a.detach(), out_r.detach())
    chk("port_map", map_a.detach(), map_r.detach())

    fixture = dict(
        name=name, flavour=flavour, cfg=dataclass_dict(cfg), batch=B,
        state0=state0, X=X, Y=Y, dOut=dOut, dMap=dMap, dTmap=dTmap,
        out=out_r.detach().contiguous(), map=map_r.detach().contiguous(),
        tmap=None if tmap_r is None else tmap_r.detach().contiguous(),
        dX=Xr.grad.contiguous(), dY=Yr.grad.contiguous(), grads=grads_r, none_grads=none_grads,
        buffers1={k: v for k, v in state1.items() if "running" in k or "num_batches" in k},
        eval_out=out_e, eval_map=map_e,
    )
    return fixture, errs


def fd_check(ref, cfg, X, Y, dOut, dMap, dTmap):
    """float64 central finite differences of the reference loss along random directions in X, Y and
    every parameter, against the reference's own autograd: proves the golden gradients are the true
    derivative (and not an artefact of the PyTorch CPU BatchNorm bug)."""
    import copy
    r64 = copy.deepcopy(ref).double().train()
    gen = torch.Generator().manual_seed(7)
    Xd, Yd = X.double(), Y.double()
    params = [v for v in r64.parameters()]

    def loss_fn(Xv, Yv):
        res = r64(Xv.permute(0, 2, 1).unsqueeze(-1), Yv.permute(0, 2, 1).unsqueeze(-1))
        l = (res[0].squeeze(-1).permute(0, 2, 1) * dOut.double()).sum() + (res[1].squeeze(1) * dMap.double()).sum()
        if cfg.temporal:
            l = l + (res[2].reshape(-1) * dTmap.double()).sum()
        return l

    Xr, Yr = Xd.clone().requires_grad_(True), Yd.clone().requires_grad_(True)
    loss_fn(Xr, Yr).backward()
    VX = torch.randn(X.shape, generator=gen, dtype=torch.float64)
    VY = torch.randn(Y.shape, generator=gen, dtype=torch.float64)
    VP = [torch.randn(v.shape, generator=gen, dtype=torch.float64) for v in params]
    ana = (Xr.grad * VX).sum() + (Yr.grad * VY).sum() + sum((v.grad * d).sum() for v, d in zip(params, VP) if v.grad is not None)
    eps = 1e-6
    with torch.no_grad():
        vals = []
        for sgn in (+1, -1):
            for v, d in zip(params, VP):
                v.add_(sgn * eps * d)
            vals.append(loss_fn(Xd + sgn * eps * VX, Yd + sgn * eps * VY))
            for v, d in zip(params, VP):
                v.sub_(sgn * eps * d)
    fd = (vals[0] - vals[1]) / (2 * eps)
    err = abs(fd.item() - ana.item()) / max(1.0, abs(fd.item()))
    assert err < 1e-5, f"reference autograd disagrees with finite differences: {ana.item()} vs {fd.item()}"
    return err


STACK_STAGES = [dict(layers=1, Nv=16, Cv=32, Na=36, Ca=16), dict(layers=2, Nv=9, Cv=48, Na=16, Ca=32)]


def make_stack_fixture(seed=500):
    """a-10: the AVE layer loop (net_trans.py:880-916) over reference adapters with identity backbone
    blocks, two tiny stages.  Outputs, last spatial maps, input gradients and every parameter gradient."""
    cls = load_reference_class("ave")
    opt = SimpleNamespace(is_multimodal=1, num_conv_group=2, is_before_layernorm=1, is_post_layernorm=1, num_tokens=4)
    torch.manual_seed(seed)
    hidden, hidden_a, conv, conv_a = [], [], [], []
    for st in STACK_STAGES:
        for _ in range(st["layers"]):
            hidden.append(st["Cv"]); hidden_a.append(st["Ca"]); conv.append(st["Nv"]); conv_a.append(st["Na"])

    def audio(i):
        return cls(input_dim=hidden_a[i], output_dim=hidden_a[i], adapter_kind="bottleneck", dim_list=hidden_a, layer_idx=i,
                   reduction_factor=8, opt=opt, use_bn=True, use_gate=True, num_tk=4, conv_dim_in=conv[i],
                   conv_dim_out=conv_a[i], linear_in=hidden[i], linear_out=hidden_a[i])

    def visual(i):
        return cls(input_dim=hidden[i], output_dim=hidden[i], adapter_kind="bottleneck", dim_list=hidden, layer_idx=i,
                   reduction_factor=8, opt=opt, use_bn=True, use_gate=True, num_tk=4, conv_dim_in=conv_a[i],
                   conv_dim_out=conv[i], linear_in=hidden_a[i], linear_out=hidden[i])

    n = len(hidden)
    net = nn.Module()
    net.audio_adapter_blocks_p1 = nn.ModuleList([audio(i) for i in range(n)])
    net.vis_adapter_blocks_p1 = nn.ModuleList([visual(i) for i in range(n)])
    net.audio_adapter_blocks_p2 = nn.ModuleList([audio(i) for i in range(n)])
    net.vis_adapter_blocks_p2 = nn.ModuleList([visual(i) for i in range(n)])
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, v in net.named_parameters():
            if k.endswith(".gate"):
                v.fill_(0.7)
            elif k.endswith(".gate_av"):
                v.fill_(0.3)
    state0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    BT = 10
    feats = [(torch.randn(BT, st["Nv"], st["Cv"], generator=gen), torch.randn(BT, st["Na"], st["Ca"], generator=gen))
             for st in STACK_STAGES]
    cots = [(torch.randn(BT, st["Nv"], st["Cv"], generator=gen), torch.randn(BT, st["Na"], st["Ca"], generator=gen))
            for st in STACK_STAGES]
    mcots = (torch.randn(BT, 1, STACK_STAGES[-1]["Nv"], generator=gen), torch.randn(BT, 1, STACK_STAGES[-1]["Na"], generator=gen))
    nn.BatchNorm2d.forward = _bn_forward_workaround
    net.train()
    leaves = [(a.clone().requires_grad_(True), b.clone().requires_grad_(True)) for a, b in feats]
    view = lambda f: f.permute(0, 2, 1).unsqueeze(-1)
    outs, idx = [], 0
    for st, (f_v, f_a) in zip(STACK_STAGES, leaves):
        for _ in range(st["layers"]):
            a_res, _ = net.audio_adapter_blocks_p1[idx](view(f_a), view(f_v))
            v_res, _ = net.vis_adapter_blocks_p1[idx](view(f_v), view(f_a))
            f_v = f_v + v_res.squeeze(-1).permute(0, 2, 1)           # frozen Swin / HTS-AT half-blocks: identity stand-ins
            f_a = f_a + a_res.squeeze(-1).permute(0, 2, 1)
            a_res, a_map = net.audio_adapter_blocks_p2[idx](view(f_a), view(f_v))
            v_res, v_map = net.vis_adapter_blocks_p2[idx](view(f_v), view(f_a))
            f_v = f_v + v_res.squeeze(-1).permute(0, 2, 1)
            f_a = f_a + a_res.squeeze(-1).permute(0, 2, 1)
            idx += 1
        outs.append((f_v, f_a))
    tensors = [t for pr in outs for t in pr] + [v_map, a_map]
    grads = [g for pr in cots for g in pr] + [mcots[0], mcots[1]]
    torch.autograd.backward(tensors, grads)
    nn.BatchNorm2d.forward = _BN_FORWARD
    fx = dict(stages=STACK_STAGES, state0=state0, feats=feats, cots=cots, mcots=mcots,
              outs=[(a.detach().clone(), b.detach().clone()) for a, b in outs],
              maps=(v_map.detach().clone(), a_map.detach().clone()),
              dfeats=[(a.grad.clone(), b.grad.clone()) for a, b in leaves],
              grads={k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None})
    return fx


def dataclass_dict(cfg):
    import dataclasses
    return dataclasses.asdict(cfg)


def main():
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    report = {}
    for i, name in enumerate(CASES):
        fx, errs = make_case(name, seed=100 + i)
        path = os.path.join(outdir, name + ".pt")
        torch.save(fx, path)
        worst = max(errs.values())
        report[name] = dict(worst_rel_err=worst, n_checked=len(errs), none_grads=fx["none_grads"],
                            bytes=os.path.getsize(path))
        print(f"{name:22s} worst |oracle-reference| = {worst:.2e} over {len(errs)} tensors; "
              f"no-grad params: {fx['none_grads']}; {os.path.getsize(path)} B")
    fx = make_stack_fixture()
    path = os.path.join(outdir, "stack_2stage.pt")
    torch.save(fx, path)
    report["stack_2stage"] = dict(bytes=os.path.getsize(path), n_param_grads=len(fx["grads"]))
    print(f"stack_2stage           AVE layer loop over 12 reference adapters, identity backbone; {os.path.getsize(path)} B")
    json.dump(report, open(os.path.join(outdir, "VALIDATION.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
