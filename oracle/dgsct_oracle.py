"""CPU oracle for the DG-SCT cross-modal adapter hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  The product path (``dg-sct_amd``) never
imports anything under ``oracle/`` and fails loudly when the HIP library is missing.

Parity status: PINNED.  ``oracle/make_golden.py`` (run in the build container, where
``/root/reference`` exists) extracts the reference ``VisualAdapter`` classes from the reference
sources, runs them on CPU and checks this restatement against them for every flavour, forward
and backward, train and eval; the resulting vectors are committed under ``tests/golden/``.

What is restated (token-major: X=[BT,N,C] own modality, Y=[BT,No,Co] other modality):

  reference (DG-SCT/AVE/nets/net_trans.py)          here
  -----------------------------------------------   ----------------------------------
  :553-555  conv_adapter + fc cross-modal remap      ``remap``        (F1)
  :572-580  latent tokens attend to remapped tokens  ``tok``/``P1``   (F2)
  :583-589  X attends to latent tokens, gate_av      ``X1``/``P2``    (F3)
  :592-598  channel gate                             ``ch``           (F4-F6)
  :601-608  spatial gate + softmax(tanh) map         ``s``/``map``    (F7)
  :611-612  modulation (alpha, beta[, gamma])        ``X2``           (F8)
  :627      ln_before                                ``X3``
  :629-643  grouped bottleneck + BatchNorm2d         ``Zp,Z,Op,O``    (F9-F10)
  :668-671  ln_post, gate                            ``out``          (F11)

Flavour deltas (AVVP mgn.py:162-414, AVS-S4/MS3 PVT_AVSModel.py:90-316/90-300,
AVQA net_avst.py:27-218, pretrain/few/zero-shot net_trans.py:343-600) are flags of
``AdapterConfig``.  ``forward`` is written as the explicit kernel-level decomposition the HIP
library follows and returns every intermediate; ``backward`` is the hand-derived gradient (no
autograd) that the HIP backward follows step by step.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclasses.dataclass
class AdapterConfig:
    N: int                      # own-modality tokens       (conv_dim_out)
    C: int                      # own-modality width         (input_dim == output_dim == linear_out)
    No: int                     # other-modality tokens      (conv_dim_in)
    Co: int                     # other-modality width       (linear_in)
    tk: int = 32                # latent tokens              (num_tk / opt.num_tokens)
    r: int = 8                  # reduction_factor           (opt.Adapter_downsample)
    g: int = 2                  # opt.num_conv_group
    use_bn: bool = True
    use_gate: bool = True
    ln_before: bool = True      # opt.is_before_layernorm (ignored by the AVS flavours)
    ln_post: bool = True        # opt.is_post_layernorm
    gate_before_ln_post: bool = False   # AVS-S4/MS3 order: gate then ln_post
    remap: str = "conv"         # "conv" (conv_adapter+fc) | "bicubic" (AVS-S4: fc then resize)
    alpha: float = 0.3
    beta: float = 0.05
    gamma: float = 0.0          # temporal gate weight; only used when temporal=True
    temporal: bool = False      # pretrain/few/zero-shot flavour: + gamma * sigmoid(temporal_gated(a))
    T: int = 10
    eps: float = 1e-5
    bn_momentum: float = 0.1

    @property
    def d(self) -> int:
        return self.C // 2

    @property
    def ds(self) -> int:
        return self.C // self.r

    def remap_order(self) -> str:
        """'A': (Wn.Y).Wc^T  |  'B': Wn.(Y.Wc^T) -- whichever is cheaper."""
        a = self.N * self.No * self.Co + self.N * self.Co * self.C
        b = self.No * self.Co * self.C + self.N * self.No * self.C
        return "A" if a <= b else "B"


FLAVOURS = {
    # name: overrides relative to the AVE defaults above
    "ave": dict(),
    "avvp": dict(),
    "avs_s4": dict(remap="bicubic", ln_before=False, gate_before_ln_post=True, T=5),
    "avs_ms3": dict(alpha=0.2, beta=0.1, ln_before=False, gate_before_ln_post=True, T=5),
    "avqa": dict(tk=2, g=4, use_bn=False),
    "pretrain": dict(alpha=0.3, beta=0.01, gamma=0.05, temporal=True),
}


def bicubic_matrix(No: int, N: int) -> Tensor:
    """Dense [N, No] operator equal to F.interpolate(mode='bicubic', align_corners=False) from a
    sqrt(No)^2 grid to a sqrt(N)^2 grid (AVS-S4 remap, PVT_AVSModel.py:190-197)."""
    hi, ho = int(math.isqrt(No)), int(math.isqrt(N))
    assert hi * hi == No and ho * ho == N
    eye = torch.eye(No, dtype=torch.float64).view(No, 1, hi, hi)
    out = F.interpolate(eye, size=[ho, ho], mode="bicubic")          # [No,1,ho,ho]
    return out.view(No, N).t().contiguous().float()


# ----------------------------------------------------------------------------- helpers
def _ln(x: Tensor, w: Tensor, b: Tensor, eps: float):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    xh = (x - mu) * rstd
    return xh * w + b, xh, rstd


def _ln_bwd(dy: Tensor, xh: Tensor, rstd: Tensor, w: Tensor):
    g = dy * w
    dx = rstd * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
    dw = (dy * xh).reshape(-1, xh.shape[-1]).sum(0)
    db = dy.reshape(-1, xh.shape[-1]).sum(0)
    return dx, dw, db


def _groupmm(x: Tensor, w: Tensor, g: int) -> Tensor:
    """x [..., Cin], w [Cout, Cin/g] (1x1 grouped conv weight) -> [..., Cout]."""
    cin = x.shape[-1]
    cout = w.shape[0]
    xs = x.reshape(*x.shape[:-1], g, cin // g)
    ws = w.reshape(g, cout // g, cin // g)
    return torch.einsum("...gi,goi->...go", xs, ws).reshape(*x.shape[:-1], cout)


def _groupmm_bwd(dy: Tensor, x: Tensor, w: Tensor, g: int):
    cin = x.shape[-1]
    cout = w.shape[0]
    xs = x.reshape(-1, g, cin // g)
    dys = dy.reshape(-1, g, cout // g)
    ws = w.reshape(g, cout // g, cin // g)
    dx = torch.einsum("rgo,goi->rgi", dys, ws).reshape(x.shape)
    dw = torch.einsum("rgo,rgi->goi", dys, xs).reshape(w.shape)
    return dx, dw


# ----------------------------------------------------------------------------- forward
def _q8(x: Tensor, per_tensor: bool) -> Tensor:
    """OCP e4m3 quantisation of one MFMA operand as the fp8 projections do it (csrc/gemm_fp8.hip; BASELINE configs[4]): weights with a
    per-tensor scale max|w| / 448, activations converted directly (saturating at +-448).  Returns the DE-quantised fp32 values."""
    if per_tensor:
        sc = 448.0 / x.abs().max().clamp_min(1e-30)
        return (x * sc).clamp(-448, 448).to(torch.float8_e4m3fn).float() / sc
    return x.clamp(-448, 448).to(torch.float8_e4m3fn).float()


def forward(p: Dict[str, Tensor], X: Tensor, Y: Tensor, cfg: AdapterConfig, training: bool = True,
            update_running: bool = True, fp8: bool = False) -> Tuple[Tensor, Tensor, Optional[Tensor], Dict[str, Tensor]]:
    """p: parameter/buffer dict keyed by the reference state_dict names (conv weights squeezed to
    2-D views is NOT required: 4-D [out,in,1,1] tensors are accepted).  Returns
    (out [BT,N,C], map [BT,N], tmap [BT] | None, saved-intermediates)."""
    B, N, C = X.shape
    No, Co = Y.shape[1], Y.shape[2]
    assert (N, C, No, Co) == (cfg.N, cfg.C, cfg.No, cfg.Co)
    s: Dict[str, Tensor] = {}
    Wc, bc = p["fc.weight"], p["fc.bias"]
    if cfg.remap == "conv":
        Wn = p["conv_adapter.weight"].reshape(N, No)
        bn = p["conv_adapter.bias"]
        rowb, colb = bn, Wc.sum(1)             # rank-1 bias: bn[m] * sum_co Wc[c,co]
        colb2 = bc
    else:
        Wn = p["_bicubic"]
        rowb, colb = Wn.sum(1), bc             # rank-1 bias: rowsum(Wbic)[m] * bc[c]
        colb2 = torch.zeros_like(bc)
    order = cfg.remap_order()
    s["order"] = order
    # F1 remap -------------------------------------------------------------- net_trans.py:553-555
    if order == "A":
        T1 = torch.einsum("mn,bnk->bmk", Wn, Y)                       # [B,N,Co]
        s["T1"] = T1
        # (fp8: the three weight-stationary forward projections with e4m3 operands -- fc in this association, fc_affine_video_1 / 2;
        #  the saved intermediates then carry that perturbation and backward() differentiates the UN-quantised graph at them, which is
        #  what the device does: its backward products use the bf16 weights and activations)
        Yp = (_q8(T1.bfloat16().float(), False) @ _q8(Wc, True).t()) if fp8 else T1 @ Wc.t()
    else:
        T2t = torch.einsum("ck,bnk->bcn", Wc, Y)                      # [B,C,No]  (= (Y Wc^T)^T)
        s["T2t"] = T2t
        Yp = torch.einsum("mn,bcn->bmc", Wn, T2t)
    Yp = Yp + rowb[None, :, None] * colb[None, None, :] + colb2
    s["Yp"] = Yp
    # F2 latent tokens <- remapped tokens ------------------------------------------ :572-580
    T0 = p["my_tokens"]
    S1 = torch.einsum("tc,bnc->btn", T0, Yp)
    P1 = torch.softmax(S1, dim=-1)
    tok = T0[None] + P1 @ Yp                                          # [B,tk,C]
    a = Yp.mean(1)                                                    # [B,C]            :592
    s.update(P1=P1, tok=tok, a=a)
    # F3 X <- latent tokens ----------------------------------------------------------- :583-589
    S2 = X @ tok.transpose(1, 2)                                      # [B,N,tk]
    P2 = torch.softmax(S2, dim=-1)
    gate_av = p["gate_av"]
    X1 = X + gate_av * (P2 @ tok)
    s.update(P2=P2, X1=X1)
    # F4-F6 channel gate ---------------------------------------------------------------- :593-598
    aq1 = F.relu(F.linear(a, p["fc_affine_audio_1.weight"], p["fc_affine_audio_1.bias"]))
    aq2 = F.relu(F.linear(a, p["fc_affine_audio_2.weight"], p["fc_affine_audio_2.bias"]))
    if fp8:
        vq1 = F.relu(_q8(X1.bfloat16().float(), False) @ _q8(p["fc_affine_video_1.weight"], True).t() + p["fc_affine_video_1.bias"])
    else:
        vq1 = F.relu(F.linear(X1, p["fc_affine_video_1.weight"], p["fc_affine_video_1.bias"]))
    mvq1 = vq1.mean(1)                                                # [B,C]
    m1 = aq1 * mvq1
    q = F.relu(F.linear(m1, p["fc_affine_bottleneck.weight"], p["fc_affine_bottleneck.bias"]))
    ch = torch.sigmoid(F.linear(q, p["fc_affine_v_c_att.weight"], p["fc_affine_v_c_att.bias"]))
    s.update(aq1=aq1, aq2=aq2, vq1=vq1, mvq1=mvq1, m1=m1, q=q, ch=ch)
    # F7 spatial gate + map ------------------------------------------------------------- :601-608
    Xc = X1 * (1 + ch[:, None, :])
    if fp8:
        vq2 = F.relu(_q8(Xc.bfloat16().float(), False) @ _q8(p["fc_affine_video_2.weight"], True).t() + p["fc_affine_video_2.bias"])
    else:
        vq2 = F.relu(F.linear(Xc, p["fc_affine_video_2.weight"], p["fc_affine_video_2.bias"]))
    ws, bs = p["fc_affine_v_s_att.weight"].reshape(-1), p["fc_affine_v_s_att.bias"]
    sl = (vq2 * (aq2 * ws)[:, None, :]).sum(-1) + bs                  # [B,N]
    sg = torch.sigmoid(sl)
    amap = torch.softmax(torch.tanh(sl), dim=-1)                      # [B,N]
    s.update(Xc=Xc, vq2=vq2, sl=sl, sg=sg, map=amap)
    # temporal gate (pretrain flavour only) ----------------- pretrain/nets/net_trans.py:532-548
    tg = None
    if cfg.temporal:
        wt, bt = p["temporal_gated.0.weight"].reshape(-1), p["temporal_gated.0.bias"]
        tg = torch.sigmoid(a @ wt + bt)                               # [B]
        s["tg"] = tg
    # F8 modulation + ln_before ---------------------------------------------------------- :611-627
    mod = cfg.alpha * ch[:, None, :] + cfg.beta * sg[:, :, None] + (1 - cfg.alpha)
    if cfg.temporal:
        mod = mod + cfg.gamma * tg[:, None, None]
    X2 = X1 * mod
    if cfg.ln_before:
        X3, xh_b, rstd_b = _ln(X2, p["ln_before.weight"], p["ln_before.bias"], cfg.eps)
        s.update(xh_b=xh_b, rstd_b=rstd_b)
    else:
        X3 = X2
    s.update(mod=mod, X2=X2, X3=X3)
    # F9-F10 grouped bottleneck + BN ------------------------------------------------------ :629-643
    Wd = p["down_sampler.weight"].reshape(cfg.ds, C // cfg.g)
    Wu = p["up_sampler.weight"].reshape(C, cfg.ds // cfg.g)
    Zp = _groupmm(X3, Wd, cfg.g)                                      # [B,N,ds]
    R = B * N

    def bn(x, name):
        w, b = p[name + ".weight"], p[name + ".bias"]
        if training:
            xf = x.reshape(R, -1)
            mu = xf.mean(0)
            var = ((xf - mu) ** 2).mean(0)
            if update_running:
                with torch.no_grad():
                    m = cfg.bn_momentum
                    p[name + ".running_mean"].mul_(1 - m).add_(m * mu)
                    p[name + ".running_var"].mul_(1 - m).add_(m * var * (R / max(R - 1, 1)))
                    p[name + ".num_batches_tracked"].add_(1)
        else:
            mu, var = p[name + ".running_mean"], p[name + ".running_var"]
        rstd = torch.rsqrt(var + cfg.eps)
        xh = (x - mu) * rstd
        return xh * w + b, xh, rstd

    if cfg.use_bn:
        Zb, zh, rstd1 = bn(Zp, "bn1")
        s.update(zh=zh, rstd1=rstd1)
    else:
        Zb = Zp
    Z = F.relu(Zb)
    Op = _groupmm(Z, Wu, cfg.g)                                       # [B,N,C]
    if cfg.use_bn:
        O, oh, rstd2 = bn(Op, "bn2")
        s.update(oh=oh, rstd2=rstd2)
    else:
        O = Op
    s.update(Zp=Zp, Z=Z, Op=Op, O=O)
    # F11 ln_post / gate ------------------------------------------------------------------- :668-671
    gate = p["gate"] if cfg.use_gate else None
    if cfg.gate_before_ln_post:
        G = O * gate if gate is not None else O
        if cfg.ln_post:
            out, xh_p, rstd_p = _ln(G, p["ln_post.weight"], p["ln_post.bias"], cfg.eps)
            s.update(xh_p=xh_p, rstd_p=rstd_p)
        else:
            out = G
    else:
        if cfg.ln_post:
            L, xh_p, rstd_p = _ln(O, p["ln_post.weight"], p["ln_post.bias"], cfg.eps)
            s.update(xh_p=xh_p, rstd_p=rstd_p)
        else:
            L = O
        s["L"] = L
        out = L * gate if gate is not None else L
    s["X"], s["Y"] = X, Y
    return out, amap, tg, s


# ----------------------------------------------------------------------------- backward
def backward(p: Dict[str, Tensor], s: Dict[str, Tensor], cfg: AdapterConfig, dOut: Tensor,
             dMap: Optional[Tensor] = None, dTmap: Optional[Tensor] = None, training: bool = True,
             masks: Optional[Dict[str, Tensor]] = None):
    """Hand-derived gradient of ``forward``.  Returns (dX, dY, grads) where grads is keyed by
    reference parameter names (4-D conv weights keep their 4-D shape).

    ``masks`` (test hook): boolean ReLU masks to use INSTEAD of ``s[name] > 0`` for name in {"Z", "vq2", "q", "vq1", "aq1",
    "aq2"}.  The bf16 parity tests feed the masks the device run actually took: a unit whose pre-activation is within bf16
    rounding of zero lands on either side, and each such flip changes that unit's whole gradient contribution -- with the
    masks pinned that noise is gone and what remains is the arithmetic error of the kernels."""
    masks = masks or {}
    mk = lambda name: masks[name] if name in masks else (s[name] > 0)          # noqa: E731
    X, Y = s["X"], s["Y"]
    B, N, C = X.shape
    No, Co = Y.shape[1], Y.shape[2]
    R = B * N
    g: Dict[str, Tensor] = {}
    gate = p["gate"] if cfg.use_gate else None
    # B11 ---- ln_post / gate
    if cfg.gate_before_ln_post:
        if cfg.ln_post:
            dG, g["ln_post.weight"], g["ln_post.bias"] = _ln_bwd(dOut, s["xh_p"], s["rstd_p"], p["ln_post.weight"])
        else:
            dG = dOut
        if gate is not None:
            if cfg.ln_post:
                # LayerNorm is scale invariant: d/dgate of LN(gate * O) cancels down to its eps term, and summed from
                # the fp32 dG this scalar is rounding noise (+-3e-4 at 2 M elements).  As the yardstick of the fp32
                # tests this ONE reduction is therefore evaluated in float64 from the fp32 forward values (same formula).
                O64, w64 = s["O"].double(), p["ln_post.weight"].double()
                G64 = O64 * gate.double()
                mu64 = G64.mean(-1, keepdim=True)
                rs64 = (G64.var(-1, unbiased=False, keepdim=True) + cfg.eps).rsqrt()
                xh64 = (G64 - mu64) * rs64
                dyw = dOut.double() * w64
                dG64 = rs64 * (dyw - dyw.mean(-1, keepdim=True) - xh64 * (dyw * xh64).mean(-1, keepdim=True))
                g["gate"] = (dG64 * O64).sum().reshape(1).to(dOut.dtype)
            else:
                g["gate"] = (dG * s["O"]).sum().reshape(1)
            dO = dG * gate
        else:
            dO = dG
    else:
        if gate is not None:
            g["gate"] = (dOut * s["L"]).sum().reshape(1)
            dL = dOut * gate
        else:
            dL = dOut
        if cfg.ln_post:
            dO, g["ln_post.weight"], g["ln_post.bias"] = _ln_bwd(dL, s["xh_p"], s["rstd_p"], p["ln_post.weight"])
        else:
            dO = dL

    def bn_bwd(dy, xh, rstd, name):
        w = p[name + ".weight"]
        dyf, xhf = dy.reshape(R, -1), xh.reshape(R, -1)
        dw = (dyf * xhf).sum(0)
        db = dyf.sum(0)
        if training:
            dx = w * rstd * (dy - db / R - xh * (dw / R))
        else:
            dx = dy * (w * rstd)
        g[name + ".weight"], g[name + ".bias"] = dw, db
        return dx

    # B10 ---- BN2, up projection
    dOp = bn_bwd(dO, s["oh"], s["rstd2"], "bn2") if cfg.use_bn else dO
    Wd = p["down_sampler.weight"].reshape(cfg.ds, C // cfg.g)
    Wu = p["up_sampler.weight"].reshape(C, cfg.ds // cfg.g)
    dZ, dWu = _groupmm_bwd(dOp, s["Z"], Wu, cfg.g)
    g["up_sampler.weight"] = dWu.reshape(p["up_sampler.weight"].shape)
    # B9 ---- relu, BN1, down projection
    dZb = dZ * mk("Z")
    dZp = bn_bwd(dZb, s["zh"], s["rstd1"], "bn1") if cfg.use_bn else dZb
    dX3, dWd = _groupmm_bwd(dZp, s["X3"], Wd, cfg.g)
    g["down_sampler.weight"] = dWd.reshape(p["down_sampler.weight"].shape)
    # B8 ---- ln_before, modulation
    if cfg.ln_before:
        dX2, g["ln_before.weight"], g["ln_before.bias"] = _ln_bwd(dX3, s["xh_b"], s["rstd_b"], p["ln_before.weight"])
    else:
        dX2 = dX3
    X1, ch, sg = s["X1"], s["ch"], s["sg"]
    dX1 = dX2 * s["mod"]
    dmod = dX2 * X1
    dch = cfg.alpha * dmod.sum(1)                                     # [B,C]
    dsg = cfg.beta * dmod.sum(2)                                      # [B,N]
    da = torch.zeros_like(s["a"])
    if cfg.temporal:
        tg = s["tg"]
        dtg = cfg.gamma * dmod.sum((1, 2))
        if dTmap is not None:
            dtg = dtg + dTmap
        dpre_t = dtg * tg * (1 - tg)
        wt = p["temporal_gated.0.weight"].reshape(-1)
        g["temporal_gated.0.weight"] = (dpre_t[:, None] * s["a"]).sum(0).reshape(p["temporal_gated.0.weight"].shape)
        g["temporal_gated.0.bias"] = dpre_t.sum().reshape(1)
        da = da + dpre_t[:, None] * wt[None, :]
    # B7 ---- spatial gate
    dsl = dsg * sg * (1 - sg)
    if dMap is not None:
        amap = s["map"]
        dt = amap * (dMap - (amap * dMap).sum(-1, keepdim=True))
        dsl = dsl + dt * (1 - torch.tanh(s["sl"]) ** 2)
    ws = p["fc_affine_v_s_att.weight"].reshape(-1)
    aq2, vq2 = s["aq2"], s["vq2"]
    u = (dsl[:, :, None] * vq2).sum(1)                                # [B,d]
    g["fc_affine_v_s_att.bias"] = dsl.sum().reshape(1)
    g["fc_affine_v_s_att.weight"] = (u * aq2).sum(0).reshape(p["fc_affine_v_s_att.weight"].shape)
    daq2 = u * ws
    dvq2 = dsl[:, :, None] * (aq2 * ws)[:, None, :] * mk("vq2")       # [B,N,d]
    Wv2 = p["fc_affine_video_2.weight"]
    dXc = dvq2 @ Wv2                                                  # [B,N,C]
    g["fc_affine_video_2.weight"] = dvq2.reshape(R, -1).t() @ s["Xc"].reshape(R, C)
    g["fc_affine_video_2.bias"] = dvq2.reshape(R, -1).sum(0)
    dX1 = dX1 + dXc * (1 + ch[:, None, :])
    dch = dch + (dXc * X1).sum(1)
    # B6 ---- channel gate head
    dpre_c = dch * ch * (1 - ch)                                      # [B,C]
    g["fc_affine_v_c_att.weight"] = dpre_c.t() @ s["q"]
    g["fc_affine_v_c_att.bias"] = dpre_c.sum(0)
    dq = (dpre_c @ p["fc_affine_v_c_att.weight"]) * mk("q")           # [B,d]
    g["fc_affine_bottleneck.weight"] = dq.t() @ s["m1"]
    g["fc_affine_bottleneck.bias"] = dq.sum(0)
    dm1 = dq @ p["fc_affine_bottleneck.weight"]                       # [B,C]
    daq1 = dm1 * s["mvq1"]
    dmvq1 = dm1 * s["aq1"]
    # B5 ---- video query 1
    dvq1 = (dmvq1 / N)[:, None, :] * mk("vq1")                        # [B,N,C]
    dX1 = dX1 + dvq1 @ p["fc_affine_video_1.weight"]
    g["fc_affine_video_1.weight"] = dvq1.reshape(R, C).t() @ X1.reshape(R, C)
    g["fc_affine_video_1.bias"] = dvq1.reshape(R, C).sum(0)
    # B4 ---- audio queries
    dpa1 = daq1 * mk("aq1")
    dpa2 = daq2 * mk("aq2")
    g["fc_affine_audio_1.weight"] = dpa1.t() @ s["a"]
    g["fc_affine_audio_1.bias"] = dpa1.sum(0)
    g["fc_affine_audio_2.weight"] = dpa2.t() @ s["a"]
    g["fc_affine_audio_2.bias"] = dpa2.sum(0)
    da = da + dpa1 @ p["fc_affine_audio_1.weight"] + dpa2 @ p["fc_affine_audio_2.weight"]
    # B3 ---- X <- tokens attention
    tok, P2, P1, Yp = s["tok"], s["P2"], s["P1"], s["Yp"]
    gate_av = p["gate_av"]
    U = dX1 @ tok.transpose(1, 2)                                     # [B,N,tk] = dR.tok^T / gate_av
    g["gate_av"] = (P2 * U).sum().reshape(1)
    dP2 = gate_av * U
    dS2 = P2 * (dP2 - (P2 * dP2).sum(-1, keepdim=True))
    dX = dX1 + dS2 @ tok
    dtok = gate_av * (P2.transpose(1, 2) @ dX1) + dS2.transpose(1, 2) @ X      # [B,tk,C]
    # B2 ---- tokens <- remapped tokens attention
    T0 = p["my_tokens"]
    dP1 = dtok @ Yp.transpose(1, 2)                                   # [B,tk,N]
    dS1 = P1 * (dP1 - (P1 * dP1).sum(-1, keepdim=True))
    g["my_tokens"] = dtok.sum(0) + torch.einsum("btn,bnc->tc", dS1, Yp)
    dYp = P1.transpose(1, 2) @ dtok + torch.einsum("btn,tc->bnc", dS1, T0) + (da / N)[:, None, :]
    # B1 ---- remap
    Wc = p["fc.weight"]
    if cfg.remap == "conv":
        Wn = p["conv_adapter.weight"].reshape(N, No)
        bn_ = p["conv_adapter.bias"]
        wcsum = Wc.sum(1)
        g["fc.bias"] = dYp.sum((0, 1))
        g["conv_adapter.bias"] = torch.einsum("bmc,c->m", dYp, wcsum)
        dwcsum = torch.einsum("bmc,m->c", dYp, bn_)
    else:
        Wn = p["_bicubic"]
        g["fc.bias"] = torch.einsum("bmc,m->c", dYp, Wn.sum(1))
        dwcsum = None
    if s["order"] == "A":
        dT1 = dYp @ Wc                                                # [B,N,Co]
        dWc = torch.einsum("bmc,bmk->ck", dYp, s["T1"])
        dY = torch.einsum("mn,bmk->bnk", Wn, dT1)
        dWn = torch.einsum("bmk,bnk->mn", dT1, Y)
    else:
        dT2t = torch.einsum("bmc,mn->bcn", dYp, Wn)                   # [B,C,No]
        dWn = torch.einsum("bmc,bcn->mn", dYp, s["T2t"])
        dY = torch.einsum("bcn,ck->bnk", dT2t, Wc)
        dWc = torch.einsum("bcn,bnk->ck", dT2t, Y)
    if dwcsum is not None:
        dWc = dWc + dwcsum[:, None]
    g["fc.weight"] = dWc
    if cfg.remap == "conv":
        g["conv_adapter.weight"] = dWn.reshape(p["conv_adapter.weight"].shape)
    return dX, dY, g


# ----------------------------------------------------------------------------- parameter factory
def param_shapes(cfg: AdapterConfig, flavour: str = "ave") -> Dict[str, Tuple[int, ...]]:
    """state_dict names/shapes of the reference module (probe-confirmed)."""
    C, d, ds, g = cfg.C, cfg.d, cfg.ds, cfg.g
    sh = {
        "conv_adapter.weight": (cfg.N, cfg.No, 1, 1), "conv_adapter.bias": (cfg.N,),
        "fc.weight": (C, cfg.Co), "fc.bias": (C,),
        "fc_affine_audio_1.weight": (C, C), "fc_affine_audio_1.bias": (C,),
        "fc_affine_video_1.weight": (C, C), "fc_affine_video_1.bias": (C,),
        "fc_affine_bottleneck.weight": (d, C), "fc_affine_bottleneck.bias": (d,),
        "fc_affine_video_2.weight": (d, C), "fc_affine_video_2.bias": (d,),
        "fc_affine_audio_2.weight": (d, C), "fc_affine_audio_2.bias": (d,),
        "fc_affine_v_s_att.weight": (1, d), "fc_affine_v_s_att.bias": (1,),
        "fc_affine_v_c_att.weight": (C, d), "fc_affine_v_c_att.bias": (C,),
        "my_tokens": (cfg.tk, C), "gate_av": (1,),
        "down_sampler.weight": (ds, C // g, 1, 1), "up_sampler.weight": (C, ds // g, 1, 1),
    }
    if cfg.use_gate:
        sh["gate"] = (1,)
    if flavour in ("ave", "avvp", "pretrain"):
        sh["gate_tk"] = (1,)
    if flavour in ("avvp", "pretrain"):
        sh["fc_caption.weight"] = (192, 512)
        sh["fc_caption.bias"] = (192,)
    if flavour in ("avvp", "avs_s4", "avs_ms3", "pretrain"):
        sh["temporal_gated.0.weight"] = (1, C)
        sh["temporal_gated.0.bias"] = (1,)
    if cfg.use_bn:
        for n, c in (("bn1", ds), ("bn2", C)):
            sh[n + ".weight"] = (c,)
            sh[n + ".bias"] = (c,)
            sh[n + ".running_mean"] = (c,)
            sh[n + ".running_var"] = (c,)
            sh[n + ".num_batches_tracked"] = ()
    if cfg.ln_before or flavour in ("avs_s4", "avs_ms3"):
        sh["ln_before.weight"] = (C,)
        sh["ln_before.bias"] = (C,)
    if cfg.ln_post:
        sh["ln_post.weight"] = (C,)
        sh["ln_post.bias"] = (C,)
    return sh


def random_params(cfg: AdapterConfig, flavour: str = "ave", seed: int = 0, scale: float = 1.0) -> Dict[str, Tensor]:
    """Deterministic, non-degenerate parameters (gate/gate_av != 0) for tests and benchmarks."""
    gen = torch.Generator().manual_seed(seed)
    p: Dict[str, Tensor] = {}
    for k, shp in param_shapes(cfg, flavour).items():
        if k.endswith("num_batches_tracked"):
            p[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var"):
            p[k] = torch.ones(shp)
        elif k.endswith("running_mean"):
            p[k] = torch.zeros(shp)
        elif k in ("gate", "gate_av"):
            p[k] = torch.full(shp, 0.7 if k == "gate" else 0.3)
        elif k == "gate_tk":
            p[k] = torch.ones(shp)
        elif k == "my_tokens":
            p[k] = torch.rand(shp, generator=gen)
        elif k.startswith(("bn", "ln_")) and k.endswith("weight"):
            p[k] = 1.0 + 0.1 * torch.randn(shp, generator=gen)
        elif k.startswith(("bn", "ln_")) and k.endswith("bias"):
            p[k] = 0.1 * torch.randn(shp, generator=gen)
        else:
            fan_in = shp[1] if len(shp) > 1 else shp[0]
            p[k] = scale * torch.randn(shp, generator=gen) / math.sqrt(max(fan_in, 1))
    if cfg.remap == "bicubic":
        p["_bicubic"] = bicubic_matrix(cfg.No, cfg.N)
    return p


# ----------------------------------------------------------------------------- autograd port
class _ContigGrad(torch.autograd.Function):
    """Identity whose backward hands a CONTIGUOUS cotangent upstream.

    PyTorch 2.10 CPU ``native_batch_norm_backward`` returns wrong input gradients when the incoming
    cotangent is a permuted (channels-last-strided) view while the saved input is NCHW-contiguous --
    exactly what the reference's ``ln_post(output.squeeze(-1).permute(0,2,1))`` after ``bn2`` produces.
    (Finite differences in float64 disagree with stock autograd by O(1); see oracle/make_golden.py.)
    """

    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


def _bn_safe(x, p, name, training, cfg):
    y = F.batch_norm(x.contiguous(), p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"],
                     p[name + ".bias"], training, cfg.bn_momentum, cfg.eps)
    return _ContigGrad.apply(y)


def forward_autograd(p: Dict[str, Tensor], X: Tensor, Y: Tensor, cfg: AdapterConfig, training: bool = True):
    """Same math through ATen ops + autograd, in the reference's op order (conv on the token axis,
    then fc; bmm/softmax/bmm; four Linear gates; grouped 1x1 convs; BatchNorm; LayerNorm).  Used
    (a) to cross-check ``backward`` and (b) as bench.py's cpu_baseline 'port' leg."""
    B, N, C = X.shape
    if cfg.remap == "conv":
        Yp = F.conv2d(Y.unsqueeze(-1), p["conv_adapter.weight"], p["conv_adapter.bias"]).squeeze(-1)
        Yp = F.linear(Yp, p["fc.weight"], p["fc.bias"])
    else:
        Yp = F.linear(Y, p["fc.weight"], p["fc.bias"])
        Yp = torch.einsum("mn,bnc->bmc", p["_bicubic"], Yp)
    T0 = p["my_tokens"].unsqueeze(0).expand(B, -1, -1)
    tok = T0 + torch.bmm(torch.softmax(torch.bmm(T0, Yp.transpose(1, 2)), -1), Yp)
    X1 = X + p["gate_av"] * torch.bmm(torch.softmax(torch.bmm(X, tok.transpose(1, 2)), -1), tok)
    a = Yp.mean(1)
    aq1 = F.relu(F.linear(a, p["fc_affine_audio_1.weight"], p["fc_affine_audio_1.bias"])).unsqueeze(1)
    vq1 = F.relu(F.linear(X1, p["fc_affine_video_1.weight"], p["fc_affine_video_1.bias"]))
    q = F.relu(F.linear((aq1 * vq1).mean(1), p["fc_affine_bottleneck.weight"], p["fc_affine_bottleneck.bias"]))
    ch = torch.sigmoid(F.linear(q, p["fc_affine_v_c_att.weight"], p["fc_affine_v_c_att.bias"])).unsqueeze(1)
    Xc = X1 * (ch + 1)
    vq2 = F.relu(F.linear(Xc, p["fc_affine_video_2.weight"], p["fc_affine_video_2.bias"]))
    aq2 = F.relu(F.linear(a, p["fc_affine_audio_2.weight"], p["fc_affine_audio_2.bias"])).unsqueeze(1)
    sl = F.linear(vq2 * aq2, p["fc_affine_v_s_att.weight"], p["fc_affine_v_s_att.bias"])     # [B,N,1]
    amap = torch.softmax(torch.tanh(sl).transpose(1, 2), -1).squeeze(1)
    mod = cfg.alpha * ch + cfg.beta * torch.sigmoid(sl) + 1 - cfg.alpha
    tg = None
    if cfg.temporal:
        tg = torch.sigmoid(F.linear(a, p["temporal_gated.0.weight"], p["temporal_gated.0.bias"])).squeeze(-1)
        mod = mod + cfg.gamma * tg[:, None, None]
    x = X1 * mod
    if cfg.ln_before:
        x = F.layer_norm(x, (C,), p["ln_before.weight"], p["ln_before.bias"], cfg.eps)
    x = x.transpose(1, 2).unsqueeze(-1)                                # [B,C,N,1]
    z = F.conv2d(x, p["down_sampler.weight"], None, groups=cfg.g)
    if cfg.use_bn:
        z = _bn_safe(z, p, "bn1", training, cfg)
    z = F.relu(z)
    o = F.conv2d(z, p["up_sampler.weight"], None, groups=cfg.g)
    if cfg.use_bn:
        o = _bn_safe(o, p, "bn2", training, cfg)
    o = o.squeeze(-1).transpose(1, 2)
    gate = p["gate"] if cfg.use_gate else None
    if cfg.gate_before_ln_post and gate is not None:
        o = o * gate
    if cfg.ln_post:
        o = F.layer_norm(o, (C,), p["ln_post.weight"], p["ln_post.bias"], cfg.eps)
    if not cfg.gate_before_ln_post and gate is not None:
        o = o * gate
    return o, amap, tg


def map_pool(F: Tensor, amap: Tensor) -> Tensor:
    """Spatial-map pooling after the layer loop, DG-SCT/AVE/nets/net_trans.py:922-924 (`f_v = torch.bmm(f_v_spatial_att_maps,
    f_v)`): F [BT,N,C], amap [BT,1,N] -> [BT,1,C].  Restated as the explicit weighted sum (float64 accumulation)."""
    return (amap.reshape(F.shape[0], -1, 1).double() * F.double()).sum(dim=1, keepdim=True)


def map_pool_bwd(F: Tensor, amap: Tensor, dP: Tensor):
    """Cotangents of map_pool: dF = amap^T (x) dP, damap[b,0,n] = <F[b,n,:], dP[b,0,:]>."""
    dF = amap.reshape(F.shape[0], -1, 1).double() * dP.reshape(F.shape[0], 1, -1).double()
    dmap = (F.double() * dP.reshape(F.shape[0], 1, -1).double()).sum(dim=2).unsqueeze(1)
    return dF, dmap
