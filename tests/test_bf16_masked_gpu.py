"""GPU: the bf16 BACKWARD against an oracle that is fed the device's own ReLU masks.

Why: gradients of this adapter are not 1e-2-stable under ANY bf16 evaluation -- a bottleneck / query unit whose
pre-activation lies within bf16 rounding of zero takes either side of the ReLU, and every flip moves that unit's whole
contribution -- so a comparison against the plain fp32 oracle needs 5-25 % bounds, inside which a real
kernel bug of a few per cent would hide.  Here the oracle backward uses the masks the device forward actually took
(`dgsct_saved_region` exposes vq1, vq2, Z, q, aq1, aq2; `oracle.backward(masks=...)`): flip noise is gone, what is left
is the arithmetic error of the bf16 kernels (operand rounding through the un-scaled softmax logits included), and the
bounds below are tight enough to fail on a wrong kernel.

Asserted, relative L2 per tensor (see `bounds`): dX, dY, every weight matrix, every bias / scale vector except four
cancellation residues.  Each case also prints the un-pinned errors and the fraction of flipped units for the record."""
import pytest
import torch

from helpers import device_relu_masks, param_table, spec_of
from dgsct_amd import ops
from dgsct_amd._lib import PARAM_NAMES, default_lib
from oracle import dgsct_oracle as O
from oracle import dgsct_oracle_bf16 as OB

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

# (N, C, No, Co, BT, flavour): the eight AVE shapes of BASELINE configs[1] (Swin-V2-B), the widest Swin-V2-L pair, AVS-S4's
# bicubic remap at its widest stage, and the benchmark's BT = 160 at a late and at the largest (stage-0) shape
CASES = [
    (2304, 128, 4096, 96, 4, "ave"), (4096, 96, 2304, 128, 4, "ave"), (576, 256, 1024, 192, 10, "ave"), (1024, 192, 576, 256, 10, "ave"),
    (144, 512, 256, 384, 10, "ave"), (256, 384, 144, 512, 10, "ave"), (36, 1024, 64, 768, 10, "ave"), (64, 768, 36, 1024, 10, "ave"),
    (36, 1536, 64, 768, 10, "ave"), (64, 768, 36, 1536, 10, "ave"), (36, 1536, 64, 768, 10, "avs_s4"),
    (144, 512, 256, 384, 160, "ave"), (2304, 128, 4096, 96, 160, "ave"),
]
def bounds(C, flavour, BT=10):
    """relative-L2 bounds with the device's ReLU masks pinned.  What is left after pinning is operand rounding: dX sees it
    once (0.5-1.9 % measured); dY, my_tokens and the remap weights sit behind the two UN-SCALED softmaxes, whose logits are
    sums over C channels of bf16-rounded Yp -- a 2^-9 relative rounding becomes a logit error ~ sqrt(C) 2^-9 and the measured
    error grows linearly with C: 1.0-1.5 % at C <= 256, 2.3-2.8 % at 384-512, 4.4-5.9 % at 768-1024, 8.8 % at 1536 (x 2.2 for the
    AVS-S4 flavour, whose bicubic operator mixes 9-16 source tokens per target).  Bounds = 1.5 x those lines: a kernel that is
    wrong by more than the rounding it is entitled to fails.  Weight matrices at the benchmark's 160 frames: fc.weight measured 2.4 %
    (1.0 % at 4 frames) at the stage-0 shape -- the bf16 rounding of the SHARED weights is a correlated error that the sum over
    frames does not average out -- hence the 1.7 x there."""
    f = 2.2 if flavour == "avs_s4" else 1.0
    dY = 1.5 * f * max(1e-2, 6e-5 * C)
    return dict(dX=2e-2 * (1.8 if C > 1024 else 1.0) * (1.3 if flavour == "avs_s4" else 1.0), dY=dY,
                W=dY * (1.7 if BT >= 100 else 1.0), V=3.0 * dY)


# Cancellation residues a bf16 run cannot resolve (their fp32-path parity is asserted at 1e-3 elsewhere): ln_before.bias is
# analytically zero; the two gates and the spatial bias are eps-sized sums of large terms; fc.bias sums dYp
# over every row of every frame, and softmax rows make those sums cancel to 5-20 % of their bf16 noise floor.
RESIDUES = ("ln_before.bias", "gate", "gate_av", "fc_affine_v_s_att.bias", "fc.bias")


def _l2(a, b):
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_bf16_backward_with_device_relu_masks(case):
    N, C, No, Co, BT, flavour = case
    cfg = O.AdapterConfig(**{**dict(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2), **O.FLAVOURS[flavour]})
    p = O.random_params(cfg, flavour, seed=0, scale=0.577)
    if cfg.remap == "bicubic":
        p["_bicubic"] = O.bicubic_matrix(No, N)
    gen = torch.Generator().manual_seed(1)
    rb = lambda t: t.bfloat16().float()
    X, Y = rb(torch.randn(BT, N, C, generator=gen)), rb(torch.randn(BT, No, Co, generator=gen))
    dOut, dMap = rb(torch.randn(BT, N, C, generator=gen)), torch.randn(BT, N, generator=gen)
    spec = spec_of(cfg)
    lib = default_lib()
    params = param_table(p, spec, DEV)
    dt = torch.bfloat16
    Xd, Yd = X.to(DEV, dt).contiguous(), Y.to(DEV, dt).contiguous()
    prep = ops.prepare(lib, spec, params, dt, DEV)
    # (the fused gate passes of stages 0-1 recompute vq2 in backward instead of storing it: "gatefuse" = 2 makes the forward
    #  materialise it as well, so that the device's ReLU decisions can be read back)
    # (likewise vq1 at C = 96 / 128: "vq1fuse" = 2)
    old = lib.test_tune("gatefuse", 2)
    old1 = lib.test_tune("vq1fuse", 2)
    try:
        out, amap, _, saved, d = ops.raw_forward(lib, spec, params, prep, Xd, Yd, True)
        torch.cuda.synchronize()
    finally:
        lib.test_tune("gatefuse", old)
        lib.test_tune("vq1fuse", old1)
    masks = device_relu_masks(lib, d, saved, spec, BT, dt)
    dX, dY, grads = ops.raw_backward(lib, spec, d, params, prep, Xd, Yd, saved, dOut.to(DEV, dt).contiguous(), dMap.to(DEV), None)
    torch.cuda.synchronize()

    po = {k: v.clone() for k, v in p.items()}
    out_o, map_o, _, s = O.forward(po, X, Y, cfg, training=True)
    flips = {k: float((masks[k] != (s[k] > 0)).float().mean()) for k in masks}
    dX_o, dY_o, g_o = O.backward(po, s, cfg, dOut, dMap, None, training=True, masks=masks)
    # without the pinning the same comparison (what round 2 asserted against 8-63 % bounds)
    dX_u, dY_u, _ = O.backward(po, s, cfg, dOut, dMap, None, training=True)

    bd = bounds(C, flavour, BT)
    # forward values of the same call: BASELINE's bf16 bound (1e-2 relative L2) on out and map (they were
    # computed and dropped)
    fwd = {"out": _l2(out, out_o), "map": _l2(amap, map_o)}
    assert fwd["out"] < 1e-2 and fwd["map"] < 1e-2, fwd
    rep = {"out": fwd["out"], "map": fwd["map"], "dX": _l2(dX, dX_o), "dY": _l2(dY, dY_o), "dX_unpinned": _l2(dX, dX_u), "dY_unpinned": _l2(dY, dY_u)}
    bad = [(k, rep[k], bd[k]) for k in ("dX", "dY") if rep[k] > bd[k]]
    errs = {}
    for i, g in enumerate(grads):
        name = PARAM_NAMES[i]
        if g is None or name not in g_o:
            continue
        if name in RESIDUES:
            continue
        mat = g_o[name].dim() >= 2 and min(g_o[name].shape[:2]) > 1
        e = errs[name] = _l2(g, g_o[name])
        lim = bd["W"] if mat else bd["V"]
        if e > lim:
            bad.append((name, e, lim))
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    rep.update(worst={k: round(v, 4) for k, v in top}, flips={k: round(v, 5) for k, v in flips.items()},
               bounds={k: round(v, 4) for k, v in bd.items()})
    print("MASKED", case, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rep.items()})
    assert not bad, (bad, rep)


# ---- the same backward against the ROUNDING-AWARE oracle (round 5) ---------------------------------------------------------------------
# The bounds above grow with C because the fp32 oracle does not round what a bf16 schedule must round (Wn, Wc, T, Yp in front of two
# un-scaled softmaxes).  That rounding is deterministic, so an evaluation of the oracle that rounds the same tensors
# (oracle/dgsct_oracle_bf16.py, pinned to the oracle by tests/test_host_cpu.py) lands on the device's values: what is left is accumulation
# order and the few tensors the two keep at different precision.  ONE bound for every width -- a wrong kernel has nowhere to hide at
# C = 768-1536 either.
AWARE_CASES = list(CASES)                   # (the bicubic AVS-S4 case too: 19.5 % against the fp32 oracle in round 3)
# ... and the flavours the fp32-oracle test above never ran in bf16: the temporal gate (pretrain, with a temporal-map cotangent), AVS-MS3's
# constants, AVQA (no BatchNorm, 2 latent tokens, 4 groups)
AWARE_CASES += [(144, 512, 256, 384, 10, "pretrain"), (576, 256, 1024, 192, 10, "pretrain"), (36, 1024, 64, 768, 10, "avs_ms3"),
                (144, 512, 256, 384, 10, "avqa"), (1024, 192, 576, 256, 10, "avqa")]
# measured (13 cases, tools/bf16_aware_search.py for how the rounding points were chosen): dX 0.38-0.53 %, dY 0.74-1.04 %, weight matrices
# <= 1.08 %, bias / scale vectors <= 1.3 % (2.5 % for bn1.bias over 160 frames x 4096 tokens) -- against 1.3-8.8 % on the fp32 oracle
AWARE_BOUND = dict(dX=8e-3, dY=1.3e-2, W=1.3e-2, V=3.5e-2)


def _aware_params():
    out = [(c, "default") for c in AWARE_CASES]
    # the fused row kernels (fused_gate.hip: gatemod_* at C <= 256, vq1_* at C = 96 / 128) and the fused GEMM hooks (gemm_fx.hip) exist in
    # bf16 only, so no fp32-vs-oracle test reaches them (VERDICT r4 weak #2a): here the SAME yardstick is applied to the schedule with every
    # one of them switched off -- both must sit within the same bound, i.e. a fused kernel may not cost more than the launches it replaces
    out += [(c, "unfused") for c in AWARE_CASES if c[1] <= 512 and c[4] <= 10]
    return out


@pytest.mark.parametrize("case,fusion", _aware_params(), ids=lambda v: v if isinstance(v, str) else "x".join(str(x) for x in v))
def test_bf16_backward_against_the_rounding_aware_oracle(case, fusion):
    N, C, No, Co, BT, flavour = case
    cfg = O.AdapterConfig(**{**dict(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2), **O.FLAVOURS[flavour]})
    p = O.random_params(cfg, flavour, seed=0, scale=0.577)
    if cfg.remap == "bicubic":
        p["_bicubic"] = O.bicubic_matrix(No, N)
    gen = torch.Generator().manual_seed(1)
    rb = lambda t: t.bfloat16().float()
    X, Y = rb(torch.randn(BT, N, C, generator=gen)), rb(torch.randn(BT, No, Co, generator=gen))
    dOut, dMap = rb(torch.randn(BT, N, C, generator=gen)), torch.randn(BT, N, generator=gen)
    spec = spec_of(cfg)
    lib = default_lib()
    params = param_table(p, spec, DEV)
    dt = torch.bfloat16
    Xd, Yd = X.to(DEV, dt).contiguous(), Y.to(DEV, dt).contiguous()
    prep = ops.prepare(lib, spec, params, dt, DEV)
    off = fusion == "unfused"
    old = lib.test_tune("gatefuse", 0 if off else 2)
    old1 = lib.test_tune("vq1fuse", 0 if off else 2)
    old2 = lib.test_tune("gemmfx", 0) if off else None
    try:
        out, amap, _, saved, d = ops.raw_forward(lib, spec, params, prep, Xd, Yd, True)
        torch.cuda.synchronize()
        if not off:
            lib.test_tune("gatefuse", old)
            lib.test_tune("vq1fuse", old1)
        masks = device_relu_masks(lib, d, saved, spec, BT, dt)
        dTm = torch.randn(BT, generator=gen) if cfg.temporal else None
        dX, dY, grads = ops.raw_backward(lib, spec, d, params, prep, Xd, Yd, saved, dOut.to(DEV, dt).contiguous(), dMap.to(DEV),
                                         dTm.to(DEV) if dTm is not None else None)
        torch.cuda.synchronize()
    finally:
        lib.test_tune("gatefuse", old)
        lib.test_tune("vq1fuse", old1)
        if old2 is not None:
            lib.test_tune("gemmfx", old2)
    r = OB.evaluate(cfg, p, X, Y, dOut, dMap, OB.Q(OB.DEVICE_ROUNDING), masks=masks, dTmap=dTm)
    rep = {"out": _l2(out, r["out"]), "map": _l2(amap, r["map"]), "dX": _l2(dX, r["dX"]), "dY": _l2(dY, r["dY"])}
    assert rep["out"] < 1e-2 and rep["map"] < 1e-2, rep
    # the AVS-S4 case (bicubic operator, C = 1536; 19.5 % against the fp32 oracle): dY 1.45 %, fc.weight 1.43 %, my_tokens 1.32 % -- the
    # fixed operator's 9-16 taps per target token are bf16 here as on the device, but summed in another order: 1.35 x the bound
    fl = {"avs_s4": 1.35, "avqa": 1.15}.get(flavour, 1.0)      # (avqa, 2 latent tokens: fc_affine_audio_2.weight 1.17 % at 1024 x 192)
    bound = {k: v * (fl if k in ("dY", "W") else 1.0) for k, v in AWARE_BOUND.items()}
    bad = [(k, rep[k], bound[k]) for k in ("dX", "dY") if rep[k] > bound[k]]
    errs = {}
    for i, g in enumerate(grads):
        name = PARAM_NAMES[i]
        if g is None or name not in r["g"] or name in RESIDUES:
            continue
        ref = r["g"][name]
        mat = ref.dim() >= 2 and min(ref.shape[:2]) > 1
        e = errs[name] = _l2(g, ref)
        lim = bound["W"] if mat else bound["V"]
        if e > lim:
            bad.append((name, e, lim))
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print("AWARE", case, fusion, {k: round(v, 4) for k, v in rep.items()}, {k: round(v, 4) for k, v in top})
    assert not bad, (bad, rep)
