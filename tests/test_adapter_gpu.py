"""GPU parity: libdgsct.so (hand-written gfx950 kernels, through the C ABI) against the reference golden
vectors and against the oracle at real AVE shapes.

fp32: BASELINE.json's 1e-3, measured as max(relative L2, worst element / max|ref|) per tensor -- no absolute floor (helpers.fp32_err;
round 2's max|err| / max(1, max|ref|) let an all-zero attention map pass at N >= 1000) --, every output, every gradient, BN buffers.
bf16: OUTPUTS (out, map) within BASELINE.json's 1e-2 in relative L2 and 2e-2 in the worst element (max|err| / max|ref|) at
real shapes; every tensor (outputs AND all gradients) within the emulator-derived per-case bound of
tests/golden/bf16_bounds.json (oracle/make_bf16_bounds.py: the range an IDEAL bf16-storage evaluation of the same
schedule covers when its inputs move by one bf16 ulp -- gradients of this model are not 1e-2-stable under ANY bf16
rounding: ReLU-mask flips and un-scaled softmax logits, DESIGN.md section 7 / tools/bf16_sensitivity.py).  The oracle
always sees the same bf16-representable inputs as the library."""
import json
import os
import pytest
import torch

from helpers import fp32_err, golden_names, grad_close_fp32, load_golden, nrm_err, oracle_cfg, param_table, rel_err, run_library, spec_of
from dgsct_amd import ops
from dgsct_amd._lib import P_INDEX, PARAM_NAMES, default_lib
from oracle import dgsct_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL_F32 = 1e-3
TOL_BF16 = 1e-2
BOUNDS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_bounds.json")))


def _l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def check_bounds(name, got, ref_out, ref_map, ref_dX, ref_dY, ref_grads):
    """every tensor of a bf16 run within the emulator-derived relative-L2 bound of its case"""
    b = BOUNDS[name]
    bad = []
    for k, g, r in (("out", got["out"], ref_out), ("map", got["map"], ref_map), ("dX", got["dX"], ref_dX), ("dY", got["dY"], ref_dY)):
        assert torch.isfinite(g.float()).all(), k
        if not _l2(g, r) <= b[k]:
            bad.append((k, round(_l2(g, r), 4), round(b[k], 4)))
    for k, bound in b["grads"].items():
        if bound >= 0.5:
            continue        # a bound this loose asserts nothing: these tensors are held by tests/test_bf16_masked_gpu.py instead
        e = _l2(got["grads"][k].reshape(-1), ref_grads[k].reshape(-1))
        if not e <= bound:
            bad.append((k, round(e, 4), round(bound, 4)))
    assert not bad, bad


@pytest.mark.parametrize("name", golden_names())
def test_golden_fp32(name):
    fx = load_golden(name)
    r = run_library(default_lib(), fx, DEV, torch.float32, training=True)
    torch.cuda.synchronize()
    bad = [(k, fp32_err(r[k], fx[k])) for k in ("out", "map", "dX", "dY") if not fp32_err(r[k], fx[k]) < TOL_F32]
    if fx["tmap"] is not None and not fp32_err(r["tmap"], fx["tmap"]) < TOL_F32:
        bad.append(("tmap", fp32_err(r["tmap"], fx["tmap"])))
    for k, g in fx["grads"].items():
        assert k in r["grads"], k
        if not grad_close_fp32(r["grads"][k], g, TOL_F32, name=k):
            bad.append((k, fp32_err(r["grads"][k], g)))
    assert not (set(r["grads"]) - set(fx["grads"]))
    for k, v in fx["buffers1"].items():
        if "running" in k and not fp32_err(r["params"][P_INDEX[k]], v) < TOL_F32:
            bad.append((k, fp32_err(r["params"][P_INDEX[k]], v)))
    assert not bad, bad


@pytest.mark.parametrize("name", ["ave_orderA", "avs_s4", "avqa", "pretrain"])
def test_golden_eval_fp32(name):
    fx = dict(load_golden(name))
    st = dict(fx["state0"]); st.update(fx["buffers1"]); fx["state0"] = st
    r = run_library(default_lib(), fx, DEV, torch.float32, training=False)
    assert fp32_err(r["out"], fx["eval_out"]) < TOL_F32
    assert fp32_err(r["map"], fx["eval_map"]) < TOL_F32


@pytest.mark.parametrize("name", golden_names())
def test_golden_bf16(name):
    """all 12 flavour cases in bf16, outputs AND every gradient, against the oracle on the same bf16-rounded inputs (the
    fixtures' own inputs are generic fp32, so their stored results are not the reference of a bf16 run), each tensor within
    its emulator-derived bound.  The cases are deliberately tiny (4-16 bottleneck channels): one flipped ReLU unit moves a
    gradient by 1-25 %, which is what the bounds of these cases reflect."""
    fx = load_golden(name)
    r = run_library(default_lib(), fx, DEV, torch.bfloat16, training=True)
    cfg = oracle_cfg(fx["cfg"])
    state = {k: v.clone() for k, v in fx["state0"].items()}
    if cfg.remap == "bicubic":
        state["_bicubic"] = O.bicubic_matrix(cfg.No, cfg.N)
    rb = lambda t: t.bfloat16().float()
    out_o, map_o, _, s = O.forward(state, rb(fx["X"]), rb(fx["Y"]), cfg, training=True)
    dX_o, dY_o, g_o = O.backward(state, s, cfg, rb(fx["dOut"]), fx["dMap"], fx["dTmap"], training=True)
    check_bounds(name, r, out_o, map_o, dX_o, dY_o, g_o)


def _real_case(N, C, No, Co, BT, dtype, seed=0, flavour="ave", tk=None):
    cfg = O.AdapterConfig(**{**dict(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2), **O.FLAVOURS[flavour], **(dict(tk=tk) if tk else {})})
    # weights at the reference's default-init scale: nn.Linear/Conv2d init is U(-1/sqrt(fan_in), 1/sqrt(fan_in)),
    # i.e. std = 0.577/sqrt(fan_in) (random_params draws N(0, scale^2/fan_in)); my_tokens ~ U[0,1) as in the reference
    p = O.random_params(cfg, flavour, seed=seed, scale=0.577)
    gen = torch.Generator().manual_seed(seed + 1)
    X = torch.randn(BT, N, C, generator=gen)
    Y = torch.randn(BT, No, Co, generator=gen)
    dOut = torch.randn(BT, N, C, generator=gen)
    dMap = torch.randn(BT, N, generator=gen)
    if dtype == torch.bfloat16:   # the oracle sees the same rounded inputs
        X, Y, dOut = X.bfloat16().float(), Y.bfloat16().float(), dOut.bfloat16().float()
    po = {k: v.clone() for k, v in p.items()}
    out_o, map_o, _, s = O.forward(po, X, Y, cfg, training=True)
    spec = spec_of(cfg)
    params = param_table(p, spec, DEV)
    lib = default_lib()
    Xd, Yd = X.to(DEV, dtype).contiguous(), Y.to(DEV, dtype).contiguous()
    prep = ops.prepare(lib, spec, params, dtype, DEV)
    out, amap, _, saved, d = ops.raw_forward(lib, spec, params, prep, Xd, Yd, True)
    # fp32: the oracle differentiates the ReLU branches the device took (a pre-activation within fp32 rounding of zero lands on
    # either side with a different summation order; pinned, the relative metric needs no per-flip exceptions).  bf16 keeps the
    # un-pinned oracle here -- its bounds (bf16_bounds.json) were derived that way; the pinned bf16 test is test_bf16_masked_gpu.py
    masks = None
    if dtype == torch.float32:
        from helpers import device_relu_masks
        torch.cuda.synchronize()
        masks = device_relu_masks(lib, d, saved, spec, BT, dtype)
    dX_o, dY_o, g_o = O.backward(po, s, cfg, dOut, dMap, None, training=True, masks=masks)
    dX, dY, grads = ops.raw_backward(lib, spec, d, params, prep, Xd, Yd, saved, dOut.to(DEV, dtype).contiguous(),
                                     dMap.to(DEV), None)
    return dict(out=(out, out_o), map=(amap, map_o), dX=(dX, dX_o), dY=(dY, dY_o),
                grads={PARAM_NAMES[i]: (g, g_o[PARAM_NAMES[i]]) for i, g in enumerate(grads)
                       if g is not None and PARAM_NAMES[i] in g_o},
                extra=[PARAM_NAMES[i] for i, g in enumerate(grads) if g is not None and PARAM_NAMES[i] not in g_o])


# (N, C, No, Co) of AVE stage 2 / stage 3, visual and audio direction, Swin-V2-B widths (BASELINE config 2)
REAL = [(144, 512, 256, 384), (256, 384, 144, 512), (36, 1024, 64, 768), (64, 768, 36, 1024)]


@pytest.mark.parametrize("shape", REAL)
def test_real_shapes_fp32(shape):
    r = _real_case(*shape, BT=10, dtype=torch.float32)
    for k in ("out", "map", "dX", "dY"):
        assert fp32_err(*r[k]) < TOL_F32, (k, fp32_err(*r[k]))
    bad = [(k, fp32_err(g, go)) for k, (g, go) in r["grads"].items() if not grad_close_fp32(g, go, TOL_F32, name=k)]
    assert not bad, bad


@pytest.mark.parametrize("shape", REAL)
def test_real_shapes_bf16(shape):
    """bf16 storage + bf16 MFMA at real AVE shapes against the fp32 oracle on the same (bf16-representable) inputs.
    Outputs: BASELINE.json's 1e-2 (relative L2) and 2e-2 of max|ref| in the worst element.  Every tensor, gradients
    included: the emulator-derived bound of the shape (tests/golden/bf16_bounds.json)."""
    r = _real_case(*shape, BT=10, dtype=torch.bfloat16, seed=0)
    for k in ("out", "map"):
        assert _l2(*r[k]) < TOL_BF16, (k, _l2(*r[k]))
        assert nrm_err(*r[k]) < 2 * TOL_BF16, (k, nrm_err(*r[k]))
    got = dict(out=r["out"][0], map=r["map"][0], dX=r["dX"][0], dY=r["dY"][0], grads={k: v[0] for k, v in r["grads"].items()})
    check_bounds(f"real_{shape[0]}x{shape[1]}", got, r["out"][1], r["map"][1], r["dX"][1], r["dY"][1],
                 {k: v[1] for k, v in r["grads"].items()})


# ---- num_tokens > 32: more than one 32-row MFMA tile of latent tokens per frame (csrc/attn_wide.cpp: batched products on the tiled
# engine + row softmax instead of the fused attention kernels).  87 is the reference constructor's default (net_trans.py:437).
WIDE = [(144, 512, 256, 384, 87), (64, 768, 36, 1024, 40)]       # (N, C, No, Co, tk): bounds exist for these (oracle/make_bf16_bounds.py)


@pytest.mark.parametrize("shape", WIDE + [(576, 256, 1024, 192, 87), (1024, 192, 576, 256, 33)])
def test_wide_token_path_real_shapes_fp32(shape):
    N, C, No, Co, tk = shape
    r = _real_case(N, C, No, Co, BT=10, dtype=torch.float32, tk=tk)
    for k in ("out", "map", "dX", "dY"):
        assert fp32_err(*r[k]) < TOL_F32, (k, fp32_err(*r[k]))
    assert not r["extra"], r["extra"]
    bad = [(k, fp32_err(g, go)) for k, (g, go) in r["grads"].items() if not grad_close_fp32(g, go, TOL_F32, name=k)]
    assert not bad, bad


@pytest.mark.parametrize("shape", WIDE)
def test_wide_token_path_real_shapes_bf16(shape):
    """as test_real_shapes_bf16: outputs at BASELINE.json's 1e-2, every tensor inside the emulator-derived bound of the case"""
    N, C, No, Co, tk = shape
    r = _real_case(N, C, No, Co, BT=10, dtype=torch.bfloat16, seed=0, tk=tk)
    for k in ("out", "map"):
        assert _l2(*r[k]) < TOL_BF16, (k, _l2(*r[k]))
        assert nrm_err(*r[k]) < 2 * TOL_BF16, (k, nrm_err(*r[k]))
    got = dict(out=r["out"][0], map=r["map"][0], dX=r["dX"][0], dY=r["dY"][0], grads={k: v[0] for k, v in r["grads"].items()})
    check_bounds(f"wide_tk{tk}_{N}x{C}", got, r["out"][1], r["map"][1], r["dX"][1], r["dY"][1], {k: v[1] for k, v in r["grads"].items()})


@pytest.mark.parametrize("name", golden_names())
def test_golden_fp32_on_the_wide_token_path(name, monkeypatch):
    """DGSCT_WIDE_ATTN=1 (read by the library at every layout) sends num_tokens <= 32 down the wide path too: the reference goldens
    of every flavour then check it, tk = 2 and tk = 4 included"""
    monkeypatch.setenv("DGSCT_WIDE_ATTN", "1")
    fx = load_golden(name)
    r = run_library(default_lib(), fx, DEV, torch.float32, training=True)
    torch.cuda.synchronize()
    bad = [(k, fp32_err(r[k], fx[k])) for k in ("out", "map", "dX", "dY") if not fp32_err(r[k], fx[k]) < TOL_F32]
    for k, g in fx["grads"].items():
        if not grad_close_fp32(r["grads"][k], g, TOL_F32, name=k):
            bad.append((k, fp32_err(r["grads"][k], g)))
    assert not (set(r["grads"]) - set(fx["grads"]))
    assert not bad, bad


def test_wide_token_path_matches_the_fused_kernels_bf16(monkeypatch):
    """the same bf16 call on both paths (tk = 32, a real stage-2 shape): they differ by rounding order only -- outputs to 1e-2 of each
    other, input gradients to the distance either keeps from the oracle"""
    a = _real_case(144, 512, 256, 384, BT=10, dtype=torch.bfloat16, seed=3)
    monkeypatch.setenv("DGSCT_WIDE_ATTN", "1")
    b = _real_case(144, 512, 256, 384, BT=10, dtype=torch.bfloat16, seed=3)
    for k in ("out", "map"):
        assert _l2(a[k][0], b[k][0]) < TOL_BF16, (k, _l2(a[k][0], b[k][0]))
    for k in ("dX", "dY"):
        ea, eb = _l2(*a[k]), _l2(*b[k])
        assert eb < 1.5 * ea + 1e-2, (k, ea, eb)


@pytest.mark.parametrize("flavour", ["avvp", "avs_s4", "avs_ms3", "avqa", "pretrain"])
def test_real_shapes_flavours_fp32(flavour):
    """every call-site flavour of SURVEY 8a row a-0 (bicubic remap, gate-before-LN, tk=2/g=4/no BN, temporal gate) at
    a real stage-2 shape (12x12 visual tokens <- 16x16 audio tokens) against the oracle"""
    r = _real_case(144, 512, 256, 384, BT=10, dtype=torch.float32, flavour=flavour)
    for k in ("out", "map", "dX", "dY"):
        assert fp32_err(*r[k]) < TOL_F32, (k, fp32_err(*r[k]))
    assert not r["extra"], r["extra"]
    assert r["grads"]
    bad = [(k, fp32_err(g, go)) for k, (g, go) in r["grads"].items() if not grad_close_fp32(g, go, TOL_F32, name=k)]
    assert not bad, bad


def _full_size_setup(shape, seed, gate=None):
    N, C, No, Co = shape
    cfg = O.AdapterConfig(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2)
    p = O.random_params(cfg, "ave", seed=seed, scale=0.577)
    if gate is not None:
        p["gate"] = torch.full_like(p["gate"], gate)
    spec = spec_of(cfg)
    lib = default_lib()
    params = param_table(p, spec, DEV)
    gen = torch.Generator().manual_seed(seed + 1)
    X = torch.randn(160, N, C, generator=gen).to(DEV, torch.bfloat16)
    Y = torch.randn(160, No, Co, generator=gen).to(DEV, torch.bfloat16)
    prep = ops.prepare(lib, spec, params, torch.bfloat16, DEV)
    return lib, spec, params, prep, X, Y, gen


@pytest.mark.parametrize("shape", [(2304, 128, 4096, 96), (4096, 96, 2304, 128)])
def test_full_size_zero_gate_and_map_normalisation(shape):
    """BASELINE size (160 frames, stage 0, bf16).  The reference initialises `gate` to 0 (net_trans.py:447): the adapter
    output is then exactly 0 -- and exactly the residual with the fused skip -- whatever the other 9.6 M parameters are;
    the returned map is a softmax over the N tokens of every frame (SURVEY 8c "analytic edge cases")."""
    lib, spec, params, prep, X, Y, _ = _full_size_setup(shape, seed=11, gate=0.0)
    out, amap, _, _, _ = ops.raw_forward(lib, spec, params, prep, X, Y, True)
    assert float(out.float().abs().max()) == 0.0
    out2, amap2, _, _, _ = ops.raw_forward(lib, spec, params, prep, X, Y, True, X)
    assert torch.equal(out2, X)
    assert torch.isfinite(amap).all() and float((amap.sum(-1) - 1).abs().max()) < 1e-4
    assert float(amap.min()) >= 0.0


def test_full_size_backward_is_linear_in_the_cotangents():
    """BASELINE size (160 frames, stage-0 audio adapter, bf16): backward is a linear map of (dOut, dMap), so
    bwd(g1 + g2) = bwd(g1) + bwd(g2) for dX, dY and every parameter gradient (up to bf16 rounding of the cotangents'
    intermediates) -- a size-independent check of the whole backward schedule, split-K atomics included."""
    shape = (4096, 96, 2304, 128)
    lib, spec, params, prep, X, Y, gen = _full_size_setup(shape, seed=21)
    N, C = shape[0], shape[1]
    g1 = torch.randn(160, N, C, generator=gen).to(DEV, torch.bfloat16)
    g2 = torch.randn(160, N, C, generator=gen).to(DEV, torch.bfloat16)
    m1 = torch.randn(160, N, generator=gen).to(DEV)
    m2 = torch.randn(160, N, generator=gen).to(DEV)

    def bwd(g, m):
        _, _, _, saved, d = ops.raw_forward(lib, spec, params, prep, X, Y, True)      # backward consumes `saved`
        dX, dY, grads = ops.raw_backward(lib, spec, d, params, prep, X, Y, saved, g.contiguous(), m.contiguous(), None)
        torch.cuda.synchronize()
        return dX.float(), dY.float(), [x.clone() if x is not None else None for x in grads]

    a = bwd(g1, m1)
    b = bwd(g2, m2)
    c = bwd((g1.float() + g2.float()).to(torch.bfloat16), m1 + m2)
    assert _l2(a[0] + b[0], c[0]) < 2e-2 and _l2(a[1] + b[1], c[1]) < 2e-2
    bad = []
    for i, (x, y, z) in enumerate(zip(a[2], b[2], c[2])):
        if z is None or z.numel() < 64 or float(z.norm()) == 0:
            continue
        # weight matrices: 3e-2.  Vectors (biases, LN/BN affine): sums over 655 360 rows of bf16-rounded terms with heavy
        # cancellation -- each run rounds differently, so linearity only holds to the accuracy of one run (cf. 0.6 above)
        if PARAM_NAMES[i] == "ln_before.bias":      # analytically zero (BatchNorm removes it): pure rounding noise
            continue
        tol = 3e-2 if z.numel() > 4096 else 0.25
        e = _l2(x + y, z)
        if not e < tol:
            bad.append((PARAM_NAMES[i], z.numel(), round(e, 4)))
    assert not bad, bad


@pytest.mark.parametrize("shape", [(2304, 128, 4096, 96), (4096, 96, 2304, 128), (144, 512, 256, 384)])
def test_full_size_bf16_path_matches_fp32_path(shape):
    """BASELINE size (160 frames): the bf16 production path (FAST staging, big tiles, split-K atomics, vector-unit
    projections) against the library's own fp32 parity path (generic staging, exact-fp32 MFMA), which the golden and
    real-shape tests pin to the oracle at sizes the CPU oracle can run.  Catches size-dependent addressing bugs (one was
    a 32-bit overflow in the frame split of contractions deeper than 65 536)."""
    N, C, No, Co = shape
    cfg = O.AdapterConfig(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2)
    p = O.random_params(cfg, "ave", seed=31, scale=0.577)
    spec = spec_of(cfg)
    lib = default_lib()
    gen = torch.Generator().manual_seed(32)
    X = torch.randn(160, N, C, generator=gen).bfloat16().float()
    Y = torch.randn(160, No, Co, generator=gen).bfloat16().float()
    g = torch.randn(160, N, C, generator=gen).bfloat16().float()
    m = torch.randn(160, N, generator=gen)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        params = param_table({k: v.clone() for k, v in p.items()}, spec, DEV)
        prep = ops.prepare(lib, spec, params, dt, DEV)
        Xd, Yd = X.to(DEV, dt), Y.to(DEV, dt)
        out, amap, _, saved, d = ops.raw_forward(lib, spec, params, prep, Xd, Yd, True)
        dX, dY, grads = ops.raw_backward(lib, spec, d, params, prep, Xd, Yd, saved, g.to(DEV, dt), m.to(DEV), None)
        torch.cuda.synchronize()
        res[dt] = (out.float().cpu(), amap.cpu(), dX.float().cpu(), dY.float().cpu(),
                   [x.cpu() if x is not None else None for x in grads])
        del out, amap, saved, dX, dY, grads, prep
        ops.release_workspaces()
        torch.cuda.empty_cache()
    f, h = res[torch.float32], res[torch.bfloat16]
    assert all(torch.isfinite(t).all() for t in h[:4])
    assert _l2(h[0], f[0]) < TOL_BF16 and _l2(h[1], f[1]) < TOL_BF16
    # The two paths take their OWN ReLU branches here (the fp32 path re-derives the bottleneck mask from Zp, it cannot be fed
    # the bf16 run's), so this is the un-pinned comparison: measured 4.0-4.4 % (dX) / 4.1-5.4 % (dY) at these shapes.  The tight
    # bounds -- device masks pinned, against the oracle, the 160-frame case included -- are tests/test_bf16_masked_gpu.py's.
    assert _l2(h[2], f[2]) < 0.08 and _l2(h[3], f[3]) < 0.08, (_l2(h[2], f[2]), _l2(h[3], f[3]))
    bad = []
    for i, (a, b) in enumerate(zip(h[4], f[4])):
        if b is None or b.dim() == 0 or b.numel() <= 4096:
            continue                                   # vectors / scalars: tests/test_bf16_masked_gpu.py
        assert torch.isfinite(a).all(), PARAM_NAMES[i]
        e = _l2(a, b)
        if not e < 0.12:
            bad.append((PARAM_NAMES[i], round(e, 4)))
    assert not bad, bad


@pytest.mark.parametrize("flat", [False, True])
@pytest.mark.parametrize("flavour", ["avs_s4", "avqa"])
def test_module_under_data_parallel(flavour, flat):
    """The reference's AVS and AVQA scripts wrap the model in single-process nn.DataParallel (avs_s4/train.py:139,
    net_grd_avst/main_avst.py:236): every forward re-creates the module as a REPLICA whose parameters are plain broadcast
    tensors.  (a) nn.DataParallel(device_ids=[0]); (b) a real torch.nn.parallel.replicate() replica on cuda:0, run through
    parallel_apply, with the parameter table already cached on the original; (c) two GPUs when the box has them.  All must
    reproduce the bare module's outputs and route identical gradients to its parameters."""
    from types import SimpleNamespace
    from dgsct_amd import VisualAdapter
    avs = flavour == "avs_s4"
    opt = SimpleNamespace(is_multimodal=1, num_conv_group=4 if not avs else 2, is_before_layernorm=1, is_post_layernorm=1,
                          num_tokens=2 if not avs else 8)
    N, No = (16, 36) if avs else (25, 49)

    class Wrap(torch.nn.Module):                      # DataParallel scatters tensors along dim 0 = frames
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            with __import__("warnings").catch_warnings():
                __import__("warnings").simplefilter("ignore")
                self.adapter_blocks = VisualAdapter(64, 64, "bottleneck", reduction_factor=8, opt=opt, use_bn=avs, use_gate=True,
                                                    conv_dim_in=No, conv_dim_out=N, linear_in=48, linear_out=64, flavour=flavour)
            with torch.no_grad():
                self.adapter_blocks.gate.fill_(0.7); self.adapter_blocks.gate_av.fill_(0.3)
                self.adapter_blocks.my_tokens.uniform_(0, 1)

        def forward(self, f, fo):
            out, amap = self.adapter_blocks(f.permute(0, 2, 1).unsqueeze(-1), fo.permute(0, 2, 1).unsqueeze(-1))
            return out.squeeze(-1).permute(0, 2, 1), amap

    T = 5 if avs else 10
    BT = 2 * T
    gen = torch.Generator().manual_seed(5)
    f0, fo0 = torch.randn(BT, N, 64, generator=gen), torch.randn(BT, No, 48, generator=gen)
    g_out, g_map = torch.randn(BT, N, 64, generator=gen).to(DEV), torch.randn(BT, 1, N, generator=gen).to(DEV)

    def run(call, model):
        model.zero_grad(set_to_none=True)
        f, fo = f0.to(DEV).requires_grad_(True), fo0.to(DEV).requires_grad_(True)
        out, amap = call(f, fo)
        torch.autograd.backward([out, amap], [g_out, g_map])
        return out.detach(), amap.detach(), f.grad, fo.grad, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    def same(a, b, what):
        for i, (x, y) in enumerate(zip(a[:4], b[:4])):
            assert fp32_err(x, y) < 1e-3, (what, i)        # (two runs of the same kernels: fp32 atomics reorder sums; no absolute floor)
        assert set(b[4]) <= set(a[4]) and b[4], what
        for k in b[4]:
            assert grad_close_fp32(a[4][k], b[4][k], tol=1e-3, name=k.split(".", 1)[-1] if "." in k else k), (what, k)
        for k in set(a[4]) - set(b[4]):                    # Broadcast.backward hands unused parameters a ZERO gradient (stock
            assert float(a[4][k].abs().max()) == 0.0, (what, k)   # nn.DataParallel behaviour), the bare module leaves them None

    w = Wrap().to(DEV).train()
    if flat:
        w.adapter_blocks.flatten_parameters()
    sd = {k: v.clone() for k, v in w.state_dict().items()}
    ref = run(w, w)
    w.load_state_dict(sd)                              # undo BN running-stat updates between the legs
    same(run(torch.nn.DataParallel(w, device_ids=[0]), w), ref, "DataParallel(device_ids=[0])")
    w.load_state_dict(sd)

    def via_replica(f, fo):
        rep = torch.nn.parallel.replicate(w, [DEV])[0]
        assert rep is not w and len(list(rep.parameters())) == 0
        return torch.nn.parallel.parallel_apply([rep], [(f, fo)])[0]
    same(run(via_replica, w), ref, "replicate + parallel_apply")
    w.load_state_dict(sd)
    if torch.cuda.device_count() >= 2 and not avs:     # per-replica BN statistics differ by design: the BN-less flavour only
        same(run(torch.nn.DataParallel(w, device_ids=[0, 1]), w), ref, "DataParallel over two GPUs")


def test_cpu_tensors_raise():
    from types import SimpleNamespace
    from dgsct_amd import VisualAdapter
    opt = SimpleNamespace(is_multimodal=1, num_conv_group=2, is_before_layernorm=1, is_post_layernorm=1, num_tokens=4)
    m = VisualAdapter(32, 32, "bottleneck", reduction_factor=8, opt=opt, num_tk=4, conv_dim_in=36, conv_dim_out=16,
                      linear_in=16, linear_out=32)
    with pytest.raises(RuntimeError):
        m(torch.randn(10, 32, 16, 1), torch.randn(10, 16, 36, 1))


# ---------------------------------------------------------------------------------------------------------------------
# Restored in round 4 (deleted without replacement by commit 3451bcf; VERDICT r3 weak #1 / ADVICE r3 medium), on the
# no-floor fp32 metric (helpers.fp32_err / grad_close_fp32(name=)) and with the device's ReLU masks pinned where an oracle
# backward is compared.

# (N, C, No, Co) of AVE stages 0 / 1, visual and audio direction, Swin-V2-B widths (BASELINE config 2): the benchmark's
# dominant shapes (remap GEMMs with K = 4096 / 2304, the C = 128 / 256 kernel instantiations)
STAGE01 = [(2304, 128, 4096, 96), (4096, 96, 2304, 128), (576, 256, 1024, 192), (1024, 192, 576, 256)]


@pytest.mark.parametrize("shape", STAGE01)
def test_real_shapes_stage01_fp32(shape):
    """the large-token stages against the oracle, one clip: out, map, dX, dY and every parameter gradient"""
    r = _real_case(*shape, BT=10, dtype=torch.float32)
    bad = [(k, fp32_err(*r[k])) for k in ("out", "map", "dX", "dY") if not fp32_err(*r[k]) < TOL_F32]
    bad += [(k, fp32_err(g, go)) for k, (g, go) in r["grads"].items() if not grad_close_fp32(g, go, TOL_F32, name=k)]
    assert not bad, bad
    assert not r["extra"], r["extra"]


@pytest.mark.parametrize("shape", STAGE01)
def test_real_shapes_stage01_bf16_outputs(shape):
    """bf16 production kernels at the benchmark's stage-0/1 shapes: out and map against the fp32 oracle on the same
    (bf16-representable) inputs, BASELINE's 1e-2 in relative L2 and 2e-2 of max|ref| in the worst element"""
    r = _real_case(*shape, BT=10, dtype=torch.bfloat16, seed=3)
    for k in ("out", "map"):
        assert _l2(*r[k]) < TOL_BF16, (k, _l2(*r[k]))
        assert nrm_err(*r[k]) < 2 * TOL_BF16, (k, nrm_err(*r[k]))


@pytest.mark.parametrize("shape", [STAGE01[0], STAGE01[1]])
def test_full_size_frames_are_independent(shape):
    """BASELINE size (B=16 x T=10 = 160 frames, stage 0): with BatchNorm in eval mode no operation couples frames, so
    the first clip of the 160-frame call must equal a 10-frame call on the same data (size-independent property;
    exercises the full grids, the large-offset addressing and the XCD tile remap at the benchmark's shapes)."""
    N, C, No, Co = shape
    cfg = O.AdapterConfig(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2)
    p = O.random_params(cfg, "ave", seed=5, scale=0.577)
    spec = spec_of(cfg)
    lib = default_lib()
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-2)):
        params = param_table(p, spec, DEV)
        gen = torch.Generator().manual_seed(7)
        X = torch.randn(160, N, C, generator=gen).to(DEV, dtype)
        Y = torch.randn(160, No, Co, generator=gen).to(DEV, dtype)
        prep = ops.prepare(lib, spec, params, dtype, DEV)
        big = ops.raw_forward(lib, spec, params, prep, X, Y, False)
        small = ops.raw_forward(lib, spec, params, prep, X[:10].contiguous(), Y[:10].contiguous(), False)
        torch.cuda.synchronize()
        assert torch.isfinite(big[0].float()).all()
        assert nrm_err(big[0][:10], small[0].float().cpu()) < tol
        assert nrm_err(big[1][:10], small[1].float().cpu()) < tol
        assert nrm_err(big[0][150:], ops.raw_forward(lib, spec, params, prep, X[150:].contiguous(), Y[150:].contiguous(),
                                                     False)[0].float().cpu()) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL_F32), (torch.bfloat16, 2e-2)])
def test_fused_residual_and_skip(dtype, tol):
    """SURVEY 8f row f2 on the GPU (net_trans.py:894-906): dgsct_adapter_forward_ex(residual) / backward_ex(skip_into_dx)
    against the plain entry points + the caller's own adds."""
    fx = load_golden("ave_orderB")
    lib = default_lib()
    base = run_library(lib, fx, DEV, dtype, training=True)
    Rz = torch.randn(fx["X"].shape, generator=torch.Generator().manual_seed(5))
    r = run_library(lib, fx, DEV, dtype, training=True, residual=Rz)
    assert nrm_err(r["out"], base["out"].float().cpu() + Rz.to(dtype).float()) < tol
    assert nrm_err(r["dX"], base["dX"].float().cpu()) < 1e-5 + (0 if dtype == torch.float32 else 1e-2)
    r = run_library(lib, fx, DEV, dtype, training=True, skip=True)
    X = fx["X"].to(dtype).float(); dO = fx["dOut"].to(dtype).float()
    assert nrm_err(r["out"], base["out"].float().cpu() + X) < tol
    assert nrm_err(r["dX"], base["dX"].float().cpu() + dO) < tol
    assert nrm_err(r["dY"], base["dY"].float().cpu()) < 1e-5 + (0 if dtype == torch.float32 else 1e-2)
    if dtype == torch.float32:      # and against the reference-generated vectors themselves
        assert fp32_err(r["out"], fx["out"] + fx["X"]) < TOL_F32
        assert fp32_err(r["dX"], fx["dX"] + fx["dOut"]) < TOL_F32
        assert fp32_err(r["dY"], fx["dY"]) < TOL_F32


@pytest.mark.parametrize("name", ["ave_orderA", "ave_orderB", "avs_s4", "pretrain"])
def test_results_do_not_depend_on_buffer_contents(name, monkeypatch):
    """every output / scratch / saved-activation / gradient buffer is filled with NaN bit patterns before the call
    (ops._POISON): a kernel that reads memory the call has not written yet would turn the results into NaN"""
    monkeypatch.setattr(ops, "_POISON", True)
    fx = load_golden(name)
    r = run_library(default_lib(), fx, DEV, torch.float32, training=True)
    bad = [(k, fp32_err(r[k], fx[k])) for k in ("out", "map", "dX", "dY") if not fp32_err(r[k], fx[k]) < TOL_F32]
    bad += [(k, fp32_err(r["grads"][k], g)) for k, g in fx["grads"].items() if not grad_close_fp32(r["grads"][k], g, TOL_F32, name=k)]
    assert not bad, bad
    rb = run_library(default_lib(), fx, DEV, torch.bfloat16, training=True, skip=True)
    assert all(torch.isfinite(rb[k].float()).all() for k in ("out", "map", "dX", "dY"))
    assert all(torch.isfinite(g).all() for g in rb["grads"].values())


@pytest.mark.parametrize("shape", [(2304, 128, 4096, 96), (4096, 96, 2304, 128), (576, 256, 1024, 192), (144, 512, 256, 384),
                                   (36, 1024, 64, 768)])
def test_poisoned_buffers_at_production_shapes_bf16(shape, monkeypatch):
    """the same NaN-poison check through the kernels only the production shapes reach (gemm8 split-K atomics, the skinny
    K-split products, rowdot_colsum's partial scratch, modln_gproj's BN sums): bf16, 20 frames, everything finite and the
    outputs within BASELINE's bf16 bound of the oracle"""
    monkeypatch.setattr(ops, "_POISON", True)
    r = _real_case(*shape, BT=20, dtype=torch.bfloat16, seed=9)
    for k in ("out", "map", "dX", "dY"):
        assert torch.isfinite(r[k][0].float()).all(), k
    for k, (g, _) in r["grads"].items():
        assert torch.isfinite(g).all(), k
    for k in ("out", "map"):
        assert _l2(*r[k]) < TOL_BF16, (k, _l2(*r[k]))


@pytest.mark.parametrize("wide", [False, True], ids=["fused_attn", "wide_attn"])
@pytest.mark.parametrize("flat", [False, True], ids=["unflat", "flat"])
def test_stack_on_gpu_matches_reference_fixture(flat, wide, monkeypatch):
    """SURVEY row a-10 on the device (net_trans.py:880-916): 12 adapters through AdapterStack with everything the benchmark
    uses switched on (two adapter streams, aux streams in forward and backward, fused residual/skip, flat parameters)
    against the reference-generated stack fixture: outputs, maps, input gradients and every parameter gradient; repeated
    to give stream-ordering bugs a chance to show."""
    from dgsct_amd import AdapterStack
    from dgsct_amd.stack import default_opt
    if wide:        # (DGSCT_WIDE_ATTN: the num_tokens > 32 attention path for every adapter of the stack)
        monkeypatch.setenv("DGSCT_WIDE_ATTN", "1")
    fx = load_golden("stack_2stage")
    st = AdapterStack(fx["stages"], opt=default_opt(num_tokens=4), concurrent=True)
    st.load_state_dict(fx["state0"])
    st = st.to(DEV)
    if flat:
        st.flatten_parameters()
    st.train()
    for rep in range(3):
        if rep:                                              # BN running stats moved in the previous repetition
            st.load_state_dict(fx["state0"])
        for p in st.parameters():
            p.grad = None
        feats = [(a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)) for a, b in fx["feats"]]
        outs, maps = st(feats)
        bad = []
        for i, ((fv, fa), (rv, ra)) in enumerate(zip(outs, fx["outs"])):
            bad += [(f"out{i}{m}", fp32_err(a, b)) for m, a, b in (("v", fv, rv), ("a", fa, ra)) if not fp32_err(a, b) < TOL_F32]
        bad += [(f"map{i}", fp32_err(maps[i], fx["maps"][i])) for i in (0, 1) if not fp32_err(maps[i], fx["maps"][i]) < TOL_F32]
        torch.autograd.backward([t for pr in outs for t in pr] + [maps[0], maps[1]],
                                [g.to(DEV) for pr in fx["cots"] for g in pr] + [fx["mcots"][0].to(DEV), fx["mcots"][1].to(DEV)])
        # the flat path defers the join of the weight-gradient streams (ops.DEFER_AUX_JOIN): by the time backward() returns, the
        # end-of-backward callback must have ordered this stream behind every one of them -- in EVERY pass (a once-per-pass flag
        # kept in thread-local storage drained the first pass only: backward nodes run on the engine's worker threads)
        assert not ops._PENDING, (rep, list(ops._PENDING))
        torch.cuda.synchronize()
        for i, ((fv, fa), (gv, ga)) in enumerate(zip(feats, fx["dfeats"])):
            bad += [(f"dfeat{i}{m}", fp32_err(a.grad, b)) for m, a, b in (("v", fv, gv), ("a", fa, ga)) if not fp32_err(a.grad, b) < TOL_F32]
        if flat:
            n = 0
            for name, m in st.named_modules():
                if hasattr(m, "flat_param"):
                    for pn, (off, cnt, shape) in m._flat_layout.items():
                        ref = fx["grads"].get(name + "." + pn)
                        if ref is not None:
                            if not grad_close_fp32(m.flat_param.grad[off:off + cnt].view(shape), ref, TOL_F32, name=pn):
                                bad.append((name + "." + pn, fp32_err(m.flat_param.grad[off:off + cnt].view(shape), ref)))
                            n += 1
            assert n == len(fx["grads"])
        else:
            got = {k: p.grad for k, p in st.named_parameters() if p.grad is not None}
            assert set(got) == set(fx["grads"])
            for k, g in fx["grads"].items():
                if not grad_close_fp32(got[k], g, TOL_F32, name=k.split(".", 2)[-1]):
                    bad.append((k, fp32_err(got[k], g)))
        assert not bad, (rep, bad)


@pytest.mark.parametrize("wide", [False, True], ids=["fused_attn", "wide_attn"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_pair_backward_equals_the_two_node_path(dtype, wide, monkeypatch):
    """AdapterStack's one-node path for the two adapters of a position (ops._PairFlatFn: d f = dX(own) + dY(other) formed in the
    epilogue of the product that writes dY, the two halves of each call issued around the other call's) against the two autograd
    nodes + accumulation it replaces, library against library on the same inputs: outputs and maps identical, input and parameter
    gradients equal up to the ONE bf16 rounding the fused sum no longer makes (fp32: up to summation order)."""
    from dgsct_amd import AdapterStack
    from dgsct_amd.stack import default_opt
    if wide:        # the whole stack (two streams, deferred weight gradients, pair nodes) on the num_tokens > 32 attention path
        monkeypatch.setenv("DGSCT_WIDE_ATTN", "1")
    fx = load_golden("stack_2stage")
    res = {}
    for pair in (False, True):
        st = AdapterStack(fx["stages"], opt=default_opt(num_tokens=4), concurrent=True, pair_backward=pair, compute_dtype=dtype)
        st.load_state_dict(fx["state0"])
        st = st.to(DEV).flatten_parameters()
        st.train()
        feats = [(a.to(DEV, dtype).requires_grad_(True), b.to(DEV, dtype).requires_grad_(True)) for a, b in fx["feats"]]
        r = {}
        for rep in range(2):
            st.load_state_dict(fx["state0"])
            for p in st.parameters():
                p.grad = None
            for a, b in feats:
                a.grad = b.grad = None
            outs, maps = st(feats)
            torch.autograd.backward([t for pr in outs for t in pr] + [maps[0], maps[1]],
                                    [g.to(DEV, dtype) for pr in fx["cots"] for g in pr] + [fx["mcots"][0].to(DEV), fx["mcots"][1].to(DEV)])
            assert not ops._PENDING
            torch.cuda.synchronize()
        for i, (a, b) in enumerate(outs):
            r[f"out{i}v"], r[f"out{i}a"] = a.detach().float(), b.detach().float()
        r["map0"], r["map1"] = maps[0].detach().float(), maps[1].detach().float()
        for i, (a, b) in enumerate(feats):
            r[f"dfeat{i}v"], r[f"dfeat{i}a"] = a.grad.float(), b.grad.float()
        for name, m in st.named_modules():
            if hasattr(m, "flat_param"):
                r[name] = m.flat_param.grad.clone()
        res[pair] = r
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    bad = []
    for k, a in res[False].items():
        b = res[True][k]
        assert torch.isfinite(b).all(), k
        e = ((a - b).norm() / a.norm().clamp_min(1e-30)).item()
        if k.startswith(("out", "map")):
            if not torch.equal(a, b) and e > 1e-6:          # (atomics in the forward reductions: not bit-reproducible run to run)
                bad.append((k, e))
        elif e > tol:
            bad.append((k, e))
    assert not bad, bad


@pytest.mark.parametrize("num_tk", [8, None])
def test_module_dropin_matches_oracle(num_tk):
    """nn.Module boundary on the GPU: reference call convention ([BT,C,N,1] views), state_dict names, autograd -- against the
    oracle (forward values, input gradients, every parameter gradient, BN buffers).  num_tk omitted: the reference constructor's
    default of 87 latent tokens (net_trans.py:437), i.e. the num_tokens > 32 path."""
    from types import SimpleNamespace
    from dgsct_amd import VisualAdapter
    opt = SimpleNamespace(is_multimodal=1, num_conv_group=2, is_before_layernorm=1, is_post_layernorm=1, num_tokens=8)
    torch.manual_seed(0)
    kw = dict(num_tk=num_tk) if num_tk else {}
    m = VisualAdapter(64, 64, "bottleneck", reduction_factor=8, opt=opt, use_bn=True, use_gate=True,
                      conv_dim_in=49, conv_dim_out=25, linear_in=48, linear_out=64, **kw).to(DEV)
    assert m.my_tokens.shape == (num_tk or 87, 64)
    with torch.no_grad():
        m.gate.fill_(0.7); m.gate_av.fill_(0.3)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    cfg = O.AdapterConfig(N=25, C=64, No=49, Co=48, tk=num_tk or 87, r=8, g=2)
    BT = 10
    f = torch.randn(BT, 25, 64, device=DEV, requires_grad=True)
    fo = torch.randn(BT, 49, 48, device=DEV, requires_grad=True)
    out, amap = m(f.permute(0, 2, 1).unsqueeze(-1), fo.permute(0, 2, 1).unsqueeze(-1))
    assert out.shape == (BT, 64, 25, 1) and amap.shape == (BT, 1, 25)
    g_out = torch.randn_like(out); g_map = torch.randn_like(amap)
    (out * g_out).sum().add((amap * g_map).sum()).backward()
    po = {k: v.clone() for k, v in sd.items()}
    out_o, map_o, _, s = O.forward(po, f.detach().cpu(), fo.detach().cpu(), cfg, training=True)
    dX_o, dY_o, g_o = O.backward(po, s, cfg, g_out.squeeze(-1).permute(0, 2, 1).cpu(), g_map.squeeze(1).cpu(), None)
    assert fp32_err(out.squeeze(-1).permute(0, 2, 1), out_o) < TOL_F32
    assert fp32_err(amap.squeeze(1), map_o) < TOL_F32
    assert fp32_err(f.grad, dX_o) < TOL_F32 and fp32_err(fo.grad, dY_o) < TOL_F32
    for k, p in m.named_parameters():
        if k in g_o:
            assert grad_close_fp32(p.grad, g_o[k].reshape(p.shape), TOL_F32, name=k), (k, fp32_err(p.grad, g_o[k].reshape(p.shape)))
        else:
            assert p.grad is None or k in ("gate_tk",), k
    assert int(m.bn1.num_batches_tracked) == 1
    assert fp32_err(m.bn2.running_mean, po["bn2.running_mean"]) < TOL_F32
    assert fp32_err(m.bn2.running_var, po["bn2.running_var"]) < TOL_F32


def test_deferred_aux_join_gives_the_joined_results():
    """round 5 (ops.DEFER_AUX_JOIN, DGSCT_BWD_NO_JOIN): backward calls that return with their weight gradients still running on the aux
    stream -- three calls in a row on one stream (alternating workspaces, each call letting go of the buffers of the one before), then
    drain_aux() -- against the same calls with the join inside; fp32, so only the atomic summation order differs."""
    N, C, No, Co, BT = 144, 64, 96, 48, 8
    cfg = O.AdapterConfig(N=N, C=C, No=No, Co=Co, tk=8, r=8, g=2)
    p = O.random_params(cfg, "ave", seed=5)
    spec = spec_of(cfg)
    lib = default_lib()
    params = param_table(p, spec, DEV)
    gen = torch.Generator().manual_seed(3)
    res = {}
    for defer in (False, True):
        outs = []
        prep = ops.prepare(lib, spec, params, torch.float32, DEV)
        for rep in range(3):
            g2 = torch.Generator().manual_seed(10 + rep)
            X = torch.randn(BT, N, C, generator=g2).to(DEV).contiguous()
            Y = torch.randn(BT, No, Co, generator=g2).to(DEV).contiguous()
            dOut = torch.randn(BT, N, C, generator=g2).to(DEV).contiguous()
            dMap = torch.randn(BT, N, generator=g2).to(DEV)
            out, amap, _, saved, d = ops.raw_forward(lib, spec, [t.clone() if t is not None else None for t in params], prep, X, Y, True)
            dX, dY, gflat = ops.raw_backward(lib, spec, d, params, prep, X, Y, saved, dOut, dMap, None, flat_out=True, defer_join=defer)
            del saved, X, Y, dOut, dMap                    # (the deferred path keeps what the aux stream still reads alive itself)
            outs.append((dX, dY, gflat))
        ops.drain_aux()
        torch.cuda.synchronize()
        res[defer] = [(a.clone(), b.clone(), c.clone()) for a, b, c in outs]
    for (a0, b0, c0), (a1, b1, c1) in zip(res[False], res[True]):
        assert fp32_err(a1, a0) < 1e-4 and fp32_err(b1, b0) < 1e-4
        assert fp32_err(c1, c0) < 1e-3, fp32_err(c1, c0)


def test_flat_gradients_accumulate_over_two_backward_passes():
    """the deferred join (ops.DEFER_AUX_JOIN) must not reach gradient ACCUMULATION: from the second backward pass on, AccumulateGrad
    adds the new flat gradient into the existing one on the calling stream right after the call, so the call has to join its
    weight-gradient stream itself (ops._AdapterFlatFn: defers only while `flat.grad is None`).  Two passes without zero_grad on the
    same inputs must give exactly twice the gradients of one."""
    from dgsct_amd import AdapterStack
    from dgsct_amd.stack import default_opt
    fx = load_golden("stack_2stage")
    grads = {}
    for passes in (1, 2):
        st = AdapterStack(fx["stages"], opt=default_opt(num_tokens=4), concurrent=True)
        st.load_state_dict(fx["state0"])
        st = st.to(DEV)
        st.flatten_parameters()
        st.train()
        for _ in range(passes):
            st.load_state_dict(fx["state0"])                      # same weights and BN buffers for every pass (keeps .grad)
            feats = [(a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)) for a, b in fx["feats"]]
            outs, maps = st(feats)
            torch.autograd.backward([t for pr in outs for t in pr] + [maps[0], maps[1]],
                                    [g.to(DEV) for pr in fx["cots"] for g in pr] + [fx["mcots"][0].to(DEV), fx["mcots"][1].to(DEV)])
            assert not ops._PENDING
        torch.cuda.synchronize()
        grads[passes] = {n: m.flat_param.grad.clone() for n, m in st.named_modules() if hasattr(m, "flat_param")}
    assert grads[1]
    for n, g1 in grads[1].items():
        assert fp32_err(grads[2][n], 2 * g1) < 1e-3, (n, fp32_err(grads[2][n], 2 * g1))


def test_joined_call_after_a_deferred_one_waits_for_its_weight_gradients():
    """a deferred backward (aux join left to the caller) followed on the same stream by a call that does NOT defer -- a
    forward, or a joined backward: both take workspace slot 0, which the deferred call's weight-gradient kernels may still be reading on
    the aux stream.  ops._wait_pending orders the stream behind that aux event first.  The deferred call's gradients must equal the
    joined reference whatever follows it."""
    N, C, No, Co, BT = 144, 64, 96, 48, 8
    cfg = O.AdapterConfig(N=N, C=C, No=No, Co=Co, tk=8, r=8, g=2)
    p = O.random_params(cfg, "ave", seed=5)
    spec = spec_of(cfg)
    lib = default_lib()
    params = param_table(p, spec, DEV)
    prep = ops.prepare(lib, spec, params, torch.float32, DEV)

    def one(seed):
        g2 = torch.Generator().manual_seed(seed)
        X = torch.randn(BT, N, C, generator=g2).to(DEV).contiguous()
        Y = torch.randn(BT, No, Co, generator=g2).to(DEV).contiguous()
        dOut = torch.randn(BT, N, C, generator=g2).to(DEV).contiguous()
        dMap = torch.randn(BT, N, generator=g2).to(DEV)
        return X, Y, dOut, dMap

    def ref(seed):
        X, Y, dOut, dMap = one(seed)
        _, _, _, saved, d = ops.raw_forward(lib, spec, params, prep, X, Y, True)
        r = ops.raw_backward(lib, spec, d, params, prep, X, Y, saved, dOut, dMap, None, flat_out=True, defer_join=False)
        torch.cuda.synchronize()
        return [t.clone() for t in r]

    want = {s: ref(s) for s in (21, 22)}
    for follow in ("forward", "joined_backward"):
        ops.drain_aux()
        X, Y, dOut, dMap = one(21)
        _, _, _, saved, d = ops.raw_forward(lib, spec, params, prep, X, Y, True)
        X2, Y2, dOut2, dMap2 = one(22)
        _, _, _, saved2, d2 = ops.raw_forward(lib, spec, params, prep, X2, Y2, True)
        got = ops.raw_backward(lib, spec, d, params, prep, X, Y, saved, dOut, dMap, None, flat_out=True, defer_join=True)
        assert ops._PENDING                                         # slot 0, still running on the aux stream
        if follow == "forward":
            ops.raw_forward(lib, spec, params, prep, X2, Y2, True)   # overwrites workspace slot 0 on this stream
            assert not ops._PENDING                                 # ... after having waited for the pending call
        else:
            got2 = ops.raw_backward(lib, spec, d2, params, prep, X2, Y2, saved2, dOut2, dMap2, None, flat_out=True, defer_join=False)
            assert not ops._PENDING
        ops.drain_aux()
        torch.cuda.synchronize()
        for a, b in zip(got, want[21]):
            assert fp32_err(a, b) < 1e-3, (follow, fp32_err(a, b))
        if follow != "forward":
            for a, b in zip(got2, want[22]):
                assert fp32_err(a, b) < 1e-3, (follow, fp32_err(a, b))


def test_deferral_conditions_of_the_flat_gradient():
    """ADVICE r5 (ops._may_adopt): the aux join may be deferred only when autograd adopts the flat gradient unread.  (a) ONE adapter
    called twice in a graph: the engine sums the two gradient buffers, so both calls must join -- the result is the sum of the two calls'
    gradients; (b) a foreign post-accumulate-grad hook (what torch DDP registers) reads .grad at once: it must see the complete
    gradient; (c) grad mode / tensor hooks switch the deferral off as well."""
    from dgsct_amd import VisualAdapter
    from dgsct_amd.stack import default_opt
    N, C, No, Co, BT = 64, 64, 36, 32, 10

    def build():
        torch.manual_seed(11)
        m = VisualAdapter(input_dim=C, output_dim=C, adapter_kind="bottleneck", dim_list=[C], layer_idx=0, reduction_factor=8,
                          opt=default_opt(num_tokens=8), use_bn=True, use_gate=True, conv_dim_in=No, conv_dim_out=N, linear_in=Co,
                          linear_out=C, num_tk=8).to(DEV)
        with torch.no_grad():
            m.gate.fill_(0.5); m.gate_av.fill_(0.5)
        m.flatten_parameters()
        m.train()
        return m

    gen = torch.Generator().manual_seed(2)
    xs = [torch.randn(BT, N, C, generator=gen).to(DEV) for _ in range(2)]
    ys = [torch.randn(BT, No, Co, generator=gen).to(DEV) for _ in range(2)]
    view = lambda f: f.permute(0, 2, 1).unsqueeze(-1)

    def run(m, idx, hook=None):
        m.flat_param.grad = None
        h = m.flat_param.register_post_accumulate_grad_hook(hook) if hook else None
        loss = 0
        for i in idx:
            out, amap = m(view(xs[i]), view(ys[i]))
            loss = loss + (out.float() ** 2).sum() + amap.sum()
        loss.backward()
        if h is not None:
            h.remove()
        torch.cuda.synchronize()
        return m.flat_param.grad.clone()

    m = build()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    singles = []
    for i in range(2):
        m.load_state_dict(sd)
        singles.append(run(m, [i]))
    m.load_state_dict(sd)
    both = run(m, [0, 1])                                           # (a) same flat parameter twice in one graph
    # BatchNorm running stats differ between the orders but do not enter training-mode gradients
    assert fp32_err(both, singles[0] + singles[1]) < 1e-3, fp32_err(both, singles[0] + singles[1])
    assert getattr(m.flat_param, "_dgsct_uses", 0) == 0
    seen = {}
    m.load_state_dict(sd)

    def foreign(param):                                             # (b) reads the gradient right away, on the calling stream
        seen["g"] = param.grad.clone()

    g = run(m, [0], hook=foreign)
    torch.cuda.synchronize()
    assert fp32_err(seen["g"], singles[0]) < 1e-3 and fp32_err(g, singles[0]) < 1e-3
    # (c) the predicate itself
    fp = m.flat_param
    fp.grad = None
    with torch.no_grad():
        assert ops._may_adopt(fp)
    assert not ops._may_adopt(fp)                                   # grad mode on (create_graph): AccumulateGrad clones
    with torch.no_grad():
        h = fp.register_hook(lambda g_: g_)
        assert not ops._may_adopt(fp)
        h.remove()
        fp._dgsct_uses = 2
        assert not ops._may_adopt(fp)
        fp._dgsct_uses = 0
