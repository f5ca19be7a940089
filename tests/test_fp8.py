"""fp8 (OCP e4m3) MFMA projections -- BASELINE.json configs[4] "bf16 + fp8-MFMA projections" (dtype DGSCT_BF16_FP8).

There is no reference arithmetic for fp8 (the reference is fp32 throughout), so the tolerance is set against the fp32 oracle
and stated here: OUTPUTS (out, map) within 3e-2 relative L2 (measured 1.2-1.4e-2 with the ideal host emulation at the AVQA
Swin-V2-L shapes: the ~4 % e4m3 operand error on the `fc` projection, averaged over its contraction); the backward runs in
bf16 on the fp8-perturbed activations, and the un-scaled latent-token softmax turns the perturbation of Yp into tens of
percent on dY (same mechanism as DESIGN.md section 7), so gradients are checked for finiteness, dX within 0.3, and
against the host emulation of the same schedule rather than against fp32.  The kernel itself is checked against torch's
float8_e4m3fn arithmetic."""
import dataclasses
import os
import sys

import pytest
import torch

from helpers import ROOT, load_golden, oracle_cfg, param_table, run_library, spec_of
from dgsct_amd import ops
from dgsct_amd._lib import Lib, default_lib
from oracle import dgsct_oracle as O

sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
from build_emu import build_emu  # noqa: E402


def _l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def _case(lib, dev, N, C, No, Co, flavour="avqa", BT=10, seed=0, over=None):
    cfg = O.AdapterConfig(**{**dict(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2), **O.FLAVOURS[flavour], **(over or {})})
    p = O.random_params(cfg, flavour, seed=seed, scale=0.577)
    gen = torch.Generator().manual_seed(seed + 1)
    rb = lambda t: t.bfloat16().float()
    X, Y, dOut = rb(torch.randn(BT, N, C, generator=gen)), rb(torch.randn(BT, No, Co, generator=gen)), rb(torch.randn(BT, N, C, generator=gen))
    dMap = torch.randn(BT, N, generator=gen)
    po = {k: v.clone() for k, v in p.items()}
    out_o, map_o, _, s = O.forward(po, X, Y, cfg, training=True)
    dX_o, dY_o, _ = O.backward(po, s, cfg, dOut, dMap, None, training=True)
    spec = dataclasses.replace(spec_of(cfg), fp8=True)
    params = param_table(p, spec, dev)
    dt = torch.bfloat16
    prep = ops.prepare(lib, spec, params, dt, dev)
    Xd, Yd = X.to(dev, dt).contiguous(), Y.to(dev, dt).contiguous()
    out, amap, _, saved, d = ops.raw_forward(lib, spec, params, prep, Xd, Yd, True)
    dX, dY, grads = ops.raw_backward(lib, spec, d, params, prep, Xd, Yd, saved, dOut.to(dev, dt).contiguous(), dMap.to(dev), None)
    return dict(out=out, map=amap, dX=dX, dY=dY, grads=grads, ref=(out_o, map_o, dX_o, dY_o))


def test_fp8_schedule_on_host_emulation():
    """the fp8 schedule (prepare-time per-tensor quantisation, three fp8 GEMMs, bf16 backward) through the C ABI on the host
    emulation: golden `avqa` configuration (tk = 2, g = 4, no BN)"""
    emu = Lib(build_emu())
    r = _case(emu, torch.device("cpu"), 16, 32, 36, 16, BT=10, over=dict(tk=2))
    out_o, map_o, dX_o, _ = r["ref"]
    assert _l2(r["out"], out_o) < 6e-2 and _l2(r["map"], map_o) < 1e-2      # tiny widths (K = 16 / 32): little averaging
    assert all(torch.isfinite(g).all() for g in r["grads"] if g is not None)
    from dgsct_amd.ops import AdapterSpec
    with pytest.raises(RuntimeError, match="multiples of 16"):
        emu.query(AdapterSpec(N=16, C=32, No=36, Co=24, tk=2, fp8=True).desc(10, torch.bfloat16, True))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,relu", [(300, 96, 96, True), (1000, 128, 256, False), (77, 40, 1536, True), (4096, 768, 384, True)])
def test_fp8_gemm_kernel_vs_torch_float8(M, N, K, relu):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).bfloat16()
    W = torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    scale = 448.0 / W.abs().max()
    W8 = (W * scale).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    A8 = A.float().clamp(-448, 448).to(torch.float8_e4m3fn).float()
    ref = (A8.double() @ W8.double().t() / scale.double() + bias.double()).float()
    if relu:
        ref = ref.relu()
    Ad, Wd, bd = A.to(dev), W.to(dev).contiguous(), bias.to(dev)
    D = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    w8 = torch.empty(N * K, device=dev, dtype=torch.uint8)
    sc = torch.zeros(4, device=dev)
    default_lib().test_gemm_fp8(M, N, K, Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), relu, D.data_ptr(), w8.data_ptr(), sc.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert abs(float(sc[0]) * float(scale) - 1.0) < 1e-6
    got8 = w8.cpu().view(torch.float8_e4m3fn).float().view(N, K)
    assert (got8 != W8).float().mean().item() < 1e-3          # same codes (a product within 1 ulp of a rounding tie may differ)
    assert _l2(D, ref) < 4e-3                                                              # bf16 output rounding only


@pytest.mark.gpu
@pytest.mark.parametrize("shape,use_gate", [((144, 768, 256, 384), True), ((2304, 192, 4096, 96), True), ((36, 1536, 64, 768), True),
                                            ((256, 384, 144, 768), False)])
def test_fp8_projections_avqa_swin_large(shape, use_gate):
    """configs[4]: AVQA flavour (tk = 2, g = 4, no BN; audio adapters without gate) at Swin-V2-L widths"""
    r = _case(default_lib(), torch.device("cuda:0"), *shape, over=dict(use_gate=use_gate))
    torch.cuda.synchronize()
    out_o, map_o, dX_o, _ = r["ref"]
    assert torch.isfinite(r["out"].float()).all()
    assert _l2(r["out"], out_o) < 3e-2, _l2(r["out"], out_o)
    assert _l2(r["map"], map_o) < 5e-3, _l2(r["map"], map_o)
    assert _l2(r["dX"], dX_o) < 0.3, _l2(r["dX"], dX_o)
    assert torch.isfinite(r["dY"].float()).all() and all(torch.isfinite(g).all() for g in r["grads"] if g is not None)


FP8_FIXTURE_SHAPE = (64, 96, 36, 64)


@pytest.mark.gpu
def test_fp8_gpu_matches_host_emulation():
    """the HIP fp8 path against the host emulation of the same schedule (same quantisation, fp64 accumulation), taken from
    the committed fixture (oracle/make_golden_fp8.py) so that the GPU run loads libdgsct.so only"""
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "fp8_emu_ave_64x96.pt"))
    assert tuple(fx["shape"]) == FP8_FIXTURE_SHAPE
    a = _case(default_lib(), torch.device("cuda:0"), *FP8_FIXTURE_SHAPE, flavour="ave")
    torch.cuda.synchronize()
    assert _l2(a["out"], fx["out"]) < 1e-2 and _l2(a["map"], fx["map"]) < 1e-3


def test_fp8_fixture_is_current():
    """CPU: the committed fixture is what the host emulation produces today"""
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "fp8_emu_ave_64x96.pt"))
    b = _case(Lib(build_emu()), torch.device("cpu"), *FP8_FIXTURE_SHAPE, flavour="ave")
    assert _l2(b["out"], fx["out"]) < 1e-6 and _l2(b["map"], fx["map"]) < 1e-6
