"""GPU unit tests of the fused latent-token attention kernels (csrc/attn.hip) against a plain torch fp32 reference of the
same op, through the C ABI test hook (dgsct_test_attn).  Shapes: ragged token counts (tail chunks), channel counts that
are not a multiple of the 64/128-channel slabs, tk < 32, multi-chunk frames."""
import pytest
import torch

from attn_ref import AttnCall, ref_tokattn_bwd, ref_tokattn_fwd, ref_xattn_bwd, ref_xattn_fwd
from dgsct_amd._lib import default_lib

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
SHAPES = [(3, 36, 64, 4), (2, 300, 96, 32), (4, 144, 512, 32), (2, 64, 1536, 32), (3, 257, 160, 2), (2, 1030, 128, 32)]


def l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def tol(dtype):
    return 2e-5 if dtype == torch.float32 else 6e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_tokattn_fwd(shape, dtype):
    c = AttnCall(default_lib(), dtype, *shape, DEV)
    c.run(4)                                   # packed my_tokens (dgsct_prepare's job): enables the short-frame kernel
    c.run(0)
    torch.cuda.synchronize()
    tok, lse, a = ref_tokattn_fwd(c.Yp, c.T0)
    assert l2(c.d["tok"], tok) < tol(dtype)
    assert l2(c.d["lse"], lse) < 1e-5
    assert l2(c.d["a"], a) < 1e-5 and l2(c.d["aE"], a) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_xattn_fwd(shape, dtype):
    c = AttnCall(default_lib(), dtype, *shape, DEV)
    c.run(4)
    c.run(0)                                   # produces tok and its packed bf16 images on the device
    tok = c.d["tok"].float().cpu()
    c.run(1)
    torch.cuda.synchronize()
    assert l2(c.d["out"], ref_xattn_fwd(c.X, tok, c.g)) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_r2", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_xattn_bwd(shape, with_r2, dtype):
    c = AttnCall(default_lib(), dtype, *shape, DEV)
    c.run(4)
    c.run(0)
    tok = c.d["tok"].float().cpu()
    if not with_r2:
        c.args.R2 = None
    c.run(2)
    torch.cuda.synchronize()
    dX, dtok, dgate = ref_xattn_bwd(c.X, c.dX1, tok, c.g, c.R2 if with_r2 else None)
    t = tol(dtype)
    assert l2(c.d["out"], dX) < t
    assert l2(c.d["dtok"], dtok) < 2 * t
    assert abs(float(c.d["dgate"]) - float(dgate)) < 2 * t * max(1.0, abs(float(dgate)), float((c.dX1.norm() * tok.norm())) * 1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_tokattn_bwd(shape, dtype):
    c = AttnCall(default_lib(), dtype, *shape, DEV)
    tok, lse, _ = ref_tokattn_fwd(c.Yp, c.T0)
    c.d["tok"].copy_(tok)
    c.d["lse"].copy_(lse)
    c.d["dtok"].copy_(c.dtok_in)
    c.run(4)                                   # packed my_tokens (dgsct_prepare's job)
    c.run(3)
    torch.cuda.synchronize()
    dYp, dT0b = ref_tokattn_bwd(c.Yp, c.T0, c.dtok_in, c.da, 1.0 / shape[1])
    t = tol(dtype)
    assert l2(c.d["out"], dYp) < t
    assert l2(c.d["dT0b"], dT0b) < 2 * t
