"""GPU parity of the spatial-map pooling (reference DG-SCT/AVE/nets/net_trans.py:922-924) through the C
ABI (dgsct_map_pool_forward / _backward) against the oracle restatement and against torch.bmm, the reference's own op."""
import pytest
import torch

from dgsct_amd import map_pool
from dgsct_amd._lib import default_lib
from oracle import dgsct_oracle as O

pytestmark = pytest.mark.gpu

# last-stage AVE shapes (visual 36 x 1024, audio 64 x 768 at BT = 160), a stage-0 shape, ragged / unaligned ones
SHAPES = [(160, 36, 1024), (160, 64, 768), (20, 2304, 128), (3, 7, 36), (1, 1, 8), (5, 300, 40)]


def _inputs(BT, N, C, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    F = torch.randn(BT, N, C, generator=g).to(dtype)
    amap = torch.softmax(torch.randn(BT, 1, N, generator=g), dim=-1)
    dP = torch.randn(BT, 1, C, generator=g)
    return F, amap, dP


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", SHAPES)
def test_map_pool_c_abi_matches_oracle(shape, dtype):
    BT, N, C = shape
    lib = default_lib()
    F, amap, dP = _inputs(BT, N, C, dtype)
    dev = torch.device("cuda:0")
    Fd, md, dPd = F.to(dev), amap.reshape(BT, N).to(dev).contiguous(), dP.reshape(BT, C).to(dev).contiguous()
    pooled = torch.full((BT, C), float("nan"), device=dev)
    dF = torch.full((BT, N, C), float("nan"), device=dev).to(dtype)
    dmap = torch.full((BT, N), float("nan"), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    code = 0 if dtype == torch.float32 else 1
    lib.map_pool_forward(code, BT, N, C, Fd.data_ptr(), md.data_ptr(), pooled.data_ptr(), st)
    lib.map_pool_backward(code, BT, N, C, Fd.data_ptr(), md.data_ptr(), dPd.data_ptr(), dF.data_ptr(), dmap.data_ptr(), st)
    torch.cuda.synchronize()
    ref = O.map_pool(F.float(), amap)                       # oracle on the SAME (bf16-rounded) inputs: sums are fp32 on the GPU
    rdF, rdmap = O.map_pool_bwd(F.float(), amap, dP)
    scale = lambda t: t.abs().max().clamp_min(1e-30)
    # tolerances: fp32 accumulation of <= 2304 products -> 1e-5 of the largest value; dF is ROUNDED to bf16 in bf16 mode (2^-8)
    assert ((pooled.cpu().double() - ref[:, 0]).abs().max() / scale(ref)) < 1e-5
    assert ((dmap.cpu().double() - rdmap[:, 0]).abs().max() / scale(rdmap)) < 1e-5
    assert ((dF.cpu().double() - rdF).abs().max() / scale(rdF)) < (1e-6 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_map_pool_autograd_matches_bmm(dtype):
    """the reference's own op: torch.bmm(spatial_att_maps, f) and its autograd (net_trans.py:922-924)"""
    BT, N, C = 40, 36, 1024
    dev = torch.device("cuda:0")
    F, amap, dP = _inputs(BT, N, C, dtype, seed=3)
    f1 = F.to(dev).requires_grad_(True); m1 = amap.to(dev).requires_grad_(True)
    f2 = F.to(dev).float().requires_grad_(True); m2 = amap.to(dev).requires_grad_(True)
    out = map_pool(f1, m1)
    assert out.shape == (BT, 1, C) and out.dtype == dtype
    ref = torch.bmm(m2, f2)
    out.backward(dP.to(dev).to(dtype)); ref.backward(dP.to(dev).to(dtype).float())
    tol = 1e-5 if dtype == torch.float32 else 1e-2      # bf16: the OUTPUT and dF are rounded to bf16
    rel = lambda a, b: ((a.float() - b.float()).abs().max() / b.float().abs().max()).item()
    assert rel(out, ref) < tol and rel(f1.grad, f2.grad) < tol and rel(m1.grad, m2.grad) < 1e-5 + (tol if dtype != torch.float32 else 0)


def test_map_pool_uniform_map_is_the_token_mean():
    """size-independent property at the full AVE size: a uniform map pools to the mean over tokens (the line the reference
    replaced, `f_v.mean(dim=1, keepdim=True)`, net_trans.py:921)"""
    BT, N, C = 160, 2304, 128
    dev = torch.device("cuda:0")
    F = torch.randn(BT, N, C, device=dev).to(torch.bfloat16)
    out = map_pool(F, torch.full((BT, 1, N), 1.0 / N, device=dev))
    assert (out.float() - F.float().mean(dim=1, keepdim=True)).abs().max().item() < 2e-3


def test_map_pool_rejects_cpu_and_bad_shapes():
    with pytest.raises(RuntimeError):
        map_pool(torch.randn(2, 4, 8), torch.randn(2, 1, 4))
    with pytest.raises(RuntimeError):
        map_pool(torch.randn(2, 4, 8, device="cuda:0"), torch.randn(2, 1, 5, device="cuda:0"))
    with pytest.raises(RuntimeError):                       # C must be a multiple of 4 (include/dgsct.h)
        map_pool(torch.randn(2, 4, 6, device="cuda:0"), torch.randn(2, 1, 4, device="cuda:0"))
