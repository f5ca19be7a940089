"""``TemporalAttention`` drop-in (reference net_trans.py:182-251) against the reference-generated golden
vectors of tests/golden/temporal.pt (oracle/make_golden_temporal.py): parameters by seed + checksum (12 M parameters are not
committed), outputs, input gradients, and the norm / sum of every parameter gradient.  The CPU test drives the real module
through the host-emulated gate kernel (state_dict / wiring / autograd plumbing); the GPU tests run the HIP gate kernel and
compare it with the torch restatement of the same op."""
import os
import sys

import pytest
import torch

from helpers import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
from build_emu import build_emu  # noqa: E402

from dgsct_amd._lib import Lib, default_lib  # noqa: E402
from dgsct_amd.temporal import TemporalAttention  # noqa: E402
from oracle import temporal_oracle as TO  # noqa: E402


def _fixture():
    return torch.load(os.path.join(GOLDEN, "temporal.pt"), weights_only=False)


def _build(fx, lib, device):
    torch.manual_seed(fx["seed"])
    m = TemporalAttention(lib=lib).eval()
    sd = m.state_dict()
    assert list(sd) == fx["keys"]                                       # reference names, reference order
    for k, v in sd.items():
        assert abs(float(v.double().sum()) - fx["param_sum"][k]) <= 1e-9 * max(1.0, abs(fx["param_sum"][k])), k
    return m.to(device)


def _check(m, fx, device, tol):
    fv, fa = fx["fv"].to(device).requires_grad_(True), fx["fa"].to(device).requires_grad_(True)
    ov, oa, og = m(fv, fa)
    assert ov.shape == (fx["T"], fx["B"], 256) and og.shape == (fx["T"], fx["B"], 1)
    for got, ref in ((ov, fx["out_v"]), (oa, fx["out_a"]), (og, fx["gate"])):
        assert (got.detach().cpu() - ref).abs().max().item() < tol
    torch.autograd.backward([ov, oa, og], [fx["cv"].to(device), fx["ca"].to(device), fx["cg"].to(device)])
    assert (fv.grad.cpu() - fx["d_fv"]).abs().max().item() < tol * max(1.0, fx["d_fv"].abs().max().item())
    assert (fa.grad.cpu() - fx["d_fa"]).abs().max().item() < tol * max(1.0, fx["d_fa"].abs().max().item())
    got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert set(got) == set(fx["grad_norm"])
    assert sorted(k for k, p in m.named_parameters() if p.grad is None) == fx["no_grad"]
    for k, g in got.items():
        n = float(g.double().norm())
        assert abs(n - fx["grad_norm"][k]) <= 10 * tol * max(1e-3, fx["grad_norm"][k]), (k, n, fx["grad_norm"][k])
        assert abs(float(g.double().sum()) - fx["grad_sum"][k]) <= 10 * tol * max(1.0, fx["grad_norm"][k]), k


def test_temporal_attention_matches_reference_cpu():
    fx = _fixture()
    m = _build(fx, Lib(build_emu()), torch.device("cpu"))
    _check(m, fx, torch.device("cpu"), 2e-5)


def test_temporal_no_cpu_fallback():
    m = TemporalAttention().eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.randn(2, 10, 1536), torch.randn(2, 10, 768))


@pytest.mark.gpu
def test_temporal_attention_matches_reference_gpu():
    fx = _fixture()
    dev = torch.device("cuda:0")
    m = _build(fx, None, dev)
    # MIOpen's LSTM backward only exists in training mode: train() with every dropout probability set to 0 is the eval-mode
    # function the fixture was generated with
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        elif isinstance(mod, (torch.nn.MultiheadAttention, torch.nn.LSTM)):
            mod.dropout = 0.0
    _check(m, fx, dev, 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("R,D", [(30, 256), (7, 64), (333, 1024), (160, 260)])
def test_temporal_gate_kernel_vs_torch(R, D):
    """dgsct_temporal_gate_forward/backward against plain torch fp32 (the oracle's gate arithmetic) on ragged sizes"""
    from dgsct_amd.temporal import _TemporalGateFn
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(R + D)
    mk = lambda *s: torch.randn(*s, generator=g)
    akv, vkv, vq, aq = (mk(R, D) for _ in range(4))
    wa, wv, ba, bv = mk(1, D) * 0.1, mk(1, D) * 0.1, mk(1), mk(1)
    cv, ca, cg = mk(R, D), mk(R, D), mk(R, 1)
    ins = [t.clone().requires_grad_(True) for t in (akv, vkv, vq, aq, wa, ba, wv, bv)]
    ga = torch.sigmoid(ins[0] @ ins[4].t() + ins[5]); gv = torch.sigmoid(ins[1] @ ins[6].t() + ins[7])
    ref = (ins[2] + ga * ins[2] * TO.GAMMA, ins[3] + gv * ins[3] * TO.GAMMA, ga * gv)
    torch.autograd.backward(list(ref), [cv, ca, cg])
    dins = [t.detach().to(dev).requires_grad_(True) for t in (akv, vkv, vq, aq, wa, ba, wv, bv)]
    out = _TemporalGateFn.apply(default_lib(), TO.GAMMA, *dins)
    torch.autograd.backward(list(out), [cv.to(dev), ca.to(dev), cg.to(dev)])
    for a, b in zip(out, ref):
        assert (a.detach().cpu() - b.detach()).abs().max().item() < 1e-5
    for a, b in zip(dins, ins):
        assert (a.grad.cpu() - b.grad).abs().max().item() < 1e-4 * max(1.0, b.grad.abs().max().item())


# ---- the AVVP and AVS copies of the class (mgn.py:107-159, avs_s4/model/PVT_AVSModel.py:447-582) -------------------------------
def _variant(name):
    from dgsct_amd.temporal import TemporalAttentionAVS, TemporalAttentionAVVP
    fx = torch.load(os.path.join(GOLDEN, f"temporal_{name}.pt"), weights_only=False)
    return fx, (TemporalAttentionAVVP if name == "avvp" else TemporalAttentionAVS)


def _flat(x):
    out = []
    for t in (x if isinstance(x, (list, tuple)) else [x]):
        out += list(t) if isinstance(t, (list, tuple)) else [t]
    return out


def _run_variant(name, lib, device, tol):
    fx, cls = _variant(name)
    torch.manual_seed(fx["seed"])
    m = cls(lib=lib).eval()
    sd = m.state_dict()
    assert list(sd) == fx["keys"]                                       # reference names, reference order
    for k, v in sd.items():
        assert abs(float(v.double().sum()) - fx["param_sum"][k]) <= 1e-9 * max(1.0, abs(fx["param_sum"][k])), k
    m = m.to(device)
    if device.type == "cuda":
        # MIOpen's LSTM backward only exists in training mode: train() with every dropout probability set to 0 is the eval-mode
        # function the fixture was generated with
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            elif isinstance(mod, (torch.nn.MultiheadAttention, torch.nn.LSTM)):
                mod.dropout = 0.0
    ins = [t.to(device).requires_grad_(True) for t in fx["inputs"]]
    args = (ins[0], ins[1]) if name == "avvp" else (ins[:4], ins[4])
    outs = _flat(m(*args))
    assert len(outs) == len(fx["outs"])
    for got, ref in zip(outs, fx["outs"]):
        assert got.shape == ref.shape and (got.detach().cpu() - ref).abs().max().item() < tol
    torch.autograd.backward(outs, [c.to(device) for c in fx["cots"]])
    for t, ref in zip(ins, fx["d_inputs"]):
        assert (t.grad.cpu() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    got = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert set(got) == set(fx["grad_norm"])
    assert sorted(k for k, p in m.named_parameters() if p.grad is None) == fx["no_grad"]
    for k, g in got.items():
        n = float(g.double().norm())
        assert abs(n - fx["grad_norm"][k]) <= 10 * tol * max(1e-3, fx["grad_norm"][k]), (k, n, fx["grad_norm"][k])


@pytest.mark.parametrize("name", ["avvp", "avs"])
def test_temporal_variants_match_reference_cpu(name):
    """state_dict keys / seeded parameters / outputs / input gradients / every parameter-gradient norm of the reference's AVVP and
    AVS `TemporalAttention`, with the per-frame gate application on the host emulation of the HIP kernel"""
    _run_variant(name, Lib(build_emu()), torch.device("cpu"), 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["avvp", "avs"])
def test_temporal_variants_match_reference_gpu(name):
    _run_variant(name, default_lib(), torch.device("cuda:0"), 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(50, 256, 14, 14), (10, 256, 56, 56), (320, 128), (7, 8)])
def test_frame_scale_kernel(dtype, shape):
    """dgsct_frame_scale_*: y = x (1 + gamma g[frame]) on [B*5, C, H, W] decoder maps / [B*10, 128] features, both ways"""
    from dgsct_amd.temporal import frame_scale
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=gen).to(dtype)
    g = torch.rand(shape[0], generator=gen)
    cot = torch.randn(*shape, generator=gen).to(dtype)
    xd, gd = x.to(dev).requires_grad_(True), g.to(dev).requires_grad_(True)
    y = frame_scale(xd, gd, 0.05)
    y.backward(cot.to(dev))
    xr, gr = x.float().requires_grad_(True), g.clone().requires_grad_(True)
    yr = xr * (1 + 0.05 * gr.view(-1, *([1] * (len(shape) - 1))))
    yr.backward(cot.float())
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert (y.float().cpu() - yr).abs().max().item() <= tol * max(1.0, yr.abs().max().item())
    assert (xd.grad.float().cpu() - xr.grad).abs().max().item() <= tol * max(1.0, xr.grad.abs().max().item())
    assert (gd.grad.cpu() - gr.grad).abs().max().item() <= (1e-4 if dtype == torch.float32 else 2e-2) * max(1.0, gr.grad.abs().max().item())
