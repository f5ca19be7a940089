"""CPU-only: the real kernel schedule + C ABI (csrc/plan.cpp, csrc/capi.cpp) linked against host-loop
primitives (tests/emu) must reproduce the reference golden vectors for every flavour -- checks operand
roles, strides, buffer offsets and the gradient layout without a GPU."""
import os
import sys

import pytest
import torch

from helpers import golden_names, load_golden, rel_err, run_library, nrm_err

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
from build_emu import build_emu  # noqa: E402

from dgsct_amd._lib import Lib  # noqa: E402
from oracle import dgsct_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def emu():
    return Lib(build_emu())


@pytest.mark.parametrize("name", golden_names())
def test_schedule_matches_reference_fp32(emu, name):
    fx = load_golden(name)
    r = run_library(emu, fx, torch.device("cpu"), torch.float32, training=True)
    tol = 1e-4      # fp32; relative to max(1, max|ref|)
    assert rel_err(r["out"], fx["out"]) < tol
    assert rel_err(r["map"], fx["map"]) < tol
    if fx["tmap"] is not None:
        assert rel_err(r["tmap"], fx["tmap"]) < tol
    assert rel_err(r["dX"], fx["dX"]) < tol
    assert rel_err(r["dY"], fx["dY"]) < tol
    for k, g in fx["grads"].items():
        assert k in r["grads"], f"library produced no gradient for {k}"
        assert rel_err(r["grads"][k].reshape(g.shape), g) < tol, k
    assert not (set(r["grads"]) - set(fx["grads"])), "library produced gradients the reference does not"
    # BatchNorm running statistics after the training step
    from dgsct_amd._lib import P_INDEX
    for k, v in fx["buffers1"].items():
        if "running" in k:
            assert rel_err(r["params"][P_INDEX[k]], v) < tol, k


@pytest.mark.parametrize("name", golden_names())
def test_schedule_without_the_gate_fusion_fp32(emu, name):
    """the emulation takes the fused gate / bottleneck passes (fused_gate.hip's schedule branch) for every shape by default;
    with dgsct_test_tune("gatefuse", 0) the same goldens go through the unfused launches the device uses for C > 256 and fp32"""
    fx = load_golden(name)
    old = emu.test_tune("gatefuse", 0)
    try:
        r = run_library(emu, fx, torch.device("cpu"), torch.float32, training=True)
    finally:
        emu.test_tune("gatefuse", old)
    tol = 1e-4
    for k in ("out", "map", "dX", "dY"):
        assert rel_err(r[k], fx[k]) < tol, k
    for k, g in fx["grads"].items():
        assert rel_err(r["grads"][k].reshape(g.shape), g) < tol, k


@pytest.mark.parametrize("name", golden_names())
def test_schedule_on_the_wide_token_path_fp32(emu, name, monkeypatch):
    """num_tokens > 32 runs the two latent-token attentions as batched products + row softmax (csrc/attn_wide.cpp; the goldens
    ave_tk40 / ave_tk87 take that path by themselves).  DGSCT_WIDE_ATTN=1 sends every tk down it: all reference goldens check it."""
    monkeypatch.setenv("DGSCT_WIDE_ATTN", "1")
    fx = load_golden(name)
    r = run_library(emu, fx, torch.device("cpu"), torch.float32, training=True)
    tol = 1e-4
    for k in ("out", "map", "dX", "dY"):
        assert rel_err(r[k], fx[k]) < tol, k
    for k, g in fx["grads"].items():
        assert rel_err(r["grads"][k].reshape(g.shape), g) < tol, k


@pytest.mark.parametrize("mask", [0, 31, 4 + 32, 1 + 2, 8 + 16])
@pytest.mark.parametrize("name", ["ave_orderA", "ave_orderB", "pretrain", "avs_s4", "avqa"])
def test_schedule_with_the_fused_gemm_hooks_fp32(emu, name, mask):
    """round 5 (csrc/gemm_fx.hip): the backward products of the unfused (late-stage) schedule take the elementwise launch in front of
    them as a transform of their A operand (ReLU backward of vq1 / vq2; BatchNorm backward of dO / dZ) and the channel-gate backward as
    an epilogue.  dgsct_test_tune("gemmfx", mask) picks the call sites (1 dZ, 2 dX3, 4 dXc, 8 dX1; 16: prologues at every width; 32: dXc
    with its epilogue only); 0 = the separate launches.  DGSCT_NO_GPROJ=1 sends the tiny golden bottlenecks through the GEMM branch the
    late stages use (the BatchNorm-backward sites live there); "gatefuse" = 0 selects the unfused gate chain."""
    fx = load_golden(name)
    old, oldg = emu.test_tune("gemmfx", mask), emu.test_tune("gatefuse", 0)
    os.environ["DGSCT_NO_GPROJ"] = "1"
    try:
        r = run_library(emu, fx, torch.device("cpu"), torch.float32, training=True)
    finally:
        emu.test_tune("gemmfx", old)
        emu.test_tune("gatefuse", oldg)
        os.environ.pop("DGSCT_NO_GPROJ", None)
    tol = 1e-4
    for k in ("out", "map", "dX", "dY"):
        assert rel_err(r[k], fx[k]) < tol, k
    for k, g in fx["grads"].items():
        assert rel_err(r["grads"][k].reshape(g.shape), g) < tol, k


@pytest.mark.parametrize("name", ["ave_orderA", "ave_orderB", "pretrain", "avqa"])
def test_schedule_without_the_folded_gate_products_fp32(emu, name):
    """by default the schedule folds the [BT, C] elementwise launches of the gate-MLP chain into its skinny products (SkFuse:
    m1 / sigmoid' on the operand, dm1's two consumers in the epilogue, both `da` products in one tile) wherever
    skinny_fused_supported says so -- everywhere in the emulation; with dgsct_test_tune("skfuse", 0) the same goldens go through
    the separate launches (the device's fp32 path)"""
    fx = load_golden(name)
    old = emu.test_tune("skfuse", 0)
    try:
        r = run_library(emu, fx, torch.device("cpu"), torch.float32, training=True)
    finally:
        emu.test_tune("skfuse", old)
    tol = 1e-4
    for k in ("out", "map", "dX", "dY"):
        assert rel_err(r[k], fx[k]) < tol, k
    for k, g in fx["grads"].items():
        assert rel_err(r["grads"][k].reshape(g.shape), g) < tol, k


@pytest.mark.parametrize("name", ["ave_orderA", "ave_orderB", "pretrain", "avs_s4"])
def test_schedule_with_vq1_materialised_fp32(emu, name):
    """by default the emulation takes the schedule branch in which vq1 = relu(X1 Wv1^T + b) is never stored (vq1sum_fwd /
    vq1_bwd: the device's stage-0 path); with dgsct_test_tune("vq1fuse", 0) the same goldens go through the product + column sum +
    ReLU backward + product launches"""
    fx = load_golden(name)
    tol = 1e-4
    old = emu.test_tune("vq1fuse", 3)        # 3: dWv1 accumulated inside vq1_bwd (the experiment variant of the schedule)
    os.environ["DGSCT_VQ1_DW"] = "1"         # its scratch region is laid out only on request (plan.cpp: layout())
    try:
        r3 = run_library(emu, fx, torch.device("cpu"), torch.float32, training=True)
        emu.test_tune("vq1fuse", 0)
        r = run_library(emu, fx, torch.device("cpu"), torch.float32, training=True)
    finally:
        emu.test_tune("vq1fuse", old)
        os.environ.pop("DGSCT_VQ1_DW", None)
    for k, g in fx["grads"].items():
        assert rel_err(r3["grads"][k].reshape(g.shape), g) < tol, k
    assert rel_err(r3["dX"], fx["dX"]) < tol
    for k in ("out", "map", "dX", "dY"):
        assert rel_err(r[k], fx[k]) < tol, k
    for k, g in fx["grads"].items():
        assert rel_err(r["grads"][k].reshape(g.shape), g) < tol, k


@pytest.mark.parametrize("name", ["ave_orderA", "avs_s4", "avqa"])
def test_schedule_eval_mode(emu, name):
    fx = load_golden(name)
    fx = dict(fx)
    st = dict(fx["state0"])
    st.update(fx["buffers1"])          # eval uses the running stats after the training step
    fx["state0"] = st
    r = run_library(emu, fx, torch.device("cpu"), torch.float32, training=False)
    assert rel_err(r["out"], fx["eval_out"]) < 1e-4
    assert rel_err(r["map"], fx["eval_map"]) < 1e-4


@pytest.mark.parametrize("name", ["ave_orderA", "ave_orderB", "pretrain", "ave_tk40"])
def test_schedule_bf16_storage(emu, name):
    """bf16 storage through the same schedule (host emulation rounds to bf16 at every store).  ave_tk40: the bf16 branches of the
    num_tokens > 32 path (hi / lo token operands, the E copy of dtok; csrc/attn_wide.cpp)."""
    fx = load_golden(name)
    r = run_library(emu, fx, torch.device("cpu"), torch.bfloat16, training=True)
    # bf16 storage of every intermediate on a tiny, un-averaged problem.  The gradient error is one realisation of the
    # rounding noise that the un-scaled token attention amplifies: 0.8 % .. 8 % across the
    # flavours, and it moves inside that band whenever any intermediate is rounded differently.
    wide = fx["cfg"]["tk"] > 32
    assert nrm_err(r["out"], fx["out"]) < 3e-2
    assert nrm_err(r["dX"], fx["dX"]) < (0.12 if wide else 8e-2)
    assert nrm_err(r["dY"], fx["dY"]) < 8e-2


@pytest.mark.parametrize("name", ["ave_orderA", "ave_orderB", "avs_s4"])
def test_fused_residual_and_skip(emu, name):
    """`f = f + adapter(...)[0]` (net_trans.py:894-906) fused into the adapter's last kernel.
    residual=R: out = R + ref_out, gradients unchanged.  skip: out = X + ref_out and dX = ref_dX + dOut."""
    fx = load_golden(name)
    tol = 1e-4
    Rz = torch.randn(fx["X"].shape, generator=torch.Generator().manual_seed(5))
    r = run_library(emu, fx, torch.device("cpu"), torch.float32, training=True, residual=Rz)
    assert rel_err(r["out"], fx["out"] + Rz) < tol
    assert rel_err(r["dX"], fx["dX"]) < tol and rel_err(r["dY"], fx["dY"]) < tol
    r = run_library(emu, fx, torch.device("cpu"), torch.float32, training=True, skip=True)
    assert rel_err(r["out"], fx["out"] + fx["X"]) < tol
    assert rel_err(r["map"], fx["map"]) < tol
    assert rel_err(r["dX"], fx["dX"] + fx["dOut"]) < tol
    assert rel_err(r["dY"], fx["dY"]) < tol
    for k, g in fx["grads"].items():
        assert rel_err(r["grads"][k].reshape(g.shape), g) < tol, k


@pytest.mark.parametrize("shape", [(4, 9, 16), (3, 7, 36), (2, 36, 64)])
def test_map_pool_c_abi_host_loops(emu, shape):
    """dgsct_map_pool_forward / _backward (net_trans.py:922-924) through the C ABI on the host-loop
    primitives: argument order, strides and the fp32 [BT,N] / [BT,C] layouts against the oracle restatement."""
    BT, N, C = shape
    g = torch.Generator().manual_seed(5)
    F = torch.randn(BT, N, C, generator=g)
    amap = torch.softmax(torch.randn(BT, 1, N, generator=g), dim=-1)
    dP = torch.randn(BT, 1, C, generator=g)
    m2, dp2 = amap.reshape(BT, N).contiguous(), dP.reshape(BT, C).contiguous()
    pooled, dF, dmap = torch.empty(BT, C), torch.empty(BT, N, C), torch.empty(BT, N)
    emu.map_pool_forward(0, BT, N, C, F.data_ptr(), m2.data_ptr(), pooled.data_ptr(), None)
    emu.map_pool_backward(0, BT, N, C, F.data_ptr(), m2.data_ptr(), dp2.data_ptr(), dF.data_ptr(), dmap.data_ptr(), None)
    rdF, rdmap = O.map_pool_bwd(F, amap, dP)
    assert torch.allclose(pooled.double(), O.map_pool(F, amap)[:, 0], rtol=1e-5, atol=1e-6)
    assert torch.allclose(dF.double(), rdF, rtol=1e-6, atol=1e-7)
    assert torch.allclose(dmap.double(), rdmap[:, 0], rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        emu.map_pool_forward(7, BT, N, C, F.data_ptr(), m2.data_ptr(), pooled.data_ptr(), None)


@pytest.mark.parametrize("name", golden_names())
def test_two_part_backward_equals_the_one_part_call(emu, name):
    """dgsct_adapter_backward_ex2 (include/dgsct.h): HOLD_DY + ONLY_DY on the same workspace = the plain call, for every flavour (both remap
    associations, bicubic, temporal); with dy_residual the product's epilogue adds it; the misuse cases return an error, not garbage."""
    import ctypes as C
    from dgsct_amd import ops
    from dgsct_amd._lib import BwdOpts, BWD_HOLD_DY, BWD_ONLY_DY, _PP
    from helpers import param_table, spec_of
    fx = load_golden(name)
    dev = torch.device("cpu")
    r = run_library(emu, fx, dev, torch.float32, training=True)
    spec, params, d = r["spec"], r["params"], r["desc"]
    X, Y = fx["X"].contiguous(), fx["Y"].contiguous()
    prep = ops.prepare(emu, spec, params, torch.float32, dev)
    _, _, _, saved, d = ops.raw_forward(emu, spec, params, prep, X, Y, True)
    dOut, dMap = fx["dOut"].contiguous(), fx["dMap"]
    dTm = fx["dTmap"] if fx["dTmap"] is not None else None
    dX, ws, grads = ops.raw_backward(emu, spec, d, params, prep, X, Y, saved, dOut, dMap, dTm, flat_out=True, hold_dy=True)
    extra = torch.randn_like(Y)
    dY = ops.raw_backward_dy(emu, d, params, prep, ws, Y)
    dY2 = ops.raw_backward_dy(emu, d, params, prep, ws, Y, residual=extra)            # (the workspace is read, not consumed)
    assert rel_err(dX, r["dX"]) < 1e-6 and rel_err(dY, r["dY"]) < 1e-6
    assert rel_err(dY2, r["dY"] + extra) < 1e-6
    lay = ops.grad_layout(emu, d)
    for i, (off, n) in enumerate(lay):
        if off >= 0 and r["grads"].get(__import__("dgsct_amd")._lib.PARAM_NAMES[i]) is not None:
            g = r["grads"][__import__("dgsct_amd")._lib.PARAM_NAMES[i]]
            assert rel_err(grads[off:off + n].view(g.shape), g) < 1e-6, i
    # misuse: both parts at once; a residual for the part that does not write dY; ONLY_DY without a workspace
    ptrs = C.cast(ops._ptrs(params), _PP)
    def call(flags, resid=None, ws_ptr=ws.data_ptr(), dY_ptr=dY.data_ptr()):
        o = BwdOpts(flags, resid, None, None)
        return emu.c.dgsct_adapter_backward_ex2(C.byref(d), ptrs, prep.data_ptr(), X.data_ptr(), Y.data_ptr(), saved.data_ptr(), dOut.data_ptr(),
                                                dMap.data_ptr(), None, dX.data_ptr(), dY_ptr, grads.data_ptr(), ws_ptr, None, None, C.byref(o))
    assert call(BWD_HOLD_DY | BWD_ONLY_DY) != 0 and b"two parts" in emu.c.dgsct_last_error()
    assert call(BWD_HOLD_DY, resid=extra.data_ptr()) != 0 and b"dy_residual" in emu.c.dgsct_last_error()
    assert call(BWD_ONLY_DY, ws_ptr=None) != 0 and b"NULL" in emu.c.dgsct_last_error()
