"""bf16 parity anchored to the REFERENCE: the device's bf16 results are compared with the reference's fp32 arithmetic
WITHOUT device-supplied ReLU masks and held to what the reference module itself loses when it is run in bf16 (`module.bfloat16()` on the
CPU, measured by oracle/make_golden_refbf16.py against the imported reference class, DG-SCT/AVE/nets/net_trans.py:433-674; scalars in
tests/golden/ref_bf16.pt).  Also here: fp32 real-shape cases with NO pinned masks, the ReLU flips counted, and
eval-mode (running BatchNorm statistics) forward parity at a real shape in fp32 and bf16."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import (FP32_RESIDUES, O, PARAM_NAMES, device_relu_masks, fp32_err, grad_close_fp32, load_golden, ops, oracle_cfg, param_table, spec_of)  # noqa: E402
from oracle import make_golden_refbf16 as RB  # noqa: E402  (inputs_of / params_of only: nothing in there needs /root/reference at import)

FX = load_golden("ref_bf16")
CASES = sorted(FX["cases"])
K = 1.25          # device(bf16) may be at most this much further from reference(fp32) than reference(bf16) is: out, map, dX, dY and the
HEADLINE = ("out", "map", "dX", "dY", "dfc.weight", "dconv_adapter.weight", "dmy_tokens")     # remap / token gradients
K_OTHER = 1.5     # every other weight MATRIX (measured 0.02-1.23: the gate-MLP weights sit at ~1.0-1.2, one ReLU flip moves them by a row)


def _l2(a, b):
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _oracle_fp32(case):
    cfg = oracle_cfg(case["cfg"])
    p = RB.params_of(cfg, case["seed"])
    X, Y, dOut, dMap = RB.inputs_of(cfg, case["seed"])
    po = {k: v.clone() for k, v in p.items()}
    out, amap, _, s = O.forward(po, X, Y, cfg, training=True)
    dX, dY, g = O.backward(po, s, cfg, dOut, dMap, None, training=True)
    res = {"out": out, "map": amap, "dX": dX, "dY": dY, **{"d" + k: v for k, v in g.items()}}
    return cfg, p, (X, Y, dOut, dMap), res, s


@pytest.mark.parametrize("name", CASES)
def test_box_oracle_is_the_reference_result_of_the_fixture(name):
    """CPU: the fp32 yardstick recomputed on this machine (same seeds -> same tensors, oracle fp32) has the norms and the seeded
    random projections of the reference(fp32) tensors recorded when the fixture was made -- i.e. it IS the reference's result for
    this case to ~1e-4, although the tensors themselves (megabytes) are not shipped; the shipped 2 x 8 x 8 corners agree element-wise."""
    case = FX["cases"][name]
    _, _, _, res, _ = _oracle_fp32(case)
    assert max(case["oracle_err"].values()) <= 1e-4            # oracle vs imported reference, measured at generation
    for j, k in enumerate(sorted(case["norm"])):
        if k[1:] in FP32_RESIDUES:                                  # analytically zero / eps-sized residues: their value is rounding noise
            continue
        v = res[k]
        n_ref, p_ref = case["norm"][k], case["proj"][k]
        assert abs(v.norm().item() - n_ref) <= 2e-4 * n_ref, (k, v.norm().item(), n_ref)
        got = (v.reshape(-1) * RB.proj_vec(v.numel(), 9000 + j)).sum().item()
        assert abs(got - p_ref) <= 1e-3 * n_ref, (k, got, p_ref)     # |<v - ref, g>| ~ ||v - ref|| for a unit-variance Gaussian g
    for k, c in case["corner"].items():
        v = res[k]
        v = v.reshape(v.shape[0], v.shape[1], -1) if v.dim() > 2 else v.reshape(1, *v.shape)
        assert torch.allclose(v[:2, :8, :8], c, rtol=2e-3, atol=1e-4 * float(c.abs().max())), k


def _device(case, dtype, training=True, eval_state=None):
    from dgsct_amd._lib import default_lib
    dev = torch.device("cuda:0")
    cfg, p, (X, Y, dOut, dMap), res, s = _oracle_fp32(case)
    spec = spec_of(cfg)
    lib = default_lib()
    params = param_table(p, spec, dev)
    Xd, Yd = X.to(dev, dtype).contiguous(), Y.to(dev, dtype).contiguous()
    prep = ops.prepare(lib, spec, params, dtype, dev)
    out, amap, _, saved, d = ops.raw_forward(lib, spec, params, prep, Xd, Yd, True)
    torch.cuda.synchronize()
    masks = device_relu_masks(lib, d, saved, spec, X.shape[0], dtype) if os.environ.get("DGSCT_GATEFUSE_READABLE") else None
    dX, dY, grads = ops.raw_backward(lib, spec, d, params, prep, Xd, Yd, saved, dOut.to(dev, dtype).contiguous(), dMap.to(dev), None)
    torch.cuda.synchronize()
    got = {"out": out, "map": amap, "dX": dX, "dY": dY}
    for i, g in enumerate(grads):
        if g is not None:
            got["d" + PARAM_NAMES[i]] = g
    return cfg, spec, lib, params, res, got, s, masks


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_bf16_is_no_further_from_the_reference_than_the_reference_in_bf16(name):
    """|| device(bf16) - reference(fp32) || <= 1.25 x || reference(bf16) - reference(fp32) ||, rel-L2, un-pinned (no device masks), for
    out, map, dX, dY, dWc, dWn, d my_tokens (1.5 x for the other weight matrices); BASELINE's 1e-2 for the outputs on top.
    Measured (round 6): dX 0.66-0.84 x, dY 0.27-0.79 x, dWc 0.28-0.75 x, dWn 0.27-0.70 x, out 0.61-0.69 x, map 0.02 x."""
    case = FX["cases"][name]
    _, _, _, _, res, got, _, _ = _device(case, torch.bfloat16)
    rb = case["ref_bf16_err"]
    rows, bad = [], []
    for k in sorted(rb):
        if k not in got or k not in res:
            continue
        ref = res[k].reshape(got[k].shape) if res[k].numel() == got[k].numel() else None
        if ref is None:
            continue
        is_mat = k in ("out", "map", "dX", "dY") or (ref.dim() >= 2 and min(ref.shape[:2]) > 1)
        e = _l2(got[k], ref)
        rows.append((k, e, rb[k]))
        if is_mat and e > (K if k in HEADLINE else K_OTHER) * rb[k] + 1e-4:
            bad.append((k, e, rb[k]))
    print(f"\n{name}: tensor, device(bf16) vs reference(fp32), reference(bf16) vs reference(fp32)   [rel-L2]")
    for k, e, r in rows:
        print(f"   {k:34s} {e:10.3e} {r:10.3e}   x{e / max(r, 1e-30):.2f}")
    assert not bad, bad
    assert _l2(got["out"], res["out"]) < 1e-2 and _l2(got["map"], res["map"]) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_fp32_real_shapes_without_pinned_masks(name):
    """fp32 at real widths against the oracle's OWN ReLU decisions (the other real-shape fp32 tests differentiate
    the branches the device took).  A pre-activation within fp32 rounding of zero may land on the other side; the flips are counted
    (device decisions read from the saved buffer vs the oracle's) and reported, data gradients are held to 1e-3 (fp32_err: relative L2
    and worst element), weight gradients to 1e-3 with grad_close_fp32's two-row allowance."""
    case = FX["cases"][name]
    from dgsct_amd._lib import default_lib
    lib = default_lib()
    old = (lib.test_tune("gatefuse", 2), lib.test_tune("vq1fuse", 2))         # keep the fused passes' ReLU decisions readable (values unchanged)
    os.environ["DGSCT_GATEFUSE_READABLE"] = "1"
    try:
        cfg, spec, lib, params, res, got, s, masks = _device(case, torch.float32)
    finally:
        os.environ.pop("DGSCT_GATEFUSE_READABLE", None)
        lib.test_tune("gatefuse", old[0]); lib.test_tune("vq1fuse", old[1])
    flips = {}
    omask = {"vq1": s["vq1"] > 0, "vq2": s["vq2"] > 0, "Z": s["Z"] > 0, "q": s["q"] > 0, "aq1": s["aq1"] > 0, "aq2": s["aq2"] > 0}
    for k, m in masks.items():
        flips[k] = int((m.reshape(-1) != omask[k].reshape(-1)).sum())
    print(f"\n{name}: ReLU decisions that differ device vs oracle (fp32): {flips}  of {sum(m.numel() for m in masks.values())}")
    for k in ("out", "map", "dX", "dY"):
        assert fp32_err(got[k], res[k]) < 1e-3, (k, fp32_err(got[k], res[k]), flips)
    bad = []
    for k, g in got.items():
        if k in ("out", "map", "dX", "dY") or k not in res:
            continue
        if not grad_close_fp32(g, res[k].reshape(g.shape) if res[k].numel() == g.numel() else res[k], 1e-3, name=k[1:]):
            bad.append((k, fp32_err(g, res[k].reshape(g.shape))))
    assert not bad, (bad, flips)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("name", ["c512", "c1024"])
def test_eval_mode_at_real_shapes(name, dtype):
    """model.eval(): BatchNorm uses its running statistics (net_trans.py:636,643 in eval mode).  A training step first moves the
    running buffers off their initial 0 / 1 (device and oracle each update their own copy: compared), then the eval forward of both
    on those buffers: fp32 1e-3, bf16 1e-2 (BASELINE.json)."""
    from dgsct_amd._lib import default_lib
    dev = torch.device("cuda:0")
    case = FX["cases"][name]
    cfg = oracle_cfg(case["cfg"])
    p = RB.params_of(cfg, case["seed"])
    X, Y, _, _ = RB.inputs_of(cfg, case["seed"])
    po = {k: v.clone() for k, v in p.items()}
    O.forward(po, X, Y, cfg, training=True)                       # updates po's running stats
    out_e, map_e, _, _ = O.forward(po, X, Y, cfg, training=False)
    spec = spec_of(cfg)
    lib = default_lib()
    params = param_table(p, spec, dev)
    Xd, Yd = X.to(dev, dtype).contiguous(), Y.to(dev, dtype).contiguous()
    prep = ops.prepare(lib, spec, params, dtype, dev)
    ops.raw_forward(lib, spec, params, prep, Xd, Yd, True)        # training step on the device: running stats updated in `params`
    out, amap, _, _, _ = ops.raw_forward(lib, spec, params, prep, Xd, Yd, False)
    torch.cuda.synchronize()
    tol = 1e-3 if dtype == torch.float32 else 1e-2
    for nm in ("bn1.running_mean", "bn1.running_var", "bn2.running_mean", "bn2.running_var"):
        got = params[PARAM_NAMES.index(nm)]
        assert fp32_err(got, po[nm]) < (1e-3 if dtype == torch.float32 else 2e-2), (nm, fp32_err(got, po[nm]))
    if dtype == torch.float32:
        assert fp32_err(out, out_e) < tol and fp32_err(amap, map_e) < tol, (fp32_err(out, out_e), fp32_err(amap, map_e))
    else:
        assert _l2(out, out_e) < tol and _l2(amap, map_e) < tol, (_l2(out, out_e), _l2(amap, map_e))
