"""TEST INFRASTRUCTURE: golden vectors of a Swin-V2 block -> tests/golden/swinv2_block.pt   (SURVEY.md 8(f) row f4).

The visual backbone's blocks live in timm==0.6.12 (the reference's requirements.txt:39; called at DG-SCT/AVE/nets/net_trans.py:894, :903),
which is neither vendored in /root/reference nor installed here: dg-sct_amd/backbone.py's ``SwinV2Block`` restates the published block
(Liu et al., "Swin Transformer V2", CVPR 2022: res-post-norm, scaled-cosine window attention with a clamped learned logit scale, log-spaced
continuous position bias through a 2-layer MLP, cyclic shift with a -100 mask) under timm's attribute names -- parity against timm itself
stays UNPINNED.  What this script adds is a pin against an INDEPENDENT implementation of the same published block that IS importable in
this image: ``transformers.models.swinv2.modeling_swinv2.Swinv2Layer`` (Hugging Face transformers 5.15.0; its own code base, ported from
the official Swin-V2 release).  The HF layer is built on the CPU with perturbed random weights, its parameters are renamed to timm's
layout (q / k / v projections concatenated into ``attn.qkv``, ``query.bias`` -> ``attn.q_bias``, ...), ``SwinV2Block`` must reproduce
its output and input gradient to 1e-5 in fp32, and parameters / input / cotangent / output / input gradient are stored as the fixture
(data only: no transformers or timm source is stored in this repo; the buffers that are pure functions of the geometry -- shift mask,
offset table, pair index -- are rebuilt by the block's constructor and not stored).
Run here:  python oracle/make_golden_swinv2.py
"""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name: (dim, map side, heads, window, shift)
CASES = {"plain": (32, 24, 1, 12, 0), "shifted": (32, 24, 2, 12, 6), "one_window": (96, 6, 3, 12, 0), "small_shifted": (32, 16, 4, 8, 4)}


def to_timm_names(sd):
    """HF Swinv2Layer state_dict -> the key set of timm's SwinTransformerV2Block (what dg-sct_amd/backbone.py's SwinV2Block carries)"""
    a = "attention.self."
    out = {
        "attn.logit_scale": sd[a + "logit_scale"],
        "attn.cpb_mlp.0.weight": sd[a + "continuous_position_bias_mlp.0.weight"],
        "attn.cpb_mlp.0.bias": sd[a + "continuous_position_bias_mlp.0.bias"],
        "attn.cpb_mlp.2.weight": sd[a + "continuous_position_bias_mlp.2.weight"],
        "attn.qkv.weight": torch.cat([sd[a + "query.weight"], sd[a + "key.weight"], sd[a + "value.weight"]], 0),
        "attn.q_bias": sd[a + "query.bias"],
        "attn.v_bias": sd[a + "value.bias"],
        "attn.proj.weight": sd["attention.output.dense.weight"],
        "attn.proj.bias": sd["attention.output.dense.bias"],
        "norm1.weight": sd["layernorm_before.weight"], "norm1.bias": sd["layernorm_before.bias"],     # (HF's names: both norms are POST-norms)
        "mlp.fc1.weight": sd["intermediate.dense.weight"], "mlp.fc1.bias": sd["intermediate.dense.bias"],
        "mlp.fc2.weight": sd["output.dense.weight"], "mlp.fc2.bias": sd["output.dense.bias"],
        "norm2.weight": sd["layernorm_after.weight"], "norm2.bias": sd["layernorm_after.bias"],
    }
    return {k: v.detach().clone() for k, v in out.items()}


def main():
    import transformers
    from transformers import Swinv2Config
    from transformers.models.swinv2.modeling_swinv2 import Swinv2Layer
    spec = importlib.util.spec_from_file_location("_bb", os.path.join(ROOT, "dg-sct_amd", "backbone.py"))
    bb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bb)
    cases = {}
    for name, (dim, res, heads, ws, shift) in CASES.items():
        torch.manual_seed(21)
        cfg = Swinv2Config(embed_dim=dim, depths=[1], num_heads=[heads], window_size=ws, mlp_ratio=4.0, hidden_dropout_prob=0.0,
                           attention_probs_dropout_prob=0.0, drop_path_rate=0.0, hidden_act="gelu", layer_norm_eps=1e-5)
        ref = Swinv2Layer(cfg, dim=dim, input_resolution=(res, res), num_heads=heads, shift_size=shift).eval()
        with torch.no_grad():                                   # make every parameter matter (biases / LayerNorm affine start at 0 / 1)
            for p in ref.parameters():
                p.add_(0.1 * torch.randn_like(p))
        mine = bb.SwinV2Block(dim, (res, res), heads, window_size=ws, shift_size=shift, fused=False).eval()
        sd = to_timm_names(ref.state_dict())
        missing = mine.load_timm_state_dict(sd, strict=False)
        # attn_mask: a pure function of the geometry (timm stores it as a buffer, HF rebuilds it per call); nothing else may be absent
        assert set(missing.missing_keys) <= {"attn_mask", "attn.relative_coords_table", "attn.relative_position_index"} and not missing.unexpected_keys, missing
        g = torch.Generator().manual_seed(5)
        x = torch.randn(1 if res > 16 else 2, res * res, dim, generator=g)
        cot = torch.randn(x.shape, generator=g)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr, (res, res))[0]
        yr.backward(cot)
        xm = x.clone().requires_grad_(True)
        ym = mine(xm)
        ym.backward(cot)
        ey, ed = (ym - yr).abs().max().item(), (xm.grad - xr.grad).abs().max().item()
        print(f"{name:14s} dim {dim} map {res}^2 heads {heads} window {mine.window_size} shift {mine.shift_size}:  |y - y_hf| = {ey:.2e}   |dx - dx_hf| = {ed:.2e}")
        assert ey < 1e-5 and ed < 1e-5, (name, ey, ed)
        cases[name] = dict(cfg=(dim, res, heads, ws, shift), state=sd, x=x, cot=cot, y=yr.detach(), dx=xr.grad.detach(),
                           source=f"transformers {transformers.__version__} Swinv2Layer")
    path = os.path.join(ROOT, "tests", "golden", "swinv2_block.pt")
    torch.save(cases, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
