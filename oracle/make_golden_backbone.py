"""TEST INFRASTRUCTURE: golden vectors of the reference HTS-AT block -> tests/golden/htsat_block.pt   (SURVEY.md 8(f) row f4).

Runs HERE (where /root/reference exists): ``SwinTransformerBlock`` / ``WindowAttention`` (DG-SCT/AVE/nets/htsat.py:50-251) and the
helpers they need from the reference's own layers.py are taken from the reference sources with ``ast`` (no reference text is stored
in this repo), run on CPU, and compared with dg-sct_amd/backbone.py's ``HTSATBlock`` loaded with the same state_dict (asserted
<= 1e-5 on the output, the attention tensor and the input gradient).  Fixture: state_dict, input, cotangent, output, input gradient
for an un-shifted and a shifted block (window 8 on a 16 x 16 map), and a block whose map is as large as its window.

The Swin-V2 blocks of the visual backbone live in timm==0.6.12 (requirements.txt:39), which is neither vendored nor installed:
parity against timm unpinned (oracle/make_golden_swinv2.py pins the block against an independent implementation instead).
"""
import ast
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/DG-SCT/AVE/nets"


def extract(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    return "\n\n".join(ast.get_source_segment(src, n) for n in tree.body
                       if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names)


def reference_block():
    import collections.abc
    import math
    import warnings
    from itertools import repeat
    import torch.nn as nn
    ns = dict(torch=torch, nn=nn, math=math, warnings=warnings, repeat=repeat, collections=collections)
    exec(extract(os.path.join(REF, "layers.py"), {"_ntuple", "drop_path", "DropPath", "Mlp", "_no_grad_trunc_normal_", "trunc_normal_"}), ns)
    ns["to_2tuple"] = ns["_ntuple"](2)
    exec(extract(os.path.join(REF, "htsat.py"), {"window_partition", "window_reverse", "WindowAttention", "SwinTransformerBlock"}), ns)
    return ns["SwinTransformerBlock"]


def main():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bb", os.path.join(ROOT, "dg-sct_amd", "backbone.py"))
    bb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bb)
    Ref = reference_block()
    cases = {}
    for name, (dim, res, heads, ws, shift) in {"plain": (32, 16, 4, 8, 0), "shifted": (32, 16, 4, 8, 4), "one_window": (32, 8, 4, 8, 4)}.items():
        torch.manual_seed(11)
        ref = Ref(dim, (res, res), heads, window_size=ws, shift_size=shift).eval()
        with torch.no_grad():                                   # make every parameter matter (biases / LayerNorm affine start at 0 / 1)
            for p in ref.parameters():
                p.add_(0.1 * torch.randn_like(p))
        mine = bb.HTSATBlock(dim, (res, res), heads, window_size=ws, shift_size=shift).eval()
        sd = ref.state_dict()
        assert set(sd) == set(mine.state_dict()), (sorted(set(sd) ^ set(mine.state_dict())))
        mine.load_state_dict(sd)
        g = torch.Generator().manual_seed(3)
        x = torch.randn(3, res * res, dim, generator=g, requires_grad=True)
        cot = torch.randn(3, res * res, dim, generator=g)
        y, attn = ref(x)
        y.backward(cot)
        x2 = x.detach().clone().requires_grad_(True)
        y2, attn2 = mine(x2)
        y2.backward(cot)
        err = max((y - y2).abs().max().item(), (attn - attn2).abs().max().item(), (x.grad - x2.grad).abs().max().item())
        assert err < 1e-5, (name, err)
        cases[name] = dict(cfg=(dim, res, heads, ws, shift), state=sd, x=x.detach(), cot=cot, y=y.detach(), dx=x.grad.clone())
        print(name, "max |reference - restatement|", err)
    out = os.path.join(ROOT, "tests", "golden", "htsat_block.pt")
    torch.save(cases, out)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
