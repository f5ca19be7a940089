"""TEST INFRASTRUCTURE: golden vectors of the reference ``TemporalAttention`` -> tests/golden/temporal.pt.

Runs HERE (where /root/reference exists): the reference classes are taken from the reference sources with ``ast`` (the
classes only; no reference text is stored in this repo), run on CPU in eval mode (Dropout off) and compared with the
functional restatement oracle/temporal_oracle.py (asserted <= 1e-5).  The module has ~12 M parameters (48 MB), too large
to commit: the fixture stores the SEED under which ``TemporalAttention()`` was constructed plus a checksum of every
parameter -- the drop-in module builds its sub-modules in the reference's order, so the same seed reproduces the same
parameters (asserted here and again in the tests) -- together with inputs, cotangents, outputs, input gradients and the
norm / sum of every parameter gradient.
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
    out = []
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in names:
            out.append(ast.get_source_segment(src, node))
    return "\n\n".join(out)


def reference_class():
    import copy
    import torch.nn as nn
    import torch.nn.functional as F
    from torch.nn import Dropout, LayerNorm, Linear, Module, ModuleList, MultiheadAttention
    ns = dict(torch=torch, nn=nn, F=F, copy=copy, Dropout=Dropout, LayerNorm=LayerNorm, Linear=Linear, Module=Module,
              ModuleList=ModuleList, MultiheadAttention=MultiheadAttention)
    exec(extract(os.path.join(REF, "models.py"), {"Encoder", "Decoder", "EncoderLayer", "DecoderLayer", "_get_clones",
                                                   "_get_activation_fn"}), ns)
    exec(extract(os.path.join(REF, "net_trans.py"), {"RNNEncoder", "InternalTemporalRelationModule",
                                                      "CrossModalRelationAttModule", "TemporalAttention"}), ns)
    return ns["TemporalAttention"]


def main():
    from oracle import temporal_oracle as TO
    import dgsct_amd  # noqa: F401
    from dgsct_amd.temporal import TemporalAttention
    SEED, B, T = 1234, 3, 10
    torch.manual_seed(SEED)
    ref = reference_class()().eval()
    torch.manual_seed(SEED)
    mine = TemporalAttention().eval()
    sd = ref.state_dict()
    assert list(sd) == list(mine.state_dict()), "state_dict keys / order differ from the reference"
    for k, v in mine.state_dict().items():
        assert torch.equal(v, sd[k]), f"seeded construction differs from the reference at {k}"
    g = torch.Generator().manual_seed(7)
    fv = torch.randn(B, T, 1536, generator=g, requires_grad=True)
    fa = torch.randn(B, T, 768, generator=g, requires_grad=True)
    cv, ca, cg = torch.randn(T, B, 256, generator=g), torch.randn(T, B, 256, generator=g), torch.randn(T, B, 1, generator=g)
    ov, oa, og = ref(fv, fa)
    torch.autograd.backward([ov, oa, og], [cv, ca, cg])
    pg = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    # the functional restatement against the reference
    fv2, fa2 = fv.detach().clone().requires_grad_(True), fa.detach().clone().requires_grad_(True)
    sd2 = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    o2 = TO.forward(sd2, fv2, fa2)
    torch.autograd.backward(list(o2), [cv, ca, cg])
    for a, b in zip(o2, (ov, oa, og)):
        assert (a - b).abs().max() < 1e-5
    assert (fv2.grad - fv.grad).abs().max() < 1e-5 and (fa2.grad - fa.grad).abs().max() < 1e-5
    for k, gr in pg.items():
        assert (sd2[k].grad - gr).abs().max() < 1e-4 * max(1.0, gr.abs().max().item()), k
    never = sorted(k for k, p in ref.named_parameters() if p.grad is None)
    fx = dict(seed=SEED, B=B, T=T, fv=fv.detach(), fa=fa.detach(), cv=cv, ca=ca, cg=cg, out_v=ov.detach(), out_a=oa.detach(),
              gate=og.detach(), d_fv=fv.grad.clone(), d_fa=fa.grad.clone(), keys=list(sd),
              param_sum={k: float(v.double().sum()) for k, v in sd.items()},
              grad_norm={k: float(v.double().norm()) for k, v in pg.items()}, grad_sum={k: float(v.double().sum()) for k, v in pg.items()},
              no_grad=never)
    torch.save(fx, os.path.join(ROOT, "tests", "golden", "temporal.pt"))
    print("temporal.pt written:", len(sd), "tensors,", sum(v.numel() for v in sd.values()), "parameters;", len(never), "never get a gradient")


def _ns():
    import copy
    import torch.nn as nn
    import torch.nn.functional as F
    from torch.nn import Dropout, LayerNorm, Linear, Module, ModuleList, MultiheadAttention
    ns = dict(torch=torch, nn=nn, F=F, copy=copy, Dropout=Dropout, LayerNorm=LayerNorm, Linear=Linear, Module=Module,
              ModuleList=ModuleList, MultiheadAttention=MultiheadAttention)
    exec(extract(os.path.join(REF, "models.py"), {"Encoder", "Decoder", "EncoderLayer", "DecoderLayer", "_get_clones",
                                                   "_get_activation_fn"}), ns)
    ns["CMBS_Encoder"] = ns["Encoder"]                 # AVVP/nets/mgn.py:28 imports it under this name
    return ns


def _variant(ref_cls, mine_cls, inputs, out_name, flatten):
    """seeded construction must reproduce the reference's parameters; outputs / input gradients / gradient norms are stored"""
    SEED = 4321
    torch.manual_seed(SEED)
    ref = ref_cls().eval()
    torch.manual_seed(SEED)
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    from build_emu import build_emu                      # the host-emulated kernel library (test infrastructure)
    from dgsct_amd._lib import Lib
    mine = mine_cls(lib=Lib(build_emu())).eval()
    sd = ref.state_dict()
    assert list(sd) == list(mine.state_dict()), "state_dict keys / order differ from the reference"
    for k, v in mine.state_dict().items():
        assert torch.equal(v, sd[k]), f"seeded construction differs from the reference at {k}"
    ins = [t.clone().requires_grad_(True) for t in flatten(inputs)]
    outs = flatten(ref(*_pack(inputs, ins)))
    g = torch.Generator().manual_seed(11)
    cots = [torch.randn(o.shape, generator=g) for o in outs]
    torch.autograd.backward(outs, cots)
    pg = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    # the drop-in (host-emulated gate kernel) against the reference, here, before anything is written
    ins2 = [t.detach().clone().requires_grad_(True) for t in ins]
    outs2 = flatten(mine(*_pack(inputs, ins2)))
    torch.autograd.backward(outs2, cots)
    for a, b in zip(outs2, outs):
        assert (a - b).abs().max() < 2e-5, (out_name, float((a - b).abs().max()))
    for a, b in zip(ins2, ins):
        assert (a.grad - b.grad).abs().max() < 2e-5 * max(1.0, float(b.grad.abs().max())), out_name
    never = sorted(k for k, p in ref.named_parameters() if p.grad is None)
    assert never == sorted(k for k, p in mine.named_parameters() if p.grad is None), "different parameters go without a gradient"
    fx = dict(seed=SEED, inputs=[t.detach() for t in ins], cots=cots, outs=[o.detach() for o in outs], d_inputs=[t.grad.clone() for t in ins],
              keys=list(sd), param_sum={k: float(v.double().sum()) for k, v in sd.items()},
              grad_norm={k: float(v.double().norm()) for k, v in pg.items()}, grad_sum={k: float(v.double().sum()) for k, v in pg.items()},
              no_grad=never)
    torch.save(fx, os.path.join(ROOT, "tests", "golden", out_name))
    print(out_name, "written:", len(sd), "tensors,", sum(v.numel() for v in sd.values()), "parameters;", len(never), "never get a gradient")


def _pack(template, flat):
    """re-nest `flat` like `template` (a tuple whose first element may be a list of tensors)"""
    out, i = [], 0
    for t in template:
        if isinstance(t, (list, tuple)):
            out.append(list(flat[i:i + len(t)])); i += len(t)
        else:
            out.append(flat[i]); i += 1
    return out


def _flat(x):
    out = []
    for t in (x if isinstance(x, (list, tuple)) else [x]):
        out += list(t) if isinstance(t, (list, tuple)) else [t]
    return out


def main_variants():
    """the AVVP and AVS copies of the class: DG-SCT/AVVP/nets/mgn.py:107-159,
    DG-SCT/AVS/avs_scripts/avs_s4/model/PVT_AVSModel.py:447-582"""
    import dgsct_amd  # noqa: F401
    from dgsct_amd.temporal import TemporalAttentionAVS, TemporalAttentionAVVP
    g = torch.Generator().manual_seed(5)
    ns = _ns()
    exec(extract("/root/reference/DG-SCT/AVVP/nets/mgn.py", {"RNNEncoder", "InternalTemporalRelationModule",
                                                             "CrossModalRelationAttModule", "TemporalAttention"}), ns)
    B = 3
    _variant(ns["TemporalAttention"], TemporalAttentionAVVP,
             (torch.randn(B, 10, 128, generator=g), torch.randn(B, 10, 128, generator=g)), "temporal_avvp.pt", _flat)
    ns = _ns()
    exec(extract("/root/reference/DG-SCT/AVS/avs_scripts/avs_s4/model/PVT_AVSModel.py",
                 {"RNNEncoder", "InternalTemporalRelationModule", "CrossModalRelationAttModule", "TemporalAttention"}), ns)
    B = 2
    maps = [torch.randn(B * 5, 256, hw, hw, generator=g) for hw in (4, 3, 2, 1)]      # (56, 28, 14, 7 in the model: same code path)
    _variant(ns["TemporalAttention"], TemporalAttentionAVS, (maps, torch.randn(B, 5, 128, generator=g)), "temporal_avs.pt", _flat)


if __name__ == "__main__":
    main()
    main_variants()
