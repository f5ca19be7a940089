"""GPU diagnostic (not a pytest file): run one golden case through libdgsct.so and print, per saved
intermediate / output / gradient, the error against the oracle.  `python tools/diag_gpu.py [case] [bf16]`."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch

from helpers import load_golden, oracle_cfg, param_table, spec_of
from dgsct_amd import ops
from dgsct_amd._lib import PARAM_NAMES, default_lib
from oracle import dgsct_oracle as O


def synth(N, C, No, Co, tk, BT, seed=3):
    import dataclasses
    cfg = O.AdapterConfig(N=N, C=C, No=No, Co=Co, tk=tk, r=8, g=2)
    p = O.random_params(cfg, "ave", seed=seed)
    gen = torch.Generator().manual_seed(seed + 1)
    r = lambda *s: torch.randn(*s, generator=gen).bfloat16().float()
    return dict(cfg=dataclasses.asdict(cfg), state0=p, X=r(BT, N, C), Y=r(BT, No, Co), dOut=r(BT, N, C),
                dMap=torch.randn(BT, N, generator=gen), dTmap=None)


def main(name="ave_orderA", dtype=torch.float32):
    lib = default_lib()
    if name.startswith("synth:"):
        fx = synth(*[int(x) for x in name[6:].split(",")])
    else:
        fx = load_golden(name)
    dev = torch.device("cuda:0")
    spec = spec_of(fx["cfg"])
    cfg = oracle_cfg(fx["cfg"])
    state = {k: v.clone() for k, v in fx["state0"].items()}
    if spec.remap == "bicubic":
        state["_bicubic"] = O.bicubic_matrix(spec.No, spec.N)
    p_or = {k: v.clone() for k, v in state.items()}
    out_o, map_o, tmap_o, s = O.forward(p_or, fx["X"], fx["Y"], cfg, training=True)
    dX_o, dY_o, g_o = O.backward(p_or, s, cfg, fx["dOut"], fx["dMap"], fx["dTmap"], training=True)
    params = param_table(state, spec, dev)
    X = fx["X"].to(dev, dtype).contiguous(); Y = fx["Y"].to(dev, dtype).contiguous()
    prep = ops.prepare(lib, spec, params, dtype, dev)
    old = [(k, lib.test_tune(k, 2)) for k in ("gatefuse", "vq1fuse")]     # 2: the fused passes also materialise vq2 / vq1 for this dump
    try:
        out, amap, tmap, saved, d = ops.raw_forward(lib, spec, params, prep, X, Y, True)
        torch.cuda.synchronize()
    finally:
        for k, v in old:
            lib.test_tune(k, v)
    regs = lib.saved_regions(d)
    B, N, C, No, Co, tk = X.shape[0], spec.N, spec.C, spec.No, spec.Co, spec.tk
    Np, Nop, tkp = (N + 7) // 8 * 8, (No + 7) // 8 * 8, (tk + 7) // 8 * 8
    dd, ds = C // 2, C // spec.r
    E = dtype

    def view(nm, dt, shape):
        if nm not in regs:                 # region not part of this shape's layout (P1 / P2 since round 2, Xc for the fused gate passes)
            return None
        off, nb = regs[nm]
        n = 1
        for x in shape:
            n *= x
        return saved[off:off + n * torch.empty(0, dtype=dt).element_size()].view(dt).view(*shape).float().cpu()

    def show(tag, got, ref):
        if got is None:
            print(f"  {tag:10s} (not stored by this schedule)")
            return
        ref = ref.float()
        e = (got - ref).abs().max().item()
        print(f"  {tag:10s} max|err| {e:.3e}   max|ref| {ref.abs().max().item():.3e}   {'<<<<< BAD' if not e < 2e-2 * max(1, ref.abs().max().item()) else ''}")

    print(f"== {name} dtype={dtype} order={s['order']}")
    show("Yp", view("Yp", E, (B, N, C)), s["Yp"])
    if s["order"] == "A":
        show("T1", view("T", E, (B, N, Co)), s["T1"])
    else:
        show("T2t", view("T", E, (B, C, Nop))[..., :No], s["T2t"])
    show("tok", view("tok", torch.float32, (B, tk, C)), s["tok"])
    show("a", view("a", torch.float32, (B, C)), s["a"])
    show("X1", view("X1", E, (B, N, C)), s["X1"])
    show("aq1", view("aq1", E, (B, C)), s["aq1"])
    show("aq2", view("aq2", E, (B, dd)), s["aq2"])
    show("vq1", view("vq1", E, (B, N, C)), s["vq1"])
    show("mvq1", view("mvq1", torch.float32, (B, C)), s["mvq1"])
    show("q", view("q", E, (B, dd)), s["q"])
    show("ch", view("ch", torch.float32, (B, C)), s["ch"])
    show("Xc", view("Xc", E, (B, N, C)), s["Xc"])
    show("vq2", view("vq2", E, (B, N, dd)), s["vq2"])
    show("sl", view("sl", torch.float32, (B, N)), s["sl"])
    show("sg", view("sg", torch.float32, (B, N)), s["sg"])
    show("X3", view("X3", E, (B, N, C)), s["X3"])
    show("Zp", view("Zp", E, (B, N, ds)), s["Zp"])
    show("Z", view("Z", E, (B, N, ds)), s["Z"])
    show("Op", view("Op", E, (B, N, C)), s["Op"])
    show("out", out.float().cpu(), out_o)
    show("map", amap.cpu(), map_o)
    dOut = fx["dOut"].to(dev, dtype).contiguous()
    dX, dY, grads = ops.raw_backward(lib, spec, d, params, prep, X, Y, saved, dOut, fx["dMap"].to(dev),
                                     fx["dTmap"].to(dev) if fx["dTmap"] is not None else None)
    torch.cuda.synchronize()
    show("dX", dX.float().cpu(), dX_o)
    show("dY", dY.float().cpu(), dY_o)
    for i, g in enumerate(grads):
        if g is not None and PARAM_NAMES[i] in g_o:
            show("d" + PARAM_NAMES[i][:22], g.cpu(), g_o[PARAM_NAMES[i]].reshape(-1))


if __name__ == "__main__":
    nm = sys.argv[1] if len(sys.argv) > 1 else "ave_orderA"
    dt = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
    main(nm, dt)
