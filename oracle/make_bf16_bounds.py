"""TEST INFRASTRUCTURE: emulator-derived per-case error bounds for the bf16 GPU tests -> tests/golden/bf16_bounds.json.

Why bounds and not one tolerance (tools/bf16_sensitivity.py): against an fp32 oracle the error of ANY
evaluation with bf16-rounded operands is not proportional to the rounding -- (a) the adapter's bottleneck ReLU turns a
relative perturbation eps of its input into a FRACTION ~eps of flipped mask bits, i.e. a gradient error ~sqrt(eps); (b) the
latent-token softmaxes are un-scaled (logits ~ sqrt(C)), so rounding of Yp / tok is amplified before it reaches that ReLU.
On the tiny golden problems (4-16 bottleneck channels) a single flipped unit moves a gradient by 1-25 %, and which units
flip changes with every change of rounding order.  So the honest statement for a bf16 result is: it lies inside the range
that an IDEAL bf16-storage evaluation (same schedule, every stored tensor rounded to bf16, products accumulated in fp64:
tests/emu in bf16 mode) produces when its inputs move by one bf16 ulp.

For each case: K runs of the host emulation on inputs perturbed by a random relative 2^-9 (re-rounded to bf16), each
compared with the fp32 oracle ON THE SAME INPUTS; bound = SAFETY x max over the runs (floored).  Metric: relative L2.
Run here (CPU only; the emulation needs no GPU):  python oracle/make_bf16_bounds.py
"""
from __future__ import annotations

import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

from build_emu import build_emu  # noqa: E402
from helpers import golden_names, load_golden, param_table, spec_of  # noqa: E402
from dgsct_amd import ops  # noqa: E402
from dgsct_amd._lib import PARAM_NAMES, Lib  # noqa: E402
from oracle import dgsct_oracle as O  # noqa: E402

K, SAFETY = 4, 2.0
FLOOR = dict(out=1e-2, map=2e-3, dX=2e-2, dY=2e-2, grads=3e-2)
REAL = {"real_144x512": (144, 512, 256, 384), "real_256x384": (256, 384, 144, 512), "real_36x1024": (36, 1024, 64, 768),
        "real_64x768": (64, 768, 36, 1024)}


def l2(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def perturb(t, gen):
    return (t * (1 + 2.0 ** -9 * (2 * torch.rand(t.shape, generator=gen) - 1))).bfloat16().float()


def one_run(emu, cfg, state, X, Y, dOut, dMap, dTmap):
    po = {k: v.clone() for k, v in state.items()}
    out_o, map_o, _, s = O.forward(po, X, Y, cfg, training=True)
    dX_o, dY_o, g_o = O.backward(po, s, cfg, dOut, dMap, dTmap, training=True)
    spec = spec_of(cfg)
    dev = torch.device("cpu")
    params = param_table(state, spec, dev)
    dt = torch.bfloat16
    prep = ops.prepare(emu, spec, params, dt, dev)
    out, amap, _, saved, d = ops.raw_forward(emu, spec, params, prep, X.to(dt).contiguous(), Y.to(dt).contiguous(), True)
    dX, dY, grads = ops.raw_backward(emu, spec, d, params, prep, X.to(dt).contiguous(), Y.to(dt).contiguous(), saved,
                                     dOut.to(dt).contiguous(), dMap, dTmap)
    e = dict(out=l2(out, out_o), map=l2(amap, map_o), dX=l2(dX, dX_o), dY=l2(dY, dY_o), grads={})
    for i, g in enumerate(grads):
        n = PARAM_NAMES[i]
        if g is not None and n in g_o and n != "ln_before.bias":
            e["grads"][n] = l2(g, g_o[n].reshape(-1))
    return e


def bound_case(emu, cfg, state, X, Y, dOut, dMap, dTmap):
    gen = torch.Generator().manual_seed(1234)
    runs = []
    for k in range(K):
        Xk, Yk = (X, Y) if k == 0 else (perturb(X, gen), perturb(Y, gen))
        runs.append(one_run(emu, cfg, state, Xk.bfloat16().float(), Yk.bfloat16().float(), dOut.bfloat16().float(), dMap, dTmap))
    b = {k: max(FLOOR[k], SAFETY * max(r[k] for r in runs)) for k in ("out", "map", "dX", "dY")}
    b["grads"] = {n: max(FLOOR["grads"], SAFETY * max(r["grads"][n] for r in runs)) for n in runs[0]["grads"]}
    b["emu_runs"] = [{k: round(r[k], 5) for k in ("out", "map", "dX", "dY")} for r in runs]
    return b


# stage-2 / stage-3 shapes of tests/test_configs_gpu.py (Swin-V2-L widths; AVS-S4 and AVQA flavours): (flavour, shape, overrides)
CONFIGS = {
    "cfg_ave_144x768": ("ave", (144, 768, 256, 384), {}), "cfg_ave_256x384": ("ave", (256, 384, 144, 768), {}),
    "cfg_ave_36x1536": ("ave", (36, 1536, 64, 768), {}), "cfg_ave_64x768": ("ave", (64, 768, 36, 1536), {}),
    "cfg_avs_s4_36x1536": ("avs_s4", (36, 1536, 64, 768), {}),
    "cfg_avqa_144x768": ("avqa", (144, 768, 256, 384), dict(use_gate=True)),
    "cfg_avqa_256x384": ("avqa", (256, 384, 144, 768), dict(use_gate=False)),
    "cfg_avqa_36x1536": ("avqa", (36, 1536, 64, 768), dict(use_gate=True)),
}


# num_tokens > 32 at real stage-2 / stage-3 shapes (csrc/attn_wide.cpp; tests/test_adapter_gpu.py): (N, C, No, Co, tk)
WIDE = {"wide_tk87_144x512": (144, 512, 256, 384, 87), "wide_tk40_64x768": (64, 768, 36, 1024, 40)}


def main():
    emu = Lib(build_emu())
    path = os.path.join(ROOT, "tests", "golden", "bf16_bounds.json")
    out = {"_doc": "relative-L2 bounds for the bf16 GPU tests; generated by oracle/make_bf16_bounds.py (K=%d perturbed ideal-bf16 "
                   "emulations, safety %.1f)" % (K, SAFETY)}
    keep = lambda n: True
    if len(sys.argv) > 2 and sys.argv[1] == "--only":        # add / refresh the named cases, keep the rest of the file as it is
        only = set(sys.argv[2].split(","))
        out = json.load(open(path))
        keep = lambda n: n in only
    for name in golden_names():
        if not keep(name):
            continue
        fx = load_golden(name)
        cfg = O.AdapterConfig(**fx["cfg"])
        state = {k: v.clone() for k, v in fx["state0"].items()}
        if cfg.remap == "bicubic":
            state["_bicubic"] = O.bicubic_matrix(cfg.No, cfg.N)
        out[name] = bound_case(emu, cfg, state, fx["X"], fx["Y"], fx["dOut"], fx["dMap"], fx["dTmap"])
        print(name, {k: round(out[name][k], 4) for k in ("out", "map", "dX", "dY")}, flush=True)
    for name, (N, C, No, Co) in REAL.items():
        if not keep(name):
            continue
        cfg = O.AdapterConfig(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2)
        p = O.random_params(cfg, "ave", seed=0, scale=0.577)
        gen = torch.Generator().manual_seed(1)
        X, Y = torch.randn(10, N, C, generator=gen), torch.randn(10, No, Co, generator=gen)
        dOut, dMap = torch.randn(10, N, C, generator=gen), torch.randn(10, N, generator=gen)
        out[name] = bound_case(emu, cfg, p, X, Y, dOut, dMap, None)
        print(name, {k: round(out[name][k], 4) for k in ("out", "map", "dX", "dY")}, flush=True)
    for name, (N, C, No, Co, tk) in WIDE.items():
        if not keep(name):
            continue
        cfg = O.AdapterConfig(N=N, C=C, No=No, Co=Co, tk=tk, r=8, g=2)
        p = O.random_params(cfg, "ave", seed=0, scale=0.577)
        gen = torch.Generator().manual_seed(1)
        X, Y = torch.randn(10, N, C, generator=gen), torch.randn(10, No, Co, generator=gen)
        dOut, dMap = torch.randn(10, N, C, generator=gen), torch.randn(10, N, generator=gen)
        out[name] = bound_case(emu, cfg, p, X, Y, dOut, dMap, None)
        print(name, {k: round(out[name][k], 4) for k in ("out", "map", "dX", "dY")}, flush=True)
    for name, (flavour, (N, C, No, Co), over) in CONFIGS.items():
        if not keep(name):
            continue
        cfg = O.AdapterConfig(**{**dict(N=N, C=C, No=No, Co=Co, tk=32, r=8, g=2), **O.FLAVOURS[flavour], **over})
        p = O.random_params(cfg, flavour, seed=0, scale=0.577)
        gen = torch.Generator().manual_seed(1)
        X, Y = torch.randn(10, N, C, generator=gen), torch.randn(10, No, Co, generator=gen)
        dOut, dMap = torch.randn(10, N, C, generator=gen), torch.randn(10, N, generator=gen)
        out[name] = bound_case(emu, cfg, p, X, Y, dOut, dMap, None)
        print(name, {k: round(out[name][k], 4) for k in ("out", "map", "dX", "dY")}, flush=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
