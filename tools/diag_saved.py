"""Compare the named regions of the `saved` buffer (forward intermediates) between the GPU library and the host emulation
(tests/emu) for one golden case in bf16: finds the first kernel whose result leaves the ideal-bf16 trajectory.
usage (GPU box): python tools/diag_saved.py <golden-name> [bf16|fp32]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
from helpers import load_golden, run_library
from build_emu import build_emu
from dgsct_amd._lib import Lib, default_lib
name = sys.argv[1]; dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "fp32") else torch.bfloat16
fx = load_golden(name)
emu = Lib(build_emu())
re_ = run_library(emu, fx, torch.device("cpu"), dt, training=True)
rg = run_library(default_lib(), fx, torch.device("cuda:0"), dt, training=True)
torch.cuda.synchronize()
regs = emu.saved_regions(re_["desc"])
se, sg = re_["saved"], rg["saved"].cpu()
cfg = fx["cfg"]; es = 2 if dt == torch.bfloat16 else 4
F32 = {"a", "mvq1", "bnacc1", "bnacc2", "ch", "sl", "sg", "map", "tg", "mu_b", "rstd_b", "bn1", "bn2", "mu_p", "rstd_p", "tok", "lse"}
for n, (off, nb) in regs.items():
    if n == "tokpk": continue
    a, b = se[off:off + nb], sg[off:off + nb]
    if n in F32: a, b = a.view(torch.float32), b.view(torch.float32)
    else: a, b = (a.view(dt).float(), b.view(dt).float())
    d = (a - b).abs().max().item(); m = a.abs().max().item()
    print(f"{n:8s} bytes {nb:8d} max|emu| {m:10.4g} max|diff| {d:10.4g} rel {d / max(m, 1e-20):.3g}")
for k in ("out", "dX", "dY"):
    a, b = re_[k].float(), rg[k].float().cpu()
    print(k, "emu-vs-gpu rel-L2", ((a - b).norm() / a.norm()).item(), " emu-vs-ref", ((a - fx[k]).norm() / fx[k].norm()).item(), " gpu-vs-ref", ((b - fx[k]).norm() / fx[k].norm()).item())
