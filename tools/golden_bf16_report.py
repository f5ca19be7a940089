import sys, torch
sys.path.insert(0,'tests')
from helpers import *
from dgsct_amd._lib import default_lib
DEV=torch.device('cuda:0')
def l2(a,b):
    a,b=a.float().cpu(),b.float().cpu(); return ((a-b).norm()/b.norm().clamp_min(1e-20)).item()
for name in golden_names():
    fx=load_golden(name)
    r=run_library(default_lib(), fx, DEV, torch.bfloat16, training=True)
    print(f"{name:20s} out {nrm_err(r['out'],fx['out']):.4f} map {nrm_err(r['map'],fx['map']):.4f} dX {nrm_err(r['dX'],fx['dX']):.4f}/{l2(r['dX'],fx['dX']):.4f} dY {nrm_err(r['dY'],fx['dY']):.4f}/{l2(r['dY'],fx['dY']):.4f}")
