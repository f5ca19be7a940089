"""Build libdgsct.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output so the
binary travels with the source snapshot: dg-sct_amd/libdgsct.so."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdgsct.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
SOURCES = ["gemm.hip", "gemm_fx.hip", "gemm8.hip", "gemm_skinny.hip", "gemm_tall.hip", "gemm_wgbt.hip", "gemm_fp8.hip", "attn.hip", "attn2.hip", "temporal.hip", "wattn.hip", "prims_hip.hip", "prims_strip.hip", "prims_proj.hip", "fused_gate.hip", "plan.cpp", "attn_wide.cpp", "capi.cpp", "err.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]


def _newer(src, dst):
    return not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
              [os.path.join(os.path.dirname(HERE), "include", "dgsct.h")]
    hdr_time = max(os.path.getmtime(h) for h in headers)
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s + ".o")
        if force or _newer(src, obj) or os.path.getmtime(obj) < hdr_time:
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s + ".o") for s in SOURCES]
    if force or jobs or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
