"""TEST INFRASTRUCTURE: build tests/emu/libdgsct_emu.so = csrc/plan.cpp + csrc/capi.cpp (the real
kernel schedule and C ABI) linked against tests/emu/prims_host.cpp (host loops instead of gfx950
kernels).  g++ only.  Never loaded by the product package."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "dg-sct_amd", "csrc")
OUT = os.path.join(HERE, "libdgsct_emu.so")


def build_emu() -> str:
    srcs = [os.path.join(CSRC, f) for f in ("plan.cpp", "attn_wide.cpp", "capi.cpp", "err.cpp")] + [os.path.join(HERE, "prims_host.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "dgsct.h")]
    if os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", OUT] + srcs, check=True)
    return OUT


if __name__ == "__main__":
    print(build_emu())
