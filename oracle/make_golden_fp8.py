"""TEST INFRASTRUCTURE: expected outputs of the fp8 (e4m3-projection) adapter path for tests/test_fp8.py, produced by the host
emulation of the schedule (tests/emu: the real plan.cpp against host loops with an e4m3 model of the quantisation, fp64
accumulation) -> tests/golden/fp8_emu_ave_64x96.pt.  There is no reference arithmetic for fp8 (the reference is fp32
throughout); this fixture pins the HIP kernels to the SAME quantisation points without loading the emulation library in the
`-m gpu` run.  Run here (CPU only):  python oracle/make_golden_fp8.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

from build_emu import build_emu  # noqa: E402
from dgsct_amd._lib import Lib  # noqa: E402
import test_fp8 as T  # noqa: E402


def main():
    emu = Lib(build_emu())
    r = T._case(emu, torch.device("cpu"), *T.FP8_FIXTURE_SHAPE, flavour="ave")
    torch.save({"shape": T.FP8_FIXTURE_SHAPE, "out": r["out"].float(), "map": r["map"].float()},
               os.path.join(ROOT, "tests", "golden", "fp8_emu_ave_64x96.pt"))
    print("wrote fp8_emu_ave_64x96.pt", tuple(r["out"].shape))


if __name__ == "__main__":
    main()
