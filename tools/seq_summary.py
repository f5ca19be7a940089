"""Roll the per-shape kernel sequences (tools/seq_stack.sh) up to one bench step: family x stage table (ms per step)."""
import glob
import os
import re
import sys
from collections import defaultdict

MULT = {2304: 4, 4096: 4, 576: 4, 1024: 4, 144: 12, 256: 12, 36: 4, 64: 4}
STAGE = {2304: 0, 4096: 0, 576: 1, 1024: 1, 144: 2, 256: 2, 36: 3, 64: 3}
tab = defaultdict(lambda: [0.0] * 4)
cnt = defaultdict(lambda: [0] * 4)
for f in glob.glob(os.path.join(sys.argv[1], "seq_*.txt")):
    N = int(os.path.basename(f).split("_")[1])
    for line in open(f):
        m = re.match(r"\s*([\d.]+) us\s+grid=\(([^)]*)\)\s+(\S.*)", line)
        if not m:
            continue
        us, grid, name = float(m.group(1)), m.group(2), m.group(3)
        fam = re.sub(r"<.*", "", name)
        if fam == "gemm8_kernel":
            fam = "gemm8 (deep products)"
        if fam == "gemm_kernel":
            g = [int(x) for x in grid.split(",")]
            wgs = g[0] * g[1] * g[2]
            fam = "gemm wgs<64" if wgs < 64 else ("gemm wgs<256" if wgs < 256 else ("gemm wgs<1024" if wgs < 1024 else "gemm wgs>=1024"))
        tab[fam][STAGE[N]] += us * MULT[N] / 1e3
        cnt[fam][STAGE[N]] += MULT[N]
tot = [sum(v[i] for v in tab.values()) for i in range(4)]
print(f"{'family':28s} {'st0':>8} {'st1':>8} {'st2':>8} {'st3':>8} {'total':>8}   launches")
for k, v in sorted(tab.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:28s} " + " ".join(f"{x:8.2f}" for x in v) + f" {sum(v):8.2f}   {sum(cnt[k])}")
print(f"{'TOTAL':28s} " + " ".join(f"{x:8.2f}" for x in tot) + f" {sum(tot):8.2f}")
