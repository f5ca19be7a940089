"""fixed cost vs k-loop cost of the tiled engine: time(M, N, K) over K at the late-stage shapes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench as gb
for (M, N) in ((23040, 512), (40960, 384), (655360, 96), (163840, 192)):
    row = []
    for K in (32, 64, 128, 256, 512):
        if K > 4 * N and N < 128: continue
        us = gb.run((M, N, K, 1, 1, 1, 1, 0, 0, 1, 0), iters=10)
        row.append(f"K={K}: {us:6.1f} us")
    print(f"M={M} N={N}  " + "  ".join(row))
