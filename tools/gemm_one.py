import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import gemm_bench as gb
for shape in [(4096, 2304, 96, 160, 1, 1, 0, 0, 1, 0, 0), (4096, 2304, 96, 160, 1, 1, 0, 0, 0, 0, 0), (4096, 2304, 15360, 1, 1, 1, 1, 0, 0, 1, 0), (4096, 4096, 4096, 1, 1, 1, 1, 0, 0, 1, 0), (4096, 4096, 4096, 1, 1, 1, 0, 0, 0, 1, 0), (8192, 8192, 8192, 1, 1, 1, 1, 0, 0, 1, 0)]:
    us = gb.run(shape, iters=5)
    M, N, K, KB = shape[:4]
    print(shape[:7], "atomic", shape[8], "%.1f us  %.0f TF/s" % (us, 2.0 * M * N * K * KB / us / 1e6))
