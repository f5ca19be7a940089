"""Reference points for the GEMM engine: the dWn shape with / without atomics, and plain square bf16 GEMMs.
usage: [DGSCT_GEMM_GLDS=1] python tools/gemm_one.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_bench as gb  # noqa: E402

# (M, N, K, KB, batch, ak, bk, a_shared, atomic, out_bf16, residual)
SHAPES = [(4096, 2304, 96, 160, 1, 1, 0, 0, 1, 0, 0), (2304, 4096, 96, 160, 1, 1, 1, 0, 1, 0, 0), (4096, 2304, 96, 160, 1, 1, 0, 0, 0, 0, 0),
          (4096, 2304, 15360, 1, 1, 1, 1, 0, 0, 1, 0), (4096, 4096, 4096, 1, 1, 1, 1, 0, 0, 1, 0),
          (4096, 4096, 4096, 1, 1, 1, 0, 0, 0, 1, 0), (8192, 8192, 8192, 1, 1, 1, 1, 0, 0, 1, 0)]
for shape in SHAPES:
    us = gb.run(shape, iters=5)
    M, N, K, KB = shape[:4]
    print(shape[:7], "atomic", shape[8], "%.1f us  %.0f TF/s" % (us, 2.0 * M * N * K * KB / us / 1e6))
