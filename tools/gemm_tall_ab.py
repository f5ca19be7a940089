"""gemm_tall.hip against the tiled engine's split-K on the weight-gradient shapes of stages 0-1 (alone on the chip).
usage: python tools/gemm_tall_ab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgsct_amd
from dgsct_amd._lib import GemmArgs, default_lib
DEV = "cuda:0"
lib = default_lib()
# (M, N, rows, lda, ldb, batch, a_bs, b_bs, what)
SHAPES = [(96, 96, 655360, 96, 96, 1, 0, 0, "dWv1 audio st0"), (48, 96, 655360, 48, 96, 1, 0, 0, "dWv2 audio st0"),
          (6, 48, 655360, 12, 96, 2, 6, 48, "dWd audio st0 (g=2)"), (48, 6, 655360, 96, 12, 2, 48, 6, "dWu audio st0 (g=2)"),
          (96, 128, 655360, 96, 128, 1, 0, 0, "dWc audio st0"), (128, 128, 368640, 128, 128, 1, 0, 0, "dWv1 visual st0"),
          (64, 128, 368640, 64, 128, 1, 0, 0, "dWv2 visual st0"), (8, 64, 368640, 16, 128, 2, 8, 64, "dWd visual st0"),
          (96, 192, 163840, 96, 192, 1, 0, 0, "dWv2 audio st1"), (12, 96, 163840, 24, 192, 2, 12, 96, "dWd audio st1"),
          (128, 256, 92160, 128, 256, 1, 0, 0, "dWv2 visual st1"), (16, 128, 92160, 32, 256, 2, 16, 128, "dWd visual st1")]
def run(shape, iters=10):
    M, N, K, lda, ldb, batch, abs_, bbs, _ = shape
    A = torch.randn(K * lda, device=DEV).bfloat16(); B = torch.randn(K * ldb, device=DEV).bfloat16()
    D = torch.zeros(batch * M * N, device=DEV)
    a = GemmArgs()
    a.mode, a.M, a.N, a.K, a.KB, a.batch, a.splitk, a.atomic = 1, M, N, K, 1, batch, 0, 1
    a.A, a.lda, a.a_kmajor, a.a_bs, a.a_kbs = A.data_ptr(), lda, 0, abs_, 0
    a.B, a.ldb, a.b_kmajor, a.b_bs, a.b_kbs = B.data_ptr(), ldb, 0, bbs, 0
    a.D, a.ddt, a.ldd, a.dbs = D.data_ptr(), 0, N, M * N
    a.alpha, a.beta = 1.0, 1.0
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): lib.test_gemm(a, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): lib.test_gemm(a, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, D
print(f"{'shape':28s} {'tiled us':>9s} {'tall us':>9s} {'GB/s tall':>10s}  max |tall - tiled| / max|tiled|")
for s in SHAPES:
    lib.test_tune("gemmtall", 0); t0, D0 = run(s)
    lib.test_tune("gemmtall", 1); t1, D1 = run(s)
    gb = (s[3] + s[4]) * s[2] * 2 / t1 / 1e3
    print(f"{s[8]:28s} {t0:9.1f} {t1:9.1f} {gb:10.0f}  {float((D1 - D0).abs().max() / D0.abs().max()):.2e}")
