"""Time GEMM shapes of the adapter stack under each tile configuration (DGSCT_GEMM_CFG tuning hook).
usage: python tools/gemm_bench.py            (runs the built-in shape list)"""
import os, sys, time, subprocess, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgsct_amd
from dgsct_amd._lib import GemmArgs, default_lib
DEV = "cuda:0"
# (M, N, K, KB, batch, ak, bk, a_shared, atomic, out_bf16, residual)
SHAPES = [
    # --- stage 0 (M = BT*N tokens; C = 96 / 128)
    (4096, 96, 2304, 1, 160, 1, 1, 1, 0, 1, 0),      # remap fwd audio  T = Wn.T2   (A shared across the batch)
    (2304, 96, 4096, 1, 160, 1, 0, 1, 0, 1, 0),      # remap fwd visual
    (4096, 96, 2304, 1, 160, 0, 0, 1, 0, 1, 0),      # remap bwd
    (4096, 2304, 96, 160, 1, 1, 0, 0, 1, 0, 0),      # dWn audio (two-level K, atomics)
    (2304, 4096, 96, 160, 1, 1, 1, 0, 1, 0, 0),      # dWn visual
    (655360, 96, 96, 1, 1, 1, 1, 0, 0, 1, 0),        # vq1 = X1.Wv1^T
    (655360, 96, 96, 1, 1, 1, 0, 0, 0, 1, 1),        # dX1 += dvq1.Wv1
    (655360, 48, 96, 1, 1, 1, 1, 0, 0, 1, 0),        # vq2
    (368640, 128, 128, 1, 1, 1, 1, 0, 0, 1, 0),
    (368640, 128, 96, 1, 1, 1, 1, 0, 0, 1, 0),       # Yp = T1.Wc^T
    (4096, 96, 32, 1, 160, 1, 0, 0, 0, 1, 1),        # X1 = X + P2.tok
    (4096, 32, 96, 1, 160, 1, 1, 0, 0, 0, 0),        # S2 = X.tok^T (fp32 out)
    (32, 96, 4096, 1, 160, 0, 0, 0, 0, 0, 0),        # dtok = P2^T.dX1
    (96, 96, 655360, 1, 1, 0, 0, 0, 1, 0, 0),        # dWv1 stage 0
    (128, 128, 368640, 1, 1, 0, 0, 0, 1, 0, 0),
    # --- stage 1
    (163840, 192, 192, 1, 1, 1, 1, 0, 0, 1, 0),
    (92160, 256, 256, 1, 1, 1, 1, 0, 0, 1, 0),
    (256, 256, 92160, 1, 1, 0, 0, 0, 1, 0, 0),
    (1024, 192, 576, 1, 160, 1, 1, 1, 0, 1, 0),      # remap stage 1
    (576, 1024, 192, 160, 1, 1, 0, 0, 1, 0, 0),      # dWn stage 1
    # --- stage 2
    (40960, 384, 384, 1, 1, 1, 1, 0, 0, 1, 0),
    (23040, 512, 512, 1, 1, 1, 1, 0, 0, 1, 0),
    (23040, 512, 512, 1, 1, 1, 0, 0, 0, 1, 1),
    (23040, 256, 512, 1, 1, 1, 1, 0, 0, 1, 0),
    (512, 512, 23040, 1, 1, 0, 0, 0, 1, 0, 0),
    (384, 384, 40960, 1, 1, 0, 0, 0, 1, 0, 0),
    (144, 512, 256, 1, 160, 1, 0, 1, 0, 1, 0),       # remap stage 2
    (144, 256, 384, 160, 1, 1, 1, 0, 1, 0, 0),       # dWn stage 2
    (256, 384, 32, 1, 160, 1, 0, 0, 0, 1, 1),
    (32, 256, 384, 1, 160, 1, 1, 0, 0, 0, 0),
    # --- stage 3 and the B x C gate GEMMs
    (5760, 1024, 1024, 1, 1, 1, 1, 0, 0, 1, 0),
    (5760, 512, 1024, 1, 1, 1, 1, 0, 0, 1, 0),
    (1024, 1024, 5760, 1, 1, 0, 0, 0, 1, 0, 0),
    (160, 1024, 1024, 1, 1, 1, 1, 0, 0, 1, 0),
    (160, 512, 1024, 1, 1, 1, 0, 0, 0, 1, 0),
    (160, 512, 512, 1, 1, 1, 1, 0, 0, 1, 0),
    (160, 256, 512, 1, 1, 1, 1, 0, 0, 1, 0),
]

def run(shape, iters=10):
    M, N, K, KB, batch, ak, bk, ash, atomic, obf, res = shape
    lib = default_lib()
    dt = torch.bfloat16
    nbA = 1 if ash else batch
    A = torch.randn(nbA * KB * (M * K), device=DEV).to(dt)
    B = torch.randn(batch * KB * (N * K), device=DEV).to(dt)
    D = torch.zeros(batch, M, N, device=DEV, dtype=torch.bfloat16 if obf else torch.float32)
    R = torch.randn(batch, M, N, device=DEV).to(dt) if res else None
    a = GemmArgs()
    a.mode, a.M, a.N, a.K, a.KB, a.batch, a.splitk, a.atomic = 1, M, N, K, KB, batch, 0 if atomic else 1, atomic
    a.A, a.lda, a.a_kmajor = A.data_ptr(), (K if ak else M), ak
    a.a_kbs = M * K; a.a_bs = 0 if ash else KB * M * K
    a.B, a.ldb, a.b_kmajor = B.data_ptr(), (K if bk else N), bk
    a.b_kbs = N * K; a.b_bs = KB * N * K
    a.D, a.ddt, a.ldd, a.dbs = D.data_ptr(), 1 if obf else 0, N, M * N
    a.alpha, a.beta = 1.0, 1.0
    if res:
        a.R, a.rdt, a.ldr, a.rbs = R.data_ptr(), 1, N, M * N
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.test_gemm(a, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.test_gemm(a, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        out = [run(s) for s in (SHAPES[:5] + [SHAPES[18], SHAPES[19], SHAPES[21], SHAPES[30]] if os.environ.get("BIG") else SHAPES)]
        print("RESULT " + json.dumps(out))
        sys.exit(0)
    table = {}
    for cfg in (["auto", 0, 1, 5, 6] if os.environ.get("BIG") else ["auto", 0, 1, 4, 2, 3]):
        env = dict(os.environ)
        if cfg != "auto":
            env["DGSCT_GEMM_CFG"] = str(cfg)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        table[cfg] = json.loads(line[0][7:]) if line else None
    print("shape (M,N,K,KB,batch,ak,bk)".ljust(40) + "".join(f"{str(c):>9}" for c in table))
    for i, s in enumerate(SHAPES[:5] + [SHAPES[18], SHAPES[19], SHAPES[21], SHAPES[30]] if os.environ.get("BIG") else SHAPES):
        print(str(s[:7]).ljust(40) + "".join(f"{(table[c][i] if table[c] else float('nan')):9.1f}" for c in table))
