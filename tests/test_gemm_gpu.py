"""GPU: the MFMA GEMM engine (csrc/gemm.hip) against torch, through the C ABI test hook.
Covers both modes (bf16 / fp32 MFMA), all four operand-layout combinations (K-major = ds_read_b128
fragments, MN-major = ds_read_b64_tr_b16 transpose reads), every tile configuration, ragged sizes,
two-level contraction, split-K with atomics and every epilogue feature."""
import ctypes

import pytest
import torch

import dgsct_amd
from dgsct_amd._lib import GemmArgs, default_lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _operand(batch, KB, rows, K, kmajor, shared, dtype, pad, gen):
    """logical X[b][r][(kb,k)] -> (memory tensor, ld, bs, kbs, logical fp64 tensor [batch,rows,KB*K])"""
    nb = 1 if shared else batch
    if kmajor:
        ld = K + pad
        mem = torch.randn(nb, KB, rows, ld, generator=gen).to(dtype)
        logical = mem[..., :K].double().permute(0, 2, 1, 3).reshape(nb, rows, KB * K)
        kbs, bs = rows * ld, KB * rows * ld
    else:
        ld = rows + pad
        mem = torch.randn(nb, KB, K, ld, generator=gen).to(dtype)
        logical = mem[..., :rows].double().permute(0, 3, 1, 2).reshape(nb, rows, KB * K)
        kbs, bs = K * ld, KB * K * ld
    if shared:
        logical = logical.expand(batch, -1, -1)
        bs = 0
    return mem.to(DEV).contiguous(), ld, bs, kbs, logical


def run_case(mode, M, N, K, ak, bk, batch=1, KB=1, shared_a=False, pad=0, epi=None, out_bf16=False, atomic=False,
             splitk=1, seed=0):
    lib = default_lib()
    epi = epi or {}
    gen = torch.Generator().manual_seed(seed)
    dtype = torch.bfloat16 if mode == 1 else torch.float32
    Am, lda, a_bs, a_kbs, Al = _operand(batch, KB, M, K, ak, shared_a, dtype, pad, gen)
    Bm, ldb, b_bs, b_kbs, Bl = _operand(batch, KB, N, K, bk, False, dtype, pad, gen)
    ref = torch.einsum("bmk,bnk->bmn", Al, Bl)
    ldd = N + (pad if not atomic else 0)
    ddt = 1 if out_bf16 else 0
    D = torch.full((batch, M, ldd), float("nan"), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=DEV)
    if atomic:
        D.zero_()
    a = GemmArgs()
    a.mode, a.M, a.N, a.K, a.KB, a.batch, a.splitk, a.atomic = mode, M, N, K, KB, batch, splitk, int(atomic)      # (atomic = 2: sole writer)
    a.A, a.lda, a.a_kmajor, a.a_bs, a.a_kbs = Am.data_ptr(), lda, int(ak), a_bs, a_kbs
    a.B, a.ldb, a.b_kmajor, a.b_bs, a.b_kbs = Bm.data_ptr(), ldb, int(bk), b_bs, b_kbs
    a.D, a.ddt, a.ldd, a.dbs = D.data_ptr(), ddt, ldd, M * ldd
    a.alpha, a.beta = epi.get("alpha", 1.0), epi.get("beta", 1.0)
    keep = []
    alpha = a.alpha
    if epi.get("alpha_ptr"):
        t = torch.tensor([0.37], device=DEV); keep.append(t); a.alpha_ptr = t.data_ptr(); alpha *= 0.37
    ref = alpha * ref
    m_mod = epi.get("m_mod", 0)
    a.m_mod = m_mod
    midx = torch.arange(M) % m_mod if m_mod else torch.arange(M)
    if epi.get("bias_m"):
        t = torch.randn(m_mod or M, generator=gen); keep.append(t.to(DEV)); a.bias_m = keep[-1].data_ptr()
        ref = ref + t.double()[midx][None, :, None]
    if epi.get("bias_n"):
        per_batch = epi.get("bias_n_batched", False)
        t = torch.randn(batch if per_batch else 1, N, generator=gen); keep.append(t.to(DEV)); a.bias_n = keep[-1].data_ptr()
        a.bias_n_bs = N if per_batch else 0
        ref = ref + t.double()[:, None, :]
    if epi.get("r1"):
        t1 = torch.randn(m_mod or M, generator=gen); t2 = torch.randn(N, generator=gen)
        keep += [t1.to(DEV), t2.to(DEV)]; a.r1_m, a.r1_n = keep[-2].data_ptr(), keep[-1].data_ptr()
        ref = ref + (t1.double()[midx][:, None] * t2.double()[None, :])[None]
    a.act = epi.get("act", 0)
    if a.act == 1:
        ref = ref.clamp_min(0)
    elif a.act == 2:
        ref = torch.sigmoid(ref)
    if epi.get("mask"):
        t = torch.randn(batch, M, N, generator=gen).to(dtype); keep.append(t.to(DEV)); a.mask = keep[-1].data_ptr()
        a.ldmask, a.maskbs = N, M * N
        ref = ref * (t.double() > 0)
    if epi.get("R"):
        rdt = epi.get("rdt", 0)
        t = torch.randn(batch, M, N, generator=gen).to(torch.bfloat16 if rdt else torch.float32)
        keep.append(t.to(DEV)); a.R, a.rdt, a.ldr, a.rbs = keep[-1].data_ptr(), rdt, N, M * N
        ref = ref + a.beta * t.double()
        if epi.get("R2"):
            t2 = torch.randn(batch, M, N, generator=gen).to(torch.bfloat16 if rdt else torch.float32)
            keep.append(t2.to(DEV)); a.R2 = keep[-1].data_ptr()
            ref = ref + t2.double()
    lib.test_gemm(a, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = D[..., :N].double().cpu()
    scale = max(1.0, ref.abs().max().item())
    err = ((got - ref).abs().max() / scale).item()
    tol = (4e-3 if out_bf16 else 2e-4) if mode == 1 else (4e-3 if out_bf16 else 2e-5)
    assert err == err and err < tol, f"gemm mode={mode} M={M} N={N} K={K} ak={ak} bk={bk} batch={batch} KB={KB}: err {err:.3e}"
    if pad and not atomic:
        assert torch.isnan(D[..., N:].float()).all(), "GEMM wrote outside its N columns"


LAYOUTS = [(1, 1), (1, 0), (0, 1), (0, 0)]


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("ak,bk", LAYOUTS)
def test_gemm_tile_configs(mode, ak, bk):
    # 128x128, 128x96, 128x32, 32x128, 64x64 tiles (see gemm_mode's heuristic)
    for (M, N, K, batch) in [(256, 256, 96, 3), (256, 96, 160, 2), (200, 24, 64, 2), (32, 300, 64, 2), (100, 72, 40, 2)]:
        run_case(mode, M, N, K, ak, bk, batch=batch)


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("ak,bk", LAYOUTS)
def test_gemm_ragged_unaligned(mode, ak, bk):
    # sizes that are multiples of nothing, leading dimensions that break 16-byte alignment
    for (M, N, K, pad) in [(36, 36, 36, 0), (37, 5, 19, 3), (130, 131, 33, 1), (4, 2, 2, 0), (65, 97, 8, 6)]:
        run_case(mode, M, N, K, ak, bk, batch=2, pad=pad)


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("ak,bk", LAYOUTS)
def test_gemm_two_level_k_splitk_atomic(mode, ak, bk):
    run_case(mode, 144, 100, 24, ak, bk, batch=1, KB=7, atomic=True, splitk=0)       # dWn-style contraction over (b, c)
    run_case(mode, 64, 48, 1000, ak, bk, batch=2, atomic=True, splitk=0)            # weight-gradient style, K = tokens
    run_case(mode, 64, 48, 1000, ak, bk, batch=1, atomic=True, splitk=3)


@pytest.mark.parametrize("mode", [1, 0])
def test_gemm_shared_a_and_epilogues(mode):
    run_case(mode, 150, 96, 70, 1, 0, batch=4, shared_a=True)
    run_case(mode, 80, 64, 32, 1, 1, batch=2, epi=dict(bias_n=True, act=1), out_bf16=(mode == 1))
    run_case(mode, 80, 64, 32, 1, 1, batch=2, epi=dict(bias_n=True, bias_n_batched=True, act=2))
    run_case(mode, 120, 40, 48, 1, 1, batch=1, epi=dict(r1=True, bias_n=True, m_mod=30))
    run_case(mode, 90, 40, 16, 1, 0, batch=3, epi=dict(alpha_ptr=True, R=True, rdt=mode), out_bf16=(mode == 1))
    run_case(mode, 33, 40, 64, 1, 0, batch=1, epi=dict(mask=True, bias_m=True))
    run_case(mode, 64, 64, 4, 1, 0, batch=2, epi=dict(R=True, beta=1.0))           # K smaller than one k-tile (tk = 2 / 4)


def test_gemm_second_residual():
    run_case(1, 300, 96, 32, 1, 0, batch=3, epi=dict(alpha_ptr=True, R=True, R2=True, rdt=1), out_bf16=True)   # dX = dX1 + dS2.tok + dOut
    run_case(0, 70, 40, 16, 1, 0, batch=2, epi=dict(R=True, R2=True, rdt=0))


@pytest.mark.parametrize("ak,bk", LAYOUTS)
def test_gemm_deep_one_level_contraction(ak, bk):
    """K = 200 000 > 65 536: the frame split of the FAST staging must not be applied to one-level contractions (a 32-bit
    multiply-high overflow corrupted the last k-rows of every 160-frame weight gradient before it was fixed)."""
    run_case(1, 64, 48, 200000, ak, bk, batch=1, atomic=True, splitk=0)


@pytest.mark.parametrize("M,N", [(96, 96), (128, 128), (128, 96), (96, 128), (72, 120)])
def test_gemm_single_tile_deep_weight_gradient(M, N):
    """stage-0 weight-gradient shapes (64 < M, N <= 128, K >= 131 072 token rows): one tile over the whole output, 256-way
    split-K with atomics"""
    run_case(1, M, N, 140000, 0, 0, batch=1, atomic=True, splitk=0)
    run_case(1, M, N, 960, 1, 1 if M == N else 0, batch=1, KB=160, atomic=True, splitk=0)      # two-level K, kflat = 153 600


def test_gemm_deep_tiles_and_shared_operand():
    run_case(1, 512, 96, 2048, 1, 1, batch=3, shared_a=True, out_bf16=True)      # 256 x 96 tile (remap forward)
    run_case(1, 512, 96, 2048, 1, 0, batch=3, shared_a=True, out_bf16=True)
    run_case(1, 256, 256, 96, 1, 0, batch=1, KB=40, atomic=True, splitk=0)       # dWn-style: two-level K, 128 x 128, split-K


@pytest.mark.parametrize("ak", [1, 0])
def test_gemm_shared_a_xcd_grouped_order(ak):
    """Shared A, batch a multiple of 8, one column of tiles, deep K: the launch order is regrouped per XCD (m-groups x
    batches); every (tile, batch) must still be computed exactly once -- 256 x 96 tiles (5 of them: a partial last
    group), 128 x 96 tiles (9), and 64 x 64 fp32."""
    run_case(1, 1280, 96, 640, ak, 1, batch=16, shared_a=True, out_bf16=True)
    run_case(1, 1100, 96, 576, ak, 0, batch=24, shared_a=True, out_bf16=True)
    run_case(0, 200, 40, 512, ak, 1, batch=16, shared_a=True)


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("tk", [32, 12, 4])
def test_gemm_softmax_epilogues(mode, tk):
    """ACT_SOFTMAX / ACT_SOFTMAX_BWD: column softmax over the M = tk rows, transposed output (X <- token attention)."""
    lib = default_lib()
    gen = torch.Generator().manual_seed(3)
    dtype = torch.bfloat16 if mode == 1 else torch.float32
    batch, Ntok, C = 2, 300, 96
    tkp = (tk + 7) // 8 * 8
    tok = (0.3 * torch.randn(batch, tk, C, generator=gen)).to(dtype)
    X = torch.randn(batch, Ntok, C, generator=gen).to(dtype)
    logits = torch.einsum("btc,bnc->bnt", tok.double(), X.double())            # [b, n, t]
    P_ref = torch.softmax(logits, dim=-1)
    a = GemmArgs()
    a.mode, a.M, a.N, a.K, a.KB, a.batch, a.splitk, a.atomic = mode, tk, Ntok, C, 1, batch, 1, 0
    tokd, Xd = tok.to(DEV).contiguous(), X.to(DEV).contiguous()
    a.A, a.lda, a.a_kmajor, a.a_bs, a.a_kbs = tokd.data_ptr(), C, 1, tk * C, 0
    a.B, a.ldb, a.b_kmajor, a.b_bs, a.b_kbs = Xd.data_ptr(), C, 1, Ntok * C, 0
    P = torch.zeros(batch, Ntok, tkp, dtype=dtype, device=DEV)
    a.D, a.ddt, a.ldd, a.dbs = P.data_ptr(), mode, tkp, Ntok * tkp
    a.alpha, a.beta, a.act = 1.0, 1.0, 3
    st = torch.cuda.current_stream().cuda_stream
    lib.test_gemm(a, st)
    torch.cuda.synchronize()
    tol = 1e-2 if mode == 1 else 2e-5
    assert (P[..., :tk].double().cpu() - P_ref).abs().max().item() < tol
    # backward: dS = s * P * (U - sum_t P U), U^T = tok . dX1^T ; dot = sum P U
    dX1 = torch.randn(batch, Ntok, C, generator=gen).to(dtype)
    U = torch.einsum("btc,bnc->bnt", tok.double(), dX1.double())
    Pd = P[..., :tk].double().cpu()
    s_val = 0.3
    dot = (Pd * U).sum(-1, keepdim=True)
    dS_ref = s_val * Pd * (U - dot)
    dX1d = dX1.to(DEV).contiguous()
    a.B = dX1d.data_ptr()
    dS = torch.zeros(batch, Ntok, tkp, dtype=dtype, device=DEV)
    a.D = dS.data_ptr()
    a.act = 4
    a.mask, a.ldmask, a.maskbs = P.data_ptr(), tkp, Ntok * tkp
    sc = torch.tensor([s_val], device=DEV); acc = torch.zeros(1, device=DEV)
    a.sm_scale, a.sm_dot = sc.data_ptr(), acc.data_ptr()
    lib.test_gemm(a, st)
    torch.cuda.synchronize()
    scale = max(1.0, dS_ref.abs().max().item())
    assert (dS[..., :tk].double().cpu() - dS_ref).abs().max().item() / scale < (2e-2 if mode == 1 else 2e-5)
    assert abs(acc.item() - dot.sum().item()) < (2e-2 if mode == 1 else 1e-4) * max(1.0, abs(dot.sum().item()))


# ---- gemm8.hip: the 8-wave LDS-DMA pipelined kernel of the deep products -------------------------------------------------
@pytest.fixture(params=[1, 0], ids=["staggered", "lockstep"])
def gemm8_all(request):
    """route every eligible shape to the 8-wave kernel (its size gates would send these small test matrices to the tiled engine); both
    DMA-issue schedules of its pipelined k-loop: the two wave halves a quarter k-tile apart (round 6, default) and in lock-step"""
    lib = default_lib()
    old, olds = lib.test_tune("gemm8", 2), lib.test_tune("g8stag", request.param)
    yield lib
    lib.test_tune("gemm8", old)
    lib.test_tune("g8stag", olds)


@pytest.mark.parametrize("ak,bk", LAYOUTS)
def test_gemm8_batched_shared_a(gemm8_all, ak, bk):
    """remap products Wn . Y[b] / Wn^T . dT[b]: one A for every frame, frames side by side in the column tiles (256 x 192
    tiles over 96-wide frames: two frames per tile; 128-wide frames: 256-column tiles)"""
    run_case(1, 512, 96, 1024, ak, bk, batch=6, shared_a=True, out_bf16=True)
    run_case(1, 256, 192, 1088, ak, bk, batch=3, shared_a=True, out_bf16=True)
    run_case(1, 256, 128, 1024, ak, bk, batch=4, shared_a=True, out_bf16=True)
    run_case(1, 256, 96, 1152, ak, bk, batch=8, shared_a=True, out_bf16=False)           # fp32 rows through the staged epilogue
    run_case(1, 256, 48, 1024, ak, bk, batch=8, shared_a=True, out_bf16=True)            # four 48-wide frames per 192-column tile


def test_gemm8_remap_bias_epilogue(gemm8_all):
    """Yp[b] = Wn . T2[b] + rowb (x) colb + colb2 (plan.cpp F1, association B): rank-1 + column bias in the staged epilogue"""
    run_case(1, 512, 96, 1024, 1, 1, batch=4, shared_a=True, out_bf16=True, epi=dict(r1=True, bias_n=True))
    run_case(1, 256, 128, 1024, 1, 1, batch=2, shared_a=True, out_bf16=True, epi=dict(r1=True, bias_n=True))


@pytest.mark.parametrize("ak,bk", LAYOUTS)
def test_gemm8_split_k_atomic(gemm8_all, ak, bk):
    """weight gradients: plain big output, deep one- or two-level contraction, split-K with fp32 atomics"""
    run_case(1, 256, 256, 96, ak, bk, batch=1, KB=24, atomic=True, splitk=0)              # dWn: contraction over (frame, channel)
    run_case(1, 512, 192, 2304, ak, bk, batch=1, atomic=True, splitk=0)
    run_case(1, 256, 512, 6400, ak, bk, batch=1, atomic=True, splitk=0)                   # C x C over token rows


@pytest.mark.parametrize("ak,bk", [(1, 1), (1, 0)])
def test_gemm8_unsplit_weight_gradient(ak, bk):
    """the remap weight gradient dWn at a reduced size but in its production mode (default size gates): >= 96 output tiles of a
    two-level contraction >= 64 k-tiles deep, D pre-zeroed with the product as its only writer (atomic = 2) -> ONE workgroup per
    tile, no split, plain fp32 rows instead of seven slabs of atomics"""
    import csv, os, tempfile
    lib = default_lib()
    lib.prof_enable(True)
    try:
        path = os.path.join(tempfile.mkdtemp(), "g.csv")
        os.environ["DGSCT_PROF_DUMP"] = path
        run_case(1, 2560, 2304, 96, ak, bk, batch=1, KB=48, atomic=2, splitk=0)          # 10 x 12 (192-wide) tiles, 72 k-tiles
        lib.prof_collect()
        rows = list(csv.DictReader(open(path)))
        assert rows and rows[-1]["cfg"] in ("8", "9") and rows[-1]["splitk"] == "1", rows[-1]
    finally:
        os.environ.pop("DGSCT_PROF_DUMP", None)
        lib.prof_enable(False)


@pytest.mark.parametrize("ak,bk", [(0, 0), (1, 1), (1, 0)])
def test_gemm8_residual_in_the_row_pass(gemm8_all, ak, bk):
    """the dY product of the pair backward (plan.cpp dy_product): dY[b] = Wn^T . dT[b] + dX_other[b] -- a bf16 residual laid out like
    the output, added in the staged row pass; must stay on gemm8 (a residual used to send the product to the tiled engine)"""
    import csv, os, tempfile
    lib = gemm8_all
    lib.prof_enable(True)
    try:
        path = os.path.join(tempfile.mkdtemp(), "g.csv")
        os.environ["DGSCT_PROF_DUMP"] = path
        run_case(1, 512, 96, 1024, ak, bk, batch=6, shared_a=True, out_bf16=True, epi=dict(R=True, rdt=1))
        lib.prof_collect()
        rows = list(csv.DictReader(open(path)))
        assert rows and rows[-1]["cfg"] in ("8", "9"), rows[-1]
        run_case(1, 256, 128, 1024, ak, bk, batch=4, shared_a=True, out_bf16=True, epi=dict(R=True, rdt=1))
        run_case(1, 256, 256, 1088, ak, bk, batch=1, out_bf16=True, epi=dict(R=True, rdt=1))
    finally:
        os.environ.pop("DGSCT_PROF_DUMP", None)
        lib.prof_enable(False)


def test_gemm8_is_actually_used(gemm8_all):
    """the shapes above must reach gemm8.hip (and a shape it cannot take must still be served by the tiled engine)"""
    lib = gemm8_all
    lib.prof_enable(True)
    try:
        run_case(1, 512, 96, 1024, 1, 0, batch=6, shared_a=True, out_bf16=True)
        import os, tempfile, csv
        path = os.path.join(tempfile.mkdtemp(), "g.csv")
        os.environ["DGSCT_PROF_DUMP"] = path
        lib.prof_collect()
        rows = list(csv.DictReader(open(path)))
        assert rows and rows[-1]["cfg"] in ("8", "9"), rows[-1]
        run_case(1, 500, 96, 1024, 1, 0, batch=6, shared_a=True, out_bf16=True)           # M % 256 != 0: tiled engine
        lib.prof_collect()
        rows = list(csv.DictReader(open(path)))
        assert rows and rows[-1]["cfg"] not in ("8", "9"), rows[-1]
    finally:
        os.environ.pop("DGSCT_PROF_DUMP", None)
        lib.prof_enable(False)


# ---- gemm_skinny.hip: [BT, C] x [C, C] gate-MLP products, contraction split over the four waves of a 32 x 32 tile ---------------
def test_gemm_skinny_epilogues_and_edges():
    """every epilogue the adapter schedule uses on these products (plan.cpp F2, F4-F6, B4, B6): bias + ReLU / sigmoid, fp32 or bf16
    out, the ReLU mask of another tensor, an fp32 residual; ragged M (not a multiple of 32), N = 48, K from 64 to 1536"""
    run_case(1, 160, 512, 512, 1, 1, epi=dict(bias_n=True, act=1), out_bf16=True)          # aq1 / q
    run_case(1, 160, 256, 512, 1, 1, epi=dict(bias_n=True, act=2))                          # ch (fp32 out, sigmoid)
    run_case(1, 160, 256, 512, 1, 1, epi=dict(mask=True), out_bf16=True)                    # dq = (dpre . Wcatt) * (q > 0)
    run_case(1, 160, 512, 256, 1, 1, epi=dict(R=True, rdt=0))                               # da += dpa2 . Wa2 (fp32 residual)
    run_case(1, 160, 1024, 1024, 1, 1, epi=dict(bias_n=True, act=1), out_bf16=True)
    run_case(1, 160, 1536, 1536, 1, 1, epi=dict(bias_n=True, act=1), out_bf16=True)
    run_case(1, 50, 48, 64, 1, 1)                                                           # one k-step per wave, ragged M, N = 48
    run_case(1, 7, 96, 192, 1, 1, out_bf16=True)
    run_case(1, 256, 384, 768, 1, 1, epi=dict(bias_n=True))


def test_gemm_skinny_is_actually_used():
    import csv, os, tempfile
    lib = default_lib()
    assert lib.test_tune("skinny", -1) == 1
    lib.prof_enable(True)
    try:
        path = os.path.join(tempfile.mkdtemp(), "g.csv")
        os.environ["DGSCT_PROF_DUMP"] = path
        run_case(1, 160, 512, 512, 1, 1, epi=dict(bias_n=True, act=1), out_bf16=True)
        lib.prof_collect()
        rows = list(csv.DictReader(open(path)))
        assert rows and rows[-1]["cfg"] == "10", rows[-1]
        run_case(1, 160, 512, 96, 1, 1, out_bf16=True)                                      # K % 64 != 0: tiled engine
        lib.prof_collect()
        rows = list(csv.DictReader(open(path)))
        assert rows and rows[-1]["cfg"] != "10", rows[-1]
    finally:
        os.environ.pop("DGSCT_PROF_DUMP", None)
        lib.prof_enable(False)


# ---- gemm_tall.hip: weight gradients over the token rows (both operands [rows][width], small output, very deep contraction) ----------
@pytest.mark.parametrize("M,N,pad", [(96, 96, 0), (128, 128, 0), (48, 96, 0), (8, 48, 4), (48, 8, 4), (96, 128, 0), (12, 192, 0), (32, 256, 0), (72, 120, 0)])
def test_gemm_tall_weight_gradients(M, N, pad):
    """the weight-gradient products of stages 0-1 (dWu, dWd, dWv2, dWv1, dWc: plan.cpp B10, B9, B5, B1) on gemm_tall.hip: a stream over
    two [rows, width] tensors in 64-row blocks, output tiles dealt to the waves, fp32 atomics at the end; padded row pitches (a slab of
    a wider tensor), outputs smaller than a tile, and the check that the kernel is the one that ran"""
    import csv, os, tempfile
    lib = default_lib()
    lib.prof_enable(True)
    old = lib.test_tune("gemmtall", 2)              # 2: from 16 384 rows (the default gate is the stage-0 depth)
    try:
        path = os.path.join(tempfile.mkdtemp(), "g.csv")
        os.environ["DGSCT_PROF_DUMP"] = path
        run_case(1, M, N, 64 * 1100, 0, 0, batch=1, atomic=True, splitk=0, pad=pad)
        lib.prof_collect()
        rows = list(csv.DictReader(open(path)))
        assert rows and rows[-1]["cfg"] == "11", rows[-1]
        run_case(1, M, N, 64 * 70, 0, 0, batch=1, KB=16, atomic=True, splitk=0, pad=pad)           # (frame, row) contraction, frames back to back: one flat stream
        lib.prof_collect()
        rows = list(csv.DictReader(open(path)))
        assert rows and rows[-1]["cfg"] == "11", rows[-1]
        run_case(1, M, N, 64 * 1100 + 32, 0, 0, batch=1, atomic=True, splitk=0, pad=pad)          # not whole 64-row blocks: the tiled engine
        lib.prof_collect()
        rows = list(csv.DictReader(open(path)))
        assert rows and rows[-1]["cfg"] != "11", rows[-1]
    finally:
        lib.test_tune("gemmtall", old)
        os.environ.pop("DGSCT_PROF_DUMP", None)
        lib.prof_enable(False)
