// Kernel schedule of one DG-SCT adapter forward / backward, written against prims.h only (plain C++,
// no HIP syntax).  It follows oracle/dgsct_oracle.py step by step (F1..F11 / B11..B1), which in turn
// restates reference DG-SCT/AVE/nets/net_trans.py:552-674 and its autograd.
//
// Data layout in HBM (all token-major, row = one token, channels contiguous):
//   activations  E = desc.dtype (bf16 or fp32); per-frame gate vectors, statistics, logits: fp32.
//   `saved`  : what backward needs from forward (Yp, remap intermediate, P1, tok, P2, X1, vq1, Xc,
//              vq2, X3, Zp, Z, Op + small fp32 vectors) -- one region per adapter call.
//   `ws`     : scratch that dies with the call (logits, cotangents), shared by all adapters of a stream.
//   `prep`   : MFMA-operand (E) copies of the weights + three derived bias vectors.
#include "plan.h"
#include <cstdlib>

#include <atomic>
#include <functional>
#include <vector>

#include <cstdio>
#include <cstring>

#include "err.h"
#include "prims.h"

namespace dgsct {

static inline int64_t rup(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

struct Arena {
  int64_t off = 0;
  std::vector<Region>* regs = nullptr;
  int64_t take(const char* name, int64_t bytes) {
    int64_t o = off;
    off += rup(bytes, 256);
    if (regs) regs->push_back(Region{name, o, bytes});
    return o;
  }
};

// ------------------------------------------------------------------------------------------------
Plan::Plan(const dgsct_adapter_desc& d_, bool record_regions) : record_regions_(record_regions), d(d_) {
  B = d.BT; N = d.N; C = d.C; No = d.No; Co = d.Co; tk = d.tk; g = d.g;
  dd = C / 2; ds = d.r > 0 ? C / d.r : 0;
  fp8 = d.dtype == DGSCT_BF16_FP8;
  E = fp8 ? (int)DT_BF16 : d.dtype; es = (int64_t)dt_size(E);
  Np = (int)rup(N, 8); Nop = (int)rup(No, 8); tkp = (int)rup(tk, 8);
  R = (int64_t)B * N;
  {
    const int64_t a = (int64_t)N * No * Co + (int64_t)N * Co * C;
    const int64_t b = (int64_t)No * Co * C + (int64_t)N * No * C;
    orderA = a <= b;
  }
  ok = validate();
  // more than one 32-row MFMA tile of latent tokens per frame: attn_wide.cpp instead of the fused kernels.  DGSCT_WIDE_ATTN=1 (read per
  // layout: tests set it around a call) sends every tk down that path, so that the goldens of the fused kernels check it too.
  { const char* e = getenv("DGSCT_WIDE_ATTN"); wide = tk > 32 || (e && atoi(e) != 0); }
  if (ok) xc_scratch = !fp8 && gate_bwd_fused_shape(E, N, C, ds, g);
  if (ok) layout();
}

bool Plan::validate() {
  auto bad = [&](const char* m) { set_error("dgsct: bad descriptor: %s", m); return false; };
  if (B <= 0 || N <= 0 || C <= 0 || No <= 0 || Co <= 0 || tk <= 0) return bad("non-positive dimension");
  if (d.r <= 0 || g <= 0 || C % d.r || ds % g || C % g) return bad("C must be divisible by r and g, C/r by g");
  if (C % 4 || C > 1536) return bad("C must be a multiple of 4 and <= 1536");
  if (dd % 4) return bad("C/2 must be a multiple of 4");
  if (tk > 1024) return bad("tk must be <= 1024");
  if (d.dtype != DGSCT_F32 && d.dtype != DGSCT_BF16 && d.dtype != DGSCT_BF16_FP8) return bad("dtype");
  if (fp8 && (C % 16 || Co % 16)) return bad("fp8 projections need C and Co to be multiples of 16");
  if (E == DT_BF16 && C % 8) return bad("C must be a multiple of 8 in bf16 mode (16-byte rows)");
  if (N > 8192) return bad("N must be <= 8192");
  if (d.temporal && d.T > 0 && B % d.T) return bad("BT must be a multiple of T (temporal gate)");
  if (E != DT_F32 && E != DT_BF16) return bad("dtype");
  if (d.remap != DGSCT_REMAP_CONV && d.remap != DGSCT_REMAP_FIXED) return bad("remap");
  return true;
}

// ---- what-if switches (TIMING EXPERIMENTS ONLY: results are garbage).  dgsct_test_tune("skip", mask) leaves classes of launches out
// of the schedule so that the wall-time share of each class can be measured inside the real step (tools/call_overlap.py):
//   1 weight-gradient GEMMs on the aux stream    2 relu_bwd_scale    4 xc_bwd    8 bn_bwd_apply    16 scale_cols + rowdot (forward)
//   32 bn_stats + affine_act    64 colsum of vq1 / u    128 the four attention kernels    256 modln fwd / bwd    512 tail fwd / bwd
//   1024 the [rows, C] x [C, C] chain GEMMs (vq1, vq2, dXc, dX1 +=)    2048 the remap GEMMs    4096 the bottleneck GEMMs
//   8192 the backward's final join with the aux stream    16384 / 32768 / 65536 / 131072: tokattn_fwd / xattn_fwd / xattn_bwd / tokattn_bwd alone
// "skipminc": only for adapters at least that wide (default 0).
static std::atomic<int> g_skip{0}, g_skip_minc{0}, g_skip_maxc{1 << 30};
int plan_skip_mode(int set) { const int old = g_skip.load(); if (set >= 0) g_skip.store(set); return old; }
int plan_skip_minc(int set) { const int old = g_skip_minc.load(); if (set >= 0) g_skip_minc.store(set); return old; }
int plan_skip_maxc(int set) { const int old = g_skip_maxc.load(); if (set >= 0) g_skip_maxc.store(set ? set : (1 << 30)); return old; }

static bool wt_enabled() {
  static const bool off = getenv("DGSCT_NO_WT") != nullptr;      // A/B switch: no transposed weight copies
  return !off;
}

static bool vq1_dw_enabled() {
  const char* e = getenv("DGSCT_VQ1_DW");
  return e && atoi(e) != 0;
}

void Plan::layout() {
  // ---- prep: E copies of the GEMM weights (bf16 mode only) + derived fp32 vectors
  {
    Arena a;
    auto wcopy = [&](int id, int64_t numel) {
      wnumel[id] = numel;
      prep_w[id] = (E == DT_BF16) ? a.take("w", numel * es) : -1;
    };
    for (int i = 0; i < DGSCT_P_COUNT; ++i) { prep_w[i] = -1; prep_wt[i] = -1; wcols[i] = 0; wnumel[i] = 0; }
    // K-major (transposed) bf16 copies for the data-gradient GEMMs dIn = dOut . W: with W [out][in] as stored, the
    // contraction index `out` is the slow one (an MN-major B operand); at 128 x 128 that kernel variant needs 200 VGPRs
    // (2 workgroups per CU): 54.6 vs 32.9 us at 23040 x 512 x 512, and 26 vs 10 us for the M = 160 gate-MLP products
    auto wtcopy = [&](int id, int64_t cols) {
      wcols[id] = cols;
      prep_wt[id] = (E == DT_BF16 && wt_enabled()) ? a.take("wt", wnumel[id] * es) : -1;
    };
    wcopy(DGSCT_P_WN, (int64_t)N * No);
    wcopy(DGSCT_P_WC, (int64_t)C * Co);
    wcopy(DGSCT_P_WA1, (int64_t)C * C);
    wcopy(DGSCT_P_WV1, (int64_t)C * C);
    wcopy(DGSCT_P_WB, (int64_t)dd * C);
    wcopy(DGSCT_P_WV2, (int64_t)dd * C);
    wcopy(DGSCT_P_WA2, (int64_t)dd * C);
    wcopy(DGSCT_P_WCATT, (int64_t)C * dd);
    wcopy(DGSCT_P_WD, (int64_t)ds * (C / g));
    wcopy(DGSCT_P_WU, (int64_t)C * (ds / g));
    wtcopy(DGSCT_P_WC, Co); wtcopy(DGSCT_P_WA1, C); wtcopy(DGSCT_P_WV1, C); wtcopy(DGSCT_P_WB, C); wtcopy(DGSCT_P_WV2, C);
    wtcopy(DGSCT_P_WA2, C); wtcopy(DGSCT_P_WCATT, dd);
    // (round 5) the grouped bottleneck weights as well: the data-gradient products of the late stages run on the fused engine
    // (gemm_fx.hip: BatchNorm backward applied while the A operand is staged), which takes K-major operands only
    wtcopy(DGSCT_P_WU, ds / g); wtcopy(DGSCT_P_WD, C / g);
    prep_rowb = a.take("rowb", (int64_t)N * 4);
    prep_colb = a.take("colb", (int64_t)C * 4);
    prep_colb2 = a.take("colb2", (int64_t)C * 4);
    if (fp8) {
      prep_w8[0] = a.take("wc8", (int64_t)C * Co);
      prep_w8[1] = a.take("wv18", (int64_t)C * C);
      prep_w8[2] = a.take("wv28", (int64_t)dd * C);
      prep_w8scale = a.take("w8scale", 4 * 4);
    }
    prep_t0pk = (E == DT_BF16 && !wide) ? a.take("t0pk", tok_pack_elems(1, C) * 2) : -1;      // my_tokens packed for attn2.hip
    prep_t0hi = (E == DT_BF16 && wide) ? a.take("t0hi", (int64_t)tk * C * 2) : -1;            // ... as a hi / lo pair for attn_wide.cpp
    prep_t0lo = (E == DT_BF16 && wide) ? a.take("t0lo", (int64_t)tk * C * 2) : -1;
    prep_bytes = a.off;
  }
  // ---- saved
  {
    Arena a; a.regs = record_regions_ ? &saved_regions : nullptr;
    // zero block first (atomically accumulated in forward)
    s.a = a.take("a", (int64_t)B * C * 4);
    s.mvq1 = a.take("mvq1", (int64_t)B * C * 4);
    s.cnt1 = a.take("cnt1", (int64_t)B * C * 4);             // number of positive vq1 entries per (frame, channel): the bias gradient of
                                                              // fc_affine_video_1 once its ReLU backward is folded into the next product
    s.bnacc1 = a.take("bnacc1", (int64_t)3 * ds * 4);
    s.bnacc2 = a.take("bnacc2", (int64_t)3 * C * 4);
    s.zero_end = a.off;
    s.Yp = a.take("Yp", R * C * es);
    s.T = a.take("T", orderA ? R * Co * es : (int64_t)B * C * Nop * es);
    s.tok = a.take("tok", (int64_t)B * tk * C * 4);          // fp32: the un-scaled logits X . tok^T amplify its rounding
    s.tokpk = (E == DT_BF16 && !wide) ? a.take("tokpk", tok_pack_elems(B, C) * 2) : -1;   // bf16 hi / lo / transposed fragment images
    s.lse = a.take("lse", (int64_t)B * tk * 4);
    s.aE = a.take("aE", (int64_t)B * C * es);
    s.X1 = a.take("X1", R * C * es);
    s.aq1 = a.take("aq1", (int64_t)B * C * es);
    s.aq2 = a.take("aq2", (int64_t)B * dd * es);
    s.vq1 = a.take("vq1", R * C * es);
    s.m1 = a.take("m1", (int64_t)B * C * es);
    s.q = a.take("q", (int64_t)B * dd * es);
    s.ch = a.take("ch", (int64_t)B * C * 4);
    s.Xc = xc_scratch ? -1 : a.take("Xc", R * C * es);
    s.vq2 = a.take("vq2", R * dd * es);
    s.sl = a.take("sl", R * 4);
    s.sg = a.take("sg", R * 4);
    s.map = a.take("map", R * 4);
    s.tg = a.take("tg", (int64_t)B * 4);
    s.X3 = a.take("X3", R * C * es);
    s.mu_b = a.take("mu_b", R * 4);
    s.rstd_b = a.take("rstd_b", R * 4);
    s.Zp = a.take("Zp", R * ds * es);
    s.Z = a.take("Z", R * ds * es);
    s.Op = a.take("Op", R * C * es);
    s.bn1 = a.take("bn1", (int64_t)4 * ds * 4);   // mean | rstd | sc | sh
    s.bn2 = a.take("bn2", (int64_t)4 * C * 4);
    s.mu_p = a.take("mu_p", R * 4);
    s.rstd_p = a.take("rstd_p", R * 4);
    s.P1 = wide ? a.take("P1", (int64_t)B * tk * Np * es) : -1;               // softmax_N(T0 Yp^T), softmax_tk(X tok^T): saved, not recomputed
    s.P2 = wide ? a.take("P2", (int64_t)B * N * tkp * es) : -1;
    s.tokhi = (wide && E == DT_BF16) ? a.take("tokhi", (int64_t)B * tk * C * es) : -1;
    saved_bytes = a.off;
  }
  // ---- forward scratch
  {
    Arena a;
    wf.tokscr = a.take("tokscr", tokattn_scratch_floats(B, N, C) * 4);
    wf.Xc = xc_scratch ? a.take("Xc", R * C * es) : -1;       // (only the unfused test path of these shapes writes it)
    wf.wL = wide ? a.take("wL", wide_attn_image_elems(B, N, tk) * 4) : -1;
    wf.toklo = (wide && E == DT_BF16) ? a.take("toklo", (int64_t)B * tk * C * es) : -1;
    ws_fwd_bytes = a.off;
  }
  // ---- backward scratch
  {
    Arena a;
    wb.bnsums2 = a.take("bnsums2", (int64_t)2 * C * 4);
    wb.bnsums1 = a.take("bnsums1", (int64_t)2 * ds * 4);
    wb.dch = a.take("dch", (int64_t)B * C * 4);
    wb.dtg = a.take("dtg", (int64_t)B * 4);
    wb.u = a.take("u", (int64_t)B * dd * 4);
    wb.dwcsum = a.take("dwcsum", (int64_t)C * 4);
    wb.dtokF = a.take("dtokF", (int64_t)B * tk * C * 4);
    wb.dT0b = a.take("dT0b", (int64_t)B * tk * C * 4);
    wb.w2 = a.take("w2", (int64_t)B * dd * 4);                 // sum_n dsl (vq2 > 0) per (frame, channel): d bias(vq2) without the dvq2 column sum
    wb.zero_end = a.off;
    wb.dO = a.take("dO", R * C * es);
    wb.dZ = a.take("dZ", R * ds * es);
    wb.dX3 = a.take("dX3", R * C * es);
    wb.dX1 = a.take("dX1", R * C * es);
    wb.dXc = a.take("dXc", R * C * es);
    wb.Xc = xc_scratch ? a.take("Xc", R * C * es) : -1;
    wb.dvq1 = a.take("dvq1", R * C * es);                      // fused engine: the masked cotangents are written beside their inputs
    wb.dvq2 = a.take("dvq2", R * dd * es);                     // (another n-tile may still be reading the input: not in place)
    wb.dZp = a.take("dZp", R * ds * es);
    wb.t1 = a.take("t1", (int64_t)B * C * 4);
    wb.t3 = a.take("t3", (int64_t)B * dd * 4);
    wb.dsg = a.take("dsg", R * 4);
    wb.dsl = a.take("dsl", R * 4);
    wb.tmpBd = a.take("tmpBd", (int64_t)B * dd * 4);
    wb.dpre_c = a.take("dpre_c", (int64_t)B * C * es);
    wb.dq = a.take("dq", (int64_t)B * dd * es);
    wb.dm1 = a.take("dm1", (int64_t)B * C * 4);
    wb.dpa1 = a.take("dpa1", (int64_t)B * C * es);
    wb.dpa2 = a.take("dpa2", (int64_t)B * dd * es);
    wb.coef = a.take("coef", (int64_t)B * C * 4);
    wb.da = a.take("da", (int64_t)B * C * 4);
    wb.dpre_t = a.take("dpre_t", (int64_t)B * 4);
    wb.Dtok = a.take("Dtok", (int64_t)B * tk * 4);
    wb.dtokpk = (E == DT_BF16 && !wide) ? a.take("dtokpk", tok_pack_elems(B, C) * 2) : -1;
    wb.wdP = wide ? a.take("wdP", wide_attn_image_elems(B, N, tk) * 4) : -1;
    wb.wdS = wide ? a.take("wdS", wide_attn_image_elems(B, N, tk) * es) : -1;
    wb.dtokE = (wide && E == DT_BF16) ? a.take("dtokE", (int64_t)B * tk * C * es) : -1;
    wb.daN = wide ? a.take("daN", (int64_t)B * C * 4) : -1;
    wb.dYp = a.take("dYp", R * C * es);
    wb.dT = a.take("dT", orderA ? R * Co * es : (int64_t)B * No * C * es);
    wb.rowtmp = a.take("rowtmp", R * 4);
    wb.rowpart = a.take("rowpart", row_part_floats(B, C) * 4);
    wb.rowpart_v1 = a.take("rowpart_v1", row_part_floats(B, C) * 4);   // partial sums whose second stage runs on the aux stream: not reused
    wb.rowpart_v2 = a.take("rowpart_v2", row_part_floats(B, C) * 4);
    // per-workgroup dWv1 partials of vq1_bwd's experiment variant ("vq1fuse" = 3, slower, off by default): 19-34 MB that only that
    // mode reads, so the region exists only when the process opted in (DGSCT_VQ1_DW=1, read per layout: tests set it around the call);
    // without it "vq1fuse" = 3 behaves like 1
    wb.vq1part = (vq1_fused_shape(E, N, C) && vq1_dw_enabled()) ? a.take("vq1part", vq1_wpart_floats(C) * 4) : -1;
    ws_bwd_bytes = a.off;
  }
  // ---- gradients (flat fp32)
  {
    int64_t off = 0;
    for (int i = 0; i < DGSCT_P_COUNT; ++i) { grad_off[i] = -1; grad_numel[i] = 0; }
    auto gr = [&](int id, int64_t n, bool on = true) {
      if (!on) return;
      grad_off[id] = off; grad_numel[id] = n; off += rup(n, 4);
    };
    gr(DGSCT_P_GATE, 1, d.use_gate);
    gr(DGSCT_P_TOKENS, (int64_t)tk * C);
    gr(DGSCT_P_GATE_AV, 1);
    gr(DGSCT_P_WN, (int64_t)N * No, d.remap == DGSCT_REMAP_CONV);
    gr(DGSCT_P_BN, N, d.remap == DGSCT_REMAP_CONV);
    gr(DGSCT_P_WC, (int64_t)C * Co);
    gr(DGSCT_P_BC, C);
    gr(DGSCT_P_WA1, (int64_t)C * C); gr(DGSCT_P_BA1, C);
    gr(DGSCT_P_WV1, (int64_t)C * C); gr(DGSCT_P_BV1, C);
    gr(DGSCT_P_WB, (int64_t)dd * C); gr(DGSCT_P_BB, dd);
    gr(DGSCT_P_WV2, (int64_t)dd * C); gr(DGSCT_P_BV2, dd);
    gr(DGSCT_P_WA2, (int64_t)dd * C); gr(DGSCT_P_BA2, dd);
    gr(DGSCT_P_WS, dd); gr(DGSCT_P_BS, 1);
    gr(DGSCT_P_WCATT, (int64_t)C * dd); gr(DGSCT_P_BCATT, C);
    gr(DGSCT_P_WD, (int64_t)ds * (C / g));
    gr(DGSCT_P_WU, (int64_t)C * (ds / g));
    // BatchNorm gradients are laid out [d bias | d weight] back to back: that is exactly the [sum dy | sum dy*xhat]
    // pair the BN-backward reductions accumulate, so the kernels write the parameter gradients in place.
    auto gr_bn = [&](int idw, int idb, int64_t n) {
      if (!d.use_bn) return;
      grad_off[idb] = off; grad_numel[idb] = n;
      grad_off[idw] = off + n; grad_numel[idw] = n;
      off += rup(2 * n, 4);
    };
    gr_bn(DGSCT_P_BN1_W, DGSCT_P_BN1_B, ds);
    gr_bn(DGSCT_P_BN2_W, DGSCT_P_BN2_B, C);
    gr(DGSCT_P_LNB_W, C, d.ln_before); gr(DGSCT_P_LNB_B, C, d.ln_before);
    gr(DGSCT_P_LNP_W, C, d.ln_post); gr(DGSCT_P_LNP_B, C, d.ln_post);
    gr(DGSCT_P_WT, C, d.temporal); gr(DGSCT_P_BT, 1, d.temporal);
    grad_floats = off;
  }
}

// ------------------------------------------------------------------------------------------------
namespace {
inline MatOp km(const void* p, long ld, long bs = 0, long kbs = 0) { MatOp m; m.p = p; m.ld = ld; m.kmajor = 1; m.bs = bs; m.kbs = kbs; return m; }
inline MatOp mn(const void* p, long ld, long bs = 0, long kbs = 0) { MatOp m; m.p = p; m.ld = ld; m.kmajor = 0; m.bs = bs; m.kbs = kbs; return m; }
inline Gemm mk(int M, int N, int K, int batch = 1) { Gemm g; g.M = M; g.N = N; g.K = K; g.batch = batch; return g; }
inline void outE(Gemm& g, void* D, int E, long ld, long dbs = 0) { g.D = D; g.ddt = E; g.ldd = ld; g.dbs = dbs; }
inline void outF(Gemm& g, float* D, long ld, long dbs = 0) { g.D = D; g.ddt = DT_F32; g.ldd = ld; g.dbs = dbs; }
inline void resid(Gemm& g, const void* R, int rdt, long ld, long rbs = 0, float beta = 1.f) { g.R = R; g.rdt = rdt; g.ldr = ld; g.rbs = rbs; g.beta = beta; }
inline void atomic_out(Gemm& g) { g.atomic = 1; g.splitk = 0; }
inline EwArg F32(const void* p) { EwArg a; a.p = p; a.dt = DT_F32; return a; }
inline EwArg Earg(const void* p, int E) { EwArg a; a.p = p; a.dt = E; return a; }
const EwArg NOARG{};
}  // namespace

struct Bound {
  // resolved pointers of one call
  const Plan& P;
  float* const* params;
  const char* prep;
  char* saved;
  char* ws;
  Ctx ctx;
  Bound(const Plan& p, float* const* pr, const void* prep_, void* saved_, void* ws_, void* stream)
      : P(p), params(pr), prep((const char*)prep_), saved((char*)saved_), ws((char*)ws_), ctx{stream, p.E} {}
  const float* F(int id) const { return params[id]; }
  float* Fm(int id) const { return params[id]; }
  const void* W(int id) const { return P.E == DT_BF16 ? (const void*)(prep + P.prep_w[id]) : (const void*)params[id]; }
  // B operand of dIn = dOut . W for W [out][in] (row stride `in`): the K-major transposed copy when prepare made one
  MatOp WB(int id, long in, long out) const {
    return P.prep_wt[id] >= 0 ? km(prep + P.prep_wt[id], out) : mn(W(id), in);
  }
  // ... of a GROUPED projection W [out][in / g] (row stride `in_g`, group stride gs): the transposed copy [in_g][out] puts group b's
  // block at column b * out_g (K-major: row = in-group input channel, ld = out, batch stride out_g)
  MatOp WBg(int id, long in_g, long out, long out_g, long gs) const {
    return P.prep_wt[id] >= 0 ? km(prep + P.prep_wt[id], out, out_g) : mn(W(id), in_g, gs);
  }
  template <typename T = void> T* S(int64_t off) const { return reinterpret_cast<T*>(saved + off); }
  template <typename T = void> T* Wk(int64_t off) const { return reinterpret_cast<T*>(ws + off); }
  const float* rowb() const { return reinterpret_cast<const float*>(prep + P.prep_rowb); }
  const float* colb() const { return reinterpret_cast<const float*>(prep + P.prep_colb); }
  const float* colb2() const { return reinterpret_cast<const float*>(prep + P.prep_colb2); }
};

// ------------------------------------------------------------------------------------------------
int Plan::prepare(float* const* params, void* prep, void* stream) const {
  Ctx ctx{stream, E};
  char* p = (char*)prep;
  CvtSeg segs[CVT_MAX_SEG];
  int ns = 0;
  if (E == DT_BF16)
    for (int i = 0; i < DGSCT_P_COUNT; ++i)
      if (prep_w[i] >= 0) {
        if (!params[i]) { set_error("dgsct_prepare: parameter %d is NULL", i); return 2; }
        segs[ns++] = CvtSeg{params[i], p + prep_w[i], (long)wnumel[i], E, 0};
        if (prep_wt[i] >= 0) segs[ns++] = CvtSeg{params[i], p + prep_wt[i], (long)wnumel[i], E, (long)wcols[i]};
      }
  float* rowb = (float*)(p + prep_rowb);
  float* colb = (float*)(p + prep_colb);
  float* colb2 = (float*)(p + prep_colb2);
  if (d.remap == DGSCT_REMAP_CONV) {
    // Yp = Wn.Y.Wc^T + bn (x) rowsum(Wc) + 1 (x) bc                       (net_trans.py:553-554)
    segs[ns++] = CvtSeg{params[DGSCT_P_BN], rowb, (long)N, DT_F32, 0};
    segs[ns++] = CvtSeg{params[DGSCT_P_BC], colb2, (long)C, DT_F32, 0};
    cvt_multi(ctx, segs, ns);
    rowsum_f32(ctx, params[DGSCT_P_WC], C, Co, colb);
  } else {
    // Yp = Wfix.(Y.Wc^T + bc) = Wfix.Y.Wc^T + rowsum(Wfix) (x) bc        (PVT_AVSModel.py:190-197)
    segs[ns++] = CvtSeg{params[DGSCT_P_BC], colb, (long)C, DT_F32, 0};
    cvt_multi(ctx, segs, ns);
    rowsum_f32(ctx, params[DGSCT_P_WN], N, No, rowb);
    zero(ctx, colb2, (size_t)C * 4);
  }
  if (prep_t0pk >= 0) tok_pack(ctx, params[DGSCT_P_TOKENS], 1, tk, C, p + prep_t0pk);
  if (prep_t0hi >= 0) split_hilo(ctx, params[DGSCT_P_TOKENS], (long)tk * C, p + prep_t0hi, p + prep_t0lo);
  if (fp8) {
    float* sc = (float*)(p + prep_w8scale);
    fp8_quantize(ctx, params[DGSCT_P_WC], (long)C * Co, p + prep_w8[0], sc + 0, sc + 3);
    fp8_quantize(ctx, params[DGSCT_P_WV1], (long)C * C, p + prep_w8[1], sc + 1, sc + 3);
    fp8_quantize(ctx, params[DGSCT_P_WV2], (long)dd * C, p + prep_w8[2], sc + 2, sc + 3);
  }
  check_async("dgsct_prepare");
  return has_error() ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
int Plan::forward(float* const* params, const void* prep, const void* X, const void* Y, void* out, float* map,
                  float* tmap, void* saved, void* ws, void* stream, const void* residual, void* aux_stream) const {
  Bound b(*this, params, prep, saved, ws, stream);
  b.ctx.aux = aux_stream;
  const Ctx& ctx = b.ctx;
  // The audio-query branch (a = mean_N Yp -> aq1, aq2) depends on Yp only: on the aux stream it runs beside the two
  // token attentions instead of extending the chain by four small launches.
  Ctx side = ctx;
  if (aux_stream) { side.stream = aux_stream; side.aux = nullptr; }
  const float invN = 1.f / (float)N;
  const int SK = (C >= g_skip_minc.load(std::memory_order_relaxed) && C <= g_skip_maxc.load(std::memory_order_relaxed)) ? g_skip.load(std::memory_order_relaxed) : 0;   // what-if switches
  zero(ctx, b.S(0), (size_t)s.zero_end);

  // F1 ---- cross-modal remap                                            net_trans.py:553-555
  void* Yp = b.S(s.Yp);
  if (orderA) {
    Gemm g1 = mk(N, Co, No, B);                                  // T1[b] = Wn . Y[b]
    g1.A = km(b.W(DGSCT_P_WN), No);
    g1.B = mn(Y, Co, (long)No * Co);
    outE(g1, b.S(s.T), E, Co, (long)N * Co);
    if (!(SK & 2048)) gemm(ctx, g1);
    Gemm g2 = mk((int)R, C, Co);                                 // Yp = T1 . Wc^T + rank-1 bias
    g2.A = km(b.S(s.T), Co);
    g2.B = km(b.W(DGSCT_P_WC), Co);
    g2.r1_m = b.rowb(); g2.r1_n = b.colb(); g2.m_mod = N; g2.bias_n = b.colb2();
    outE(g2, Yp, E, C);
    if (fp8) {
      gemm_fp8(ctx, (int)R, C, Co, b.S(s.T), Co, b.prep + prep_w8[0], (const float*)(b.prep + prep_w8scale) + 0, b.colb2(), 0, Yp, C,
               b.rowb(), b.colb(), N);
    } else {
      if (!(SK & 2048)) gemm(ctx, g2);
    }
  } else {
    Gemm g1 = mk(C, No, Co, B);                                  // T2t[b] = Wc . Y[b]^T   [C][No]
    g1.A = km(b.W(DGSCT_P_WC), Co);
    g1.B = km(Y, Co, (long)No * Co);
    outE(g1, b.S(s.T), E, Nop, (long)C * Nop);
    if (!(SK & 2048)) gemm(ctx, g1);
    Gemm g2 = mk(N, C, No, B);                                   // Yp[b] = Wn . T2[b]
    g2.A = km(b.W(DGSCT_P_WN), No);
    g2.B = km(b.S(s.T), Nop, (long)C * Nop);
    g2.r1_m = b.rowb(); g2.r1_n = b.colb(); g2.bias_n = b.colb2();
    outE(g2, Yp, E, C, (long)N * C);
    if (!(SK & 2048)) gemm(ctx, g2);
  }
  // F2 ---- latent tokens attend to the remapped tokens (one pass over Yp)   :572-580, :592
  void* tokpk = s.tokpk >= 0 ? b.S(s.tokpk) : nullptr;
  if (wide)
    tokattn_fwd_wide(ctx, Yp, b.F(DGSCT_P_TOKENS), prep_t0hi >= 0 ? (const void*)(b.prep + prep_t0hi) : (const void*)b.F(DGSCT_P_TOKENS),
                     prep_t0lo >= 0 ? b.prep + prep_t0lo : nullptr, B, N, C, tk, invN, b.S<float>(s.tok), b.S<float>(s.a), b.S(s.aE),
                     b.Wk<float>(wf.wL), b.S(s.P1));
  else
  if (!(SK & (128 | 16384))) tokattn_fwd(ctx, Yp, b.F(DGSCT_P_TOKENS), B, N, C, tk, b.S<float>(s.tok), b.S<float>(s.lse), b.S<float>(s.a), b.S(s.aE),
              b.Wk<float>(wf.tokscr), tokpk, prep_t0pk >= 0 ? b.prep + prep_t0pk : nullptr);
  {
    stream_fork(ctx);                                            // a = mean_N(Yp) is complete on the main stream
    Gemm g1 = mk(B, C, C);                                       // aq1 = relu(a Wa1^T + b)
    g1.A = km(b.S(s.aE), C); g1.B = km(b.W(DGSCT_P_WA1), C); g1.bias_n = b.F(DGSCT_P_BA1); g1.act = ACT_RELU;
    outE(g1, b.S(s.aq1), E, C);
    gemm(side, g1);
    Gemm g2 = mk(B, dd, C);                                      // aq2 = relu(a Wa2^T + b)
    g2.A = km(b.S(s.aE), C); g2.B = km(b.W(DGSCT_P_WA2), C); g2.bias_n = b.F(DGSCT_P_BA2); g2.act = ACT_RELU;
    outE(g2, b.S(s.aq2), E, dd);
    gemm(side, g2);
  }
  // F3 ---- X attends to the latent tokens (one pass over X)             :583-589
  if (wide)
    xattn_fwd_wide(ctx, X, b.S<float>(s.tok), b.F(DGSCT_P_GATE_AV), B, N, C, tk, b.S(s.X1), s.tokhi >= 0 ? b.S(s.tokhi) : nullptr,
                   wf.toklo >= 0 ? b.Wk(wf.toklo) : nullptr, b.Wk<float>(wf.wL), b.S(s.P2));
  else
  if (!(SK & (128 | 32768))) xattn_fwd(ctx, X, b.S<float>(s.tok), b.F(DGSCT_P_GATE_AV), B, N, C, tk, b.S(s.X1), tokpk);
  // F4-F6 ---- channel gate                                              :593-598
  {
    if (vq1_fused_supported(ctx.mode, N, C) && !fp8) {
      // stages 0 (C = 96 / 128): only mean_N vq1 exists -- one pass over X1, the [rows, C] tensor is never written (the backward
      // recomputes the ReLU decisions from X1: vq1_bwd)
      vq1sum_fwd(ctx, b.S(s.X1), b.W(DGSCT_P_WV1), b.F(DGSCT_P_BV1), B, N, C, invN, b.S<float>(s.mvq1),
                 vq1fuse_mode(-1) == 2 ? b.S(s.vq1) : nullptr);
    } else {
      bool vq1_summed = false;
      Gemm g3 = mk((int)R, C, C);                                // vq1 = relu(X1 Wv1^T + b)
      g3.A = km(b.S(s.X1), C); g3.B = km(b.W(DGSCT_P_WV1), C); g3.bias_n = b.F(DGSCT_P_BV1); g3.act = ACT_RELU;
      outE(g3, b.S(s.vq1), E, C);
      if (fp8) gemm_fp8(ctx, (int)R, C, C, b.S(s.X1), C, b.prep + prep_w8[1], (const float*)(b.prep + prep_w8scale) + 1, b.F(DGSCT_P_BV1), 1, b.S(s.vq1), C);
      else {
        // (round 5) the per-frame column sums (mean_N vq1) and positive counts in the product's epilogue: no second pass over vq1
        GemmFx fv; fv.epi = EPI_COLSUM; fv.rpf = N; fv.e_acc = b.S<float>(s.mvq1); fv.e_acc2 = b.S<float>(s.cnt1); fv.e_ld = C; fv.e_scale = invN;
        if ((gemmfx_mode(-1) & 64) && gemm_fx_supported(ctx, g3, fv)) { if (!(SK & 1024)) gemm_fx(ctx, g3, fv); vq1_summed = true; }
        else if (!(SK & 1024)) gemm(ctx, g3);
      }
      // mean_N vq1, and the number of positive entries per (frame, channel) for the backward (same pass over vq1)
      if (!vq1_summed && !(SK & 64)) colsum_batched_pos(ctx, b.S(s.vq1), C, (long)N * C, B, N, C, nullptr, 0, invN, b.S<float>(s.mvq1), C, b.S<float>(s.cnt1), C);
    }
    stream_join(ctx);                                            // aq1 / aq2 / a from the aux stream
    if (skinny_fused_supported(ctx, B, dd, C, 0)) {              // q = relu(m1 Wb^T + b), m1 = aq1 * mean_N vq1 made (and stored) on the way in
      SkFuse f; f.M = B; f.N = dd; f.K = C;
      f.a_mode = 1; f.A = b.S(s.aq1); f.lda = C; f.a_mul = b.S<float>(s.mvq1); f.ld_mul = C; f.a_store = b.S(s.m1); f.ld_store = C;
      f.B = b.W(DGSCT_P_WB); f.ldb = C; f.bias_n = b.F(DGSCT_P_BB); f.act = ACT_RELU;
      f.D = b.S(s.q); f.ddt = E; f.ldd = dd;
      skinny_fused(ctx, f);
    } else {
      ew(ctx, EW_MUL, b.S(s.m1), E, Earg(b.S(s.aq1), E), F32(b.S(s.mvq1)), NOARG, (long)B * C, 0.f, 1);
      Gemm g4 = mk(B, dd, C);                                    // q = relu(m1 Wb^T + b)
      g4.A = km(b.S(s.m1), C); g4.B = km(b.W(DGSCT_P_WB), C); g4.bias_n = b.F(DGSCT_P_BB); g4.act = ACT_RELU;
      outE(g4, b.S(s.q), E, dd);
      gemm(ctx, g4);
    }
    Gemm g5 = mk(B, C, dd);                                      // ch = sigmoid(q Wcatt^T + b)
    g5.A = km(b.S(s.q), dd); g5.B = km(b.W(DGSCT_P_WCATT), dd); g5.bias_n = b.F(DGSCT_P_BCATT); g5.act = ACT_SIGMOID;
    outF(g5, b.S<float>(s.ch), C);
    gemm(ctx, g5);
  }
  // F7 ---- spatial gate and the returned map                            :601-608
  // F8 ---- modulation + ln_before                                       :611-627
  // Early stages (C <= 256, bf16): F7 + F8 + the down-projection of F9 + the BN1 sums are ONE pass over X1 (fused_gate.hip).
  const bool gfuse = gate_fused_supported(ctx.mode, N, C, ds, g) && !fp8;
  const bool fuse89 = !gfuse && modln_gproj_supported(ctx.mode, C, ds, g);
  if (d.temporal) {                                              // tg depends on a only
    temporal_fwd(ctx, b.S<float>(s.a), b.F(DGSCT_P_WT), b.F(DGSCT_P_BT), B, C, b.S<float>(s.tg));
    if (tmap) ew(ctx, EW_COPY, tmap, DT_F32, F32(b.S(s.tg)), NOARG, NOARG, B, 0.f, 1);
  }
  if (gfuse) {
    // (with the fused backward nothing downstream reads Xc or vq2: it recomputes both from X1)
    const bool bfuse = gate_bwd_fused_supported(ctx.mode, N, C, ds, g);
    if (!bfuse && !xc_scratch) scale_cols(ctx, b.S(s.X1), b.S(s.Xc), B, N, C, b.S<float>(s.ch), 1.f);   // Xc = X1 * (1 + ch): backward's dWv2 operand
    gatemod_fwd(ctx, b.S(s.X1), b.S<float>(s.ch), b.S(s.aq2), b.F(DGSCT_P_WV2), b.F(DGSCT_P_BV2), b.F(DGSCT_P_WS), b.F(DGSCT_P_BS),
                d.temporal ? b.S<float>(s.tg) : nullptr, d.alpha, d.beta, d.gamma, d.ln_before ? b.F(DGSCT_P_LNB_W) : nullptr,
                d.ln_before ? b.F(DGSCT_P_LNB_B) : nullptr, d.eps, B, N, C, ds, g, b.F(DGSCT_P_WD), b.S<float>(s.sl), b.S(s.X3),
                b.S<float>(s.mu_b), b.S<float>(s.rstd_b), b.S(s.Zp), d.use_bn && d.training ? b.S<float>(s.bnacc1) : nullptr,
                (!bfuse || gatefuse_mode(-1) == 2) ? b.S(s.vq2) : nullptr);
    spatial_fwd(ctx, b.S<float>(s.sl), B, N, b.S<float>(s.sg), b.S<float>(s.map), map);   // saved copy + the returned map
  } else {
    void* Xcf = xc_scratch ? b.Wk(wf.Xc) : b.S(s.Xc);
    if (!(SK & 16)) scale_cols(ctx, b.S(s.X1), Xcf, B, N, C, b.S<float>(s.ch), 1.f);              // Xc = X1 * (1 + ch)
    Gemm g1 = mk((int)R, dd, C);                                 // vq2 = relu(Xc Wv2^T + b)
    g1.A = km(Xcf, C); g1.B = km(b.W(DGSCT_P_WV2), C); g1.bias_n = b.F(DGSCT_P_BV2); g1.act = ACT_RELU;
    outE(g1, b.S(s.vq2), E, dd);
    if (fp8) gemm_fp8(ctx, (int)R, dd, C, Xcf, C, b.prep + prep_w8[2], (const float*)(b.prep + prep_w8scale) + 2, b.F(DGSCT_P_BV2), 1, b.S(s.vq2), dd);
    else if (!(SK & 1024)) gemm(ctx, g1);
    if (!(SK & 16)) rowdot_batched(ctx, b.S(s.vq2), dd, (long)N * dd, B, N, dd, b.S(s.aq2), E, dd, b.F(DGSCT_P_WS), b.F(DGSCT_P_BS),
                   b.S<float>(s.sl));
    spatial_fwd(ctx, b.S<float>(s.sl), B, N, b.S<float>(s.sg), b.S<float>(s.map), map);   // saved copy + the returned map
    // (stages 0-1 without the gate fusion: modulation + ln_before + down-projection + BN1 sums in one pass, see modln_gproj)
    if (fuse89)
      modln_gproj(ctx, b.S(s.X1), b.S<float>(s.ch), b.S<float>(s.sg), d.temporal ? b.S<float>(s.tg) : nullptr, d.alpha, d.beta, d.gamma,
                  d.ln_before ? b.F(DGSCT_P_LNB_W) : nullptr, d.ln_before ? b.F(DGSCT_P_LNB_B) : nullptr, d.eps, B, N, C, ds, g,
                  b.F(DGSCT_P_WD), (long)(ds / g) * (C / g), C / g, 1, b.S(s.X3), b.S<float>(s.mu_b), b.S<float>(s.rstd_b), b.S(s.Zp),
                  d.use_bn && d.training ? b.S<float>(s.bnacc1) : nullptr);
    else if (!(SK & 256))
      modln_fwd(ctx, b.S(s.X1), b.S<float>(s.ch), b.S<float>(s.sg), d.temporal ? b.S<float>(s.tg) : nullptr, d.alpha, d.beta,
                d.gamma, d.ln_before ? b.F(DGSCT_P_LNB_W) : nullptr, d.ln_before ? b.F(DGSCT_P_LNB_B) : nullptr, d.eps, B, N, C,
                b.S(s.X3), b.S<float>(s.mu_b), b.S<float>(s.rstd_b));
  }
  // F9-F10 ---- grouped bottleneck + BatchNorm                           :629-643
  float* bn1 = b.S<float>(s.bn1);
  float* bn2 = b.S<float>(s.bn2);
  {
    // dg = ds/g <= 8 (stages 0-1 of every backbone): the grouped projections are HBM streams with 6..8-wide GEMM
    // dimensions -> vector-unit row kernels (prims_proj.hip) instead of 80 %-padded MFMA tiles
    const bool vproj = gproj_supported(ctx.mode, C, ds, g);
    const long cgl = C / g, dgl = ds / g;
    bool stats1_done = false, stats2_done = false;
    if (fuse89 || gfuse) {
    } else if (vproj) {
      gproj_narrow(ctx, b.S(s.X3), R, C, ds, g, b.F(DGSCT_P_WD), dgl * cgl, cgl, 1, b.S(s.Zp));     // Zp = X3 (x)_g Wd
    } else {
      Gemm g1 = mk((int)R, ds / g, C / g, g);                    // Zp = X3 (x)_g Wd
      g1.A = km(b.S(s.X3), C, C / g);
      g1.B = km(b.W(DGSCT_P_WD), C / g, (long)(ds / g) * (C / g));
      outE(g1, b.S(s.Zp), E, ds, ds / g);
      GemmFx f1s; f1s.epi = EPI_COLSTATS; f1s.e_acc = b.S<float>(s.bnacc1) + ds; f1s.e_acc2 = b.S<float>(s.bnacc1) + 2 * ds;
      if (d.use_bn && d.training && (gemmfx_mode(-1) & 128) && gemm_fx_supported(ctx, g1, f1s)) {       // BN1 sums in the epilogue
        if (!(SK & 4096)) gemm_fx(ctx, g1, f1s);
        stats1_done = true;
      } else if (!(SK & 4096)) gemm(ctx, g1);
    }
    if (d.use_bn) {                                              // BN1: finalised inside the pass that applies it (BnFin)
      if (d.training && !fuse89 && !gfuse && !stats1_done && !(SK & 32)) bn_stats(ctx, b.S(s.Zp), R, ds, b.S<float>(s.bnacc1));
      const BnFin f1{b.S<float>(s.bnacc1), R, b.F(DGSCT_P_BN1_W), b.F(DGSCT_P_BN1_B), b.Fm(DGSCT_P_BN1_RM), b.Fm(DGSCT_P_BN1_RV),
                     d.bn_momentum, d.eps, d.training, bn1, bn1 + ds, bn1 + 2 * ds, bn1 + 3 * ds};
      if (!(SK & 32)) affine_act_bn(ctx, b.S(s.Zp), b.S(s.Z), R, ds, f1, 1);
    } else {
      affine_act(ctx, b.S(s.Zp), b.S(s.Z), R, ds, nullptr, nullptr, 1);
    }
    const bool stats2 = d.use_bn && d.training;
    if (vproj) {                                                 // Op = Z (x)_g Wu, BN2 sums in the same pass
      gproj_wide(ctx, b.S(s.Z), R, C, ds, g, b.F(DGSCT_P_WU), cgl * dgl, 1, dgl, b.S(s.Op), stats2 ? b.S<float>(s.bnacc2) : nullptr);
    } else {
      Gemm g2 = mk((int)R, C / g, ds / g, g);                    // Op = Z (x)_g Wu
      g2.A = km(b.S(s.Z), ds, ds / g);
      g2.B = km(b.W(DGSCT_P_WU), ds / g, (long)(C / g) * (ds / g));
      outE(g2, b.S(s.Op), E, C, C / g);
      GemmFx f2s; f2s.epi = EPI_COLSTATS; f2s.e_acc = b.S<float>(s.bnacc2) + C; f2s.e_acc2 = b.S<float>(s.bnacc2) + 2 * C;
      if (stats2 && (gemmfx_mode(-1) & 128) && gemm_fx_supported(ctx, g2, f2s)) {                       // BN2 sums in the epilogue
        if (!(SK & 4096)) gemm_fx(ctx, g2, f2s);
        stats2_done = true;
      } else if (!(SK & 4096)) gemm(ctx, g2);
    }
    if (d.use_bn && stats2 && !vproj && !stats2_done && !(SK & 32)) bn_stats(ctx, b.S(s.Op), R, C, b.S<float>(s.bnacc2));
  }
  // F11 ---- BN2 finalised + applied, ln_post / gate                     :668-671
  const BnFin f2{b.S<float>(s.bnacc2), R, b.F(DGSCT_P_BN2_W), b.F(DGSCT_P_BN2_B), b.Fm(DGSCT_P_BN2_RM), b.Fm(DGSCT_P_BN2_RV),
                 d.bn_momentum, d.eps, d.training, bn2, bn2 + C, bn2 + 2 * C, bn2 + 3 * C};
  if (!(SK & 512)) tail_fwd(ctx, b.S(s.Op), d.use_bn ? bn2 + 2 * C : nullptr, d.use_bn ? bn2 + 3 * C : nullptr,
           d.ln_post ? b.F(DGSCT_P_LNP_W) : nullptr, d.ln_post ? b.F(DGSCT_P_LNP_B) : nullptr,
           d.use_gate ? b.F(DGSCT_P_GATE) : nullptr, d.gate_before_ln_post, d.eps, R, C, out, b.S<float>(s.mu_p),
           b.S<float>(s.rstd_p), residual,         // f2: out = residual + adapter(X, Y)
           d.use_bn ? &f2 : nullptr);
  check_async("dgsct_adapter_forward");
  return has_error() ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
int Plan::backward(float* const* params, const void* prep, const void* X, const void* Y, const void* saved_c,
                   const void* dOut, const float* dMap, const float* dTmap, void* dX, void* dY, float* grads, void* ws,
                   void* stream, void* aux_stream, bool skip_into_dx, bool no_join, const BwdPair* pair) const {
  Bound b(*this, params, prep, const_cast<void*>(saved_c), ws, stream);
  b.ctx.aux = aux_stream;
  const Ctx& ctx = b.ctx;
  // The product that writes dY is the last link of the data-gradient chain and reads only the workspace (dT) and the prepared weights.
  // A caller that runs the two adapters of a position (net_trans.py:891-892) side by side wants d f_own = dX(own) + dY(other): it holds
  // this product back (phase 1), and issues it afterwards (phase 2) with the OTHER call's dX as the epilogue's residual -- behind an
  // event that call recorded right after its dX.  The gradient-accumulation pass over [BT][N][C] per adapter call disappears.
  const int phase = pair ? pair->phase : 0;
  // (DGSCT_BWD_NO_JOIN) the caller orders `grads` / `ws` behind the aux stream itself: this call's chain ends with dY, so everything
  // that only feeds parameter gradients -- also the three bias-side reductions that used to balance the tail of a joined call -- goes
  // to the aux stream
  const bool deferred = no_join && aux_stream;
  static const bool bias_on_main = getenv("DGSCT_BIAS_ON_MAIN") && atoi(getenv("DGSCT_BIAS_ON_MAIN"));     // (A/B switch)
  const bool bias_aux = deferred && !bias_on_main;
  auto dy_product = [&]() {
    Gemm g3;
    if (orderA) {
      g3 = mk(No, Co, N, B);                                     // dY[b] = Wn^T . dT1[b]
      g3.A = mn(b.W(DGSCT_P_WN), No);
      g3.B = mn(b.Wk(wb.dT), Co, (long)N * Co);
    } else {
      g3 = mk(No, Co, C, B);                                     // dY[b] = dT2[b] . Wc
      g3.A = km(b.Wk(wb.dT), C, (long)No * C);
      g3.B = b.WB(DGSCT_P_WC, Co, C);
    }
    outE(g3, dY, E, Co, (long)No * Co);
    if (pair && pair->dy_residual) resid(g3, pair->dy_residual, E, Co, (long)No * Co);
    if (pair && pair->dy_wait) event_wait(ctx, pair->dy_wait);
    gemm(ctx, g3);
  };
  if (phase == 2) {
    const int sk2 = (C >= g_skip_minc.load(std::memory_order_relaxed) && C <= g_skip_maxc.load(std::memory_order_relaxed)) ? g_skip.load(std::memory_order_relaxed) : 0;
    if (!(sk2 & 2048)) dy_product();                          // (what-if switch: the remap products)
    check_async("dgsct_adapter_backward (dY product)");
    return has_error() ? 1 : 0;
  }
  // Weight / bias gradients feed nothing downstream: they go to the aux stream (when given) and overlap the data-gradient
  // chain.  A fork makes aux wait for everything enqueued so far (the operands); the final join orders it all
  // before the caller's next use of `grads` / `ws`.
  Ctx side = ctx;
  if (aux_stream) { side.stream = aux_stream; side.aux = nullptr; }
  // A fork costs the MAIN stream ~6.5 us (the event record is a barrier packet in its queue), so gradient-only work is
  // queued with defer() and released in groups: side_flush() records one event and enqueues everything pending.  Deferred
  // work only reads buffers nothing overwrites before the final join (no workspace aliasing in the backward layout).
  std::vector<std::function<void()>> pend;
  auto defer = [&](std::function<void()> f) { pend.push_back(std::move(f)); };
  auto side_flush = [&]() {
    if (pend.empty()) return;
    stream_fork(ctx);
    for (auto& f : pend) f();
    pend.clear();
  };
  auto G = [&](int id) -> float* { return grad_off[id] >= 0 ? grads + grad_off[id] : nullptr; };
  zero2(ctx, b.Wk(0), (size_t)wb.zero_end, grads, (size_t)grad_floats * 4);
  const float* bn1 = b.S<float>(s.bn1);
  const float* bn2 = b.S<float>(s.bn2);
  const float* tg = d.temporal ? b.S<float>(s.tg) : nullptr;
  const int cg = C / g, dg = ds / g;
  const float invN = 1.f / (float)N;
  const bool vproj = gproj_supported(ctx.mode, C, ds, g);
  const int SK = (C >= g_skip_minc.load(std::memory_order_relaxed) && C <= g_skip_maxc.load(std::memory_order_relaxed)) ? g_skip.load(std::memory_order_relaxed) : 0;   // what-if switches
  // (weight-gradient products on the aux stream: every one of them goes through wgemm)
  auto wgemm = [SK](const Ctx& c, const Gemm& gg) { if (!(SK & 1)) gemm(c, gg); };

  // B11 ---- ln_post / gate, BN2 sums
  void* dO = b.Wk(wb.dO);
  if (!(SK & 512)) tail_bwd(ctx, dOut, b.S(s.Op), d.use_bn ? bn2 + 2 * C : nullptr, d.use_bn ? bn2 + 3 * C : nullptr, bn2, bn2 + C,
           d.ln_post ? b.F(DGSCT_P_LNP_W) : nullptr, d.ln_post ? b.F(DGSCT_P_LNP_B) : nullptr,
           d.use_gate ? b.F(DGSCT_P_GATE) : nullptr, d.gate_before_ln_post, b.S<float>(s.mu_p), b.S<float>(s.rstd_p), R, C,
           dO, G(DGSCT_P_LNP_W), G(DGSCT_P_LNP_B), G(DGSCT_P_GATE), d.use_bn ? G(DGSCT_P_BN2_B) : nullptr,
           d.eps, b.Wk<float>(wb.rowpart), row_part_floats(B, C));
  // B10 ---- BN2 backward, up projection
  const bool bnb = d.use_bn && vproj;                            // BN2 backward inside the narrow projection's pass (stages 0-1)
  // late stages (round 5): dOp = BN2 backward of dO is formed while the dZ product stages its A operand (gemm_fx.hip) and written
  // back in place for the dWu product -- no bn_bwd_apply pass over [rows, C]
  Gemm gdz = mk((int)R, dg, cg, g);                              // dZ = dOp (x)_g Wu
  GemmFx fdz;
  bool fx_dz = false;
  if (d.use_bn && !bnb && !vproj) {
    gdz.A = km(dO, C, cg);
    gdz.B = b.WBg(DGSCT_P_WU, dg, C, cg, (long)cg * dg);
    outE(gdz, b.Wk(wb.dZ), E, ds, dg);
    fdz.a_pro = APRO_BNBWD; fdz.a2 = b.S(s.Op); fdz.bn_mean = bn2; fdz.bn_rstd = bn2 + C; fdz.bn_sc = bn2 + 2 * C; fdz.bn_sh = bn2 + 3 * C;
    fdz.bn_sums = G(DGSCT_P_BN2_B); fdz.bn_rows = R; fdz.bn_C = C; fdz.bn_relu = 0; fdz.bn_training = d.training; fdz.a_store = dO;
    fx_dz = (gemmfx_mode(-1) & 1) && gemm_fx_supported(ctx, gdz, fdz);
  }
  if (d.use_bn && !bnb && !fx_dz && !(SK & 8)) {
    bn_bwd_apply(ctx, dO, b.S(s.Op), dO, R, C, bn2, bn2 + C, bn2 + 2 * C, bn2 + 3 * C, G(DGSCT_P_BN2_B), 0, 1, d.training);
  }
  {
    Gemm g1 = mk(cg, dg, (int)R, g);                             // dWu = dOp^T (x)_g Z
    g1.A = mn(dO, C, cg);
    g1.B = mn(b.S(s.Z), ds, dg);
    outF(g1, G(DGSCT_P_WU), dg, (long)cg * dg);
    atomic_out(g1);
    defer([=, &side] { wgemm(side, g1); });                       // released with dWd
    if (bnb) {                                                   // dOp = BN2 backward of dO (in place), dZ = dOp (x)_g Wu
      gproj_narrow_bnb(ctx, dO, b.S(s.Op), dO, R, C, ds, g, b.F(DGSCT_P_WU), (long)cg * dg, 1, dg, b.Wk(wb.dZ), bn2, bn2 + C, bn2 + 2 * C,
                       bn2 + 3 * C, G(DGSCT_P_BN2_B), d.training);
    } else if (vproj) {                                          // dZ = dOp (x)_g Wu
      gproj_narrow(ctx, dO, R, C, ds, g, b.F(DGSCT_P_WU), (long)cg * dg, 1, dg, b.Wk(wb.dZ));
    } else if (fx_dz) {
      if (!(SK & 4096)) gemm_fx(ctx, gdz, fdz);
    } else {
      Gemm g2 = mk((int)R, dg, cg, g);                           // dZ = dOp (x)_g Wu
      g2.A = km(dO, C, cg);
      g2.B = mn(b.W(DGSCT_P_WU), dg, (long)cg * dg);
      outE(g2, b.Wk(wb.dZ), E, ds, dg);
      if (!(SK & 4096)) gemm(ctx, g2);
    }
  }
  // B9 ---- relu, BN1 backward, down projection
  void* dZ = b.Wk(wb.dZ);
  void* dX1 = b.Wk(wb.dX1);
  const bool bfuse = gate_bwd_fused_supported(ctx.mode, N, C, ds, g) && !fp8;
  if (d.use_bn)
    bn_bwd_stats(ctx, dZ, b.S(s.Zp), R, ds, bn1, bn1 + ds, bn1 + 2 * ds, bn1 + 3 * ds, 1, G(DGSCT_P_BN1_B), b.Wk<float>(wb.rowpart),
                 row_part_floats(B, C));
  bool t3_valid = false, t1_valid = false;                       // (fused engine) bias gradients of the two query layers as [BT, C] column sums
  Gemm gwd = mk(dg, cg, (int)R, g);                              // dWd = dZp^T (x)_g X3
  gwd.A = mn(dZ, ds, dg);
  gwd.B = mn(b.S(s.X3), C, cg);
  outF(gwd, G(DGSCT_P_WD), cg, (long)dg * cg);
  atomic_out(gwd);
  Gemm gwv2 = mk(dd, C, (int)R);                                 // dWv2 = dvq2^T . Xc
  gwv2.A = mn(b.S(s.vq2), dd);
  void* Xcb = xc_scratch ? b.Wk(wb.Xc) : b.S(s.Xc);
  gwv2.B = mn(Xcb, C);
  outF(gwv2, G(DGSCT_P_WV2), C);
  atomic_out(gwv2);
  if (bfuse) {
    // B9 (BN1 apply, dX3) + B8 + B7 in ONE pass over X1 (fused_gate.hip): dZp in place, dX1, dvq2 (over the vq2 region), Xc
    gatemod_bwd(ctx, b.S(s.X1), b.S<float>(s.ch), b.S(s.aq2), b.F(DGSCT_P_WV2), b.F(DGSCT_P_BV2), b.F(DGSCT_P_WS), tg, d.alpha, d.beta,
                d.gamma, d.ln_before ? b.F(DGSCT_P_LNB_W) : nullptr, b.S<float>(s.mu_b), b.S<float>(s.rstd_b), b.S<float>(s.sl),
                b.S<float>(s.sg), b.S<float>(s.map), dMap, B, N, C, ds, g, b.F(DGSCT_P_WD), dZ, b.S(s.Zp), bn1, bn1 + ds, bn1 + 2 * ds,
                bn1 + 3 * ds, d.use_bn ? G(DGSCT_P_BN1_B) : nullptr, d.use_bn, d.training, dX1, b.S(s.vq2), Xcb,
                b.Wk<float>(wb.dch), b.Wk<float>(wb.u), b.Wk<float>(wb.dtg), G(DGSCT_P_LNB_W), G(DGSCT_P_LNB_B), G(DGSCT_P_BV2),
                G(DGSCT_P_BS), b.Wk<float>(wb.rowtmp), b.Wk<float>(wb.rowpart), row_part_floats(B, C));
    defer([=, &side] { wgemm(side, gwd); });
    defer([=, &side] { wgemm(side, gwv2); });
    side_flush();
    // tmpBd = u * aq2 (d ws = sum_b tmpBd: with the bias gradients below, colsum_multi);  dpa2 = u * ws * (aq2 > 0)
    ew2(ctx, EwCall{EW_MUL, b.Wk(wb.tmpBd), DT_F32, F32(b.Wk(wb.u)), Earg(b.S(s.aq2), E), NOARG, (long)B * dd, 0.f, 1},
        EwCall{EW_MULB_MASK, b.Wk(wb.dpa2), E, F32(b.Wk(wb.u)), F32(b.F(DGSCT_P_WS)), Earg(b.S(s.aq2), E), (long)B * dd, 0.f, dd});
  } else {
  if (xc_scratch) scale_cols(ctx, b.S(s.X1), Xcb, B, N, C, b.S<float>(s.ch), 1.f);   // (unfused test path of a fused shape: Xc is not saved)
  Gemm gdx3 = mk((int)R, cg, dg, g);                             // dX3 = dZp (x)_g Wd
  GemmFx fdx3;
  bool fx_dx3 = false;
  if (d.use_bn && !vproj) {                                      // (round 5) dZp = relu' + BN1 backward of dZ formed in the product's staging
    gdx3.A = km(dZ, ds, dg);
    gdx3.B = b.WBg(DGSCT_P_WD, cg, ds, dg, (long)dg * cg);
    outE(gdx3, b.Wk(wb.dX3), E, C, cg);
    fdx3.a_pro = APRO_BNBWD; fdx3.a2 = b.S(s.Zp); fdx3.bn_mean = bn1; fdx3.bn_rstd = bn1 + ds; fdx3.bn_sc = bn1 + 2 * ds; fdx3.bn_sh = bn1 + 3 * ds;
    fdx3.bn_sums = G(DGSCT_P_BN1_B); fdx3.bn_rows = R; fdx3.bn_C = ds; fdx3.bn_relu = 1; fdx3.bn_training = d.training; fdx3.a_store = b.Wk(wb.dZp);
    fx_dx3 = (gemmfx_mode(-1) & 2) && gemm_fx_supported(ctx, gdx3, fdx3);
  }
  if (fx_dx3) {
  } else if (SK & 8) {
  } else if (d.use_bn) {
    bn_bwd_apply(ctx, dZ, b.S(s.Zp), dZ, R, ds, bn1, bn1 + ds, bn1 + 2 * ds, bn1 + 3 * ds, G(DGSCT_P_BN1_B), 1, 1, d.training);
  } else {
    bn_bwd_apply(ctx, dZ, b.S(s.Zp), dZ, R, ds, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, 0);
  }
  {
    if (fx_dx3) {
      if (!(SK & 4096)) gemm_fx(ctx, gdx3, fdx3);
      gwd.A = mn(b.Wk(wb.dZp), ds, dg);                          // (dWd reads the cotangent the product wrote beside dZ)
    }
    defer([=, &side] { wgemm(side, gwd); });
    side_flush();
    if (fx_dx3) {
    } else if (vproj) {                                                 // dX3 = dZp (x)_g Wd
      gproj_wide(ctx, dZ, R, C, ds, g, b.F(DGSCT_P_WD), (long)dg * cg, cg, 1, b.Wk(wb.dX3), nullptr);
    } else {
      Gemm g2 = mk((int)R, cg, dg, g);                           // dX3 = dZp (x)_g Wd
      g2.A = km(dZ, ds, dg);
      g2.B = mn(b.W(DGSCT_P_WD), cg, (long)dg * cg);
      outE(g2, b.Wk(wb.dX3), E, C, cg);
      if (!(SK & 4096)) gemm(ctx, g2);
    }
  }
  // B8 ---- ln_before, modulation
  if (!(SK & 256)) modln_bwd(ctx, b.Wk(wb.dX3), b.S(s.X1), b.S<float>(s.ch), b.S<float>(s.sg), tg, d.alpha, d.beta, d.gamma,
            d.ln_before ? b.F(DGSCT_P_LNB_W) : nullptr, b.S<float>(s.mu_b), b.S<float>(s.rstd_b), B, N, C, dX1,
            G(DGSCT_P_LNB_W), G(DGSCT_P_LNB_B), b.Wk<float>(wb.dch), b.Wk<float>(wb.dsg), b.Wk<float>(wb.dtg),
            b.Wk<float>(wb.rowpart), row_part_floats(B, C));
  // B7 ---- spatial gate
  {
    spatial_bwd(ctx, b.S<float>(s.sl), b.S<float>(s.sg), b.S<float>(s.map), b.Wk<float>(wb.dsg), dMap, B, N,
                b.Wk<float>(wb.dsl), G(DGSCT_P_BS));
    // (round 5) dXc = dvq2 . Wv2 on the fused engine (gemm_fx.hip).  The channel-gate backward (dX1 += dXc (1 + ch), dch += sum_n dXc X1)
    // runs in the product's epilogue wherever the engine takes the call: no dXc tensor, no xc_bwd pass.  The ReLU backward that makes
    // dvq2 is folded into the staging of the A operand (vq2) only where few column tiles re-read (and re-transform) it -- C <= 256:
    // measured inside the step (tools/call_overlap.py AB=gemmfx=..), the prologue wins 48 us per stage-1 pair, is neutral at C = 384 /
    // 512 and loses 13 us per pair at C = 768 / 1024 (eight column tiles each redo the transform); "gemmfx" bit 16 forces it everywhere.
    Gemm gxc = mk((int)R, C, dd);
    gxc.B = b.WB(DGSCT_P_WV2, C, dd);
    resid(gxc, dX1, E, C);
    outE(gxc, dX1, E, C);
    GemmFx fxc;
    fxc.epi = EPI_XCBWD; fxc.rpf = N;
    fxc.e_cs = b.S<float>(s.ch); fxc.e_x = b.S(s.X1); fxc.e_acc = b.Wk<float>(wb.dch); fxc.e_ld = C;
    const int fxm = gemmfx_mode(-1);
    int fx_xc = 0;                                               // 0: separate launches, 1: epilogue only, 2: prologue + epilogue
    if (fxm & 4) {
      gxc.A = km(b.S(s.vq2), dd);
      if (gemm_fx_supported(ctx, gxc, fxc)) fx_xc = 1;
      if ((C <= 256 || (fxm & 16)) && !(fxm & 32)) {            // (bit 32: never -- tests reach the epilogue-only variant at small widths)
        GemmFx f2 = fxc;
        f2.a_pro = APRO_MASKSCALE; f2.a_rs = b.Wk<float>(wb.dsl); f2.a_cs = b.S(s.aq2); f2.a_cs_dt = E; f2.a_cs_ld = dd; f2.a_cs2 = b.F(DGSCT_P_WS);
        f2.a_scale = 1.f; f2.a_store = b.Wk(wb.dvq2);
        if (gemm_fx_supported(ctx, gxc, f2)) { fx_xc = 2; fxc = f2; }
      }
    }
    if (fx_xc == 2) {                                            // u, and w2 = sum_n dsl (vq2 > 0) for d bias(vq2), in the same pass over vq2
      if (!(SK & 64)) colsum_batched_pos(ctx, b.S(s.vq2), dd, (long)N * dd, B, N, dd, b.Wk<float>(wb.dsl), N, 1.f, b.Wk<float>(wb.u), dd,
                                         b.Wk<float>(wb.w2), dd);
    } else
    if (!(SK & 64)) colsum_batched(ctx, b.S(s.vq2), dd, (long)N * dd, B, N, dd, b.Wk<float>(wb.dsl), N, 1.f, b.Wk<float>(wb.u), dd);  // u
    // tmpBd = u * aq2 (d ws = sum_b tmpBd: with the bias gradients below, colsum_multi);  dpa2 = u * ws * (aq2 > 0)
    ew2(ctx, EwCall{EW_MUL, b.Wk(wb.tmpBd), DT_F32, F32(b.Wk(wb.u)), Earg(b.S(s.aq2), E), NOARG, (long)B * dd, 0.f, 1},
        EwCall{EW_MULB_MASK, b.Wk(wb.dpa2), E, F32(b.Wk(wb.u)), F32(b.F(DGSCT_P_WS)), Earg(b.S(s.aq2), E), (long)B * dd, 0.f, dd});
    if (fx_xc == 2) {
      if (!(SK & 1024)) gemm_fx(ctx, gxc, fxc);
      gwv2.A = mn(b.Wk(wb.dvq2), dd);
      defer([=, &side, &b] {
        wgemm(side, gwv2);
        // d bias(vq2)[j] = sum_b w2[b][j] aq2[b][j] ws[j]
        ew(side, EW_MUL3B, b.Wk(wb.t3), DT_F32, F32(b.Wk(wb.w2)), Earg(b.S(s.aq2), E), F32(b.F(DGSCT_P_WS)), (long)B * dd, 0.f, dd);
      });
      side_flush();
    } else {
    // dvq2 (in place over vq2) = dsl[b,n] * aq2[b,j]*ws[j] * (vq2 > 0)
    {
      PartJob pj;                                                // d bias(vq2): second stage of the column sums off the chain
      Ctx cl = ctx; cl.late = aux_stream ? &pj : nullptr;
      if (!(SK & 2)) relu_bwd_scale(cl, b.S(s.vq2), b.S(s.vq2), B, N, dd, b.Wk<float>(wb.dsl), b.S(s.aq2), E, b.F(DGSCT_P_WS), 1.f,
                     G(DGSCT_P_BV2), b.Wk<float>(wb.rowpart_v2), row_part_floats(B, C));
      if (pj.n) defer([=, &side] { part_reduce_run(side.stream, pj); });
    }
    if (fx_xc == 1) {                                            // dX1 += (dvq2 . Wv2) (1 + ch), dch += ...: the product's epilogue
      if (!(SK & 1024)) gemm_fx(ctx, gxc, fxc);
      defer([=, &side] { wgemm(side, gwv2); });
      side_flush();
    } else {
    Gemm g1 = mk((int)R, C, dd);                                 // dXc = dvq2 . Wv2
    g1.A = km(b.S(s.vq2), dd);
    g1.B = b.WB(DGSCT_P_WV2, C, dd);
    outE(g1, b.Wk(wb.dXc), E, C);
    if (!(SK & 1024)) gemm(ctx, g1);
    defer([=, &side] { wgemm(side, gwv2); });
    side_flush();
    if (!(SK & 4)) xc_bwd(ctx, b.Wk(wb.dXc), b.S(s.X1), dX1, B, N, C, b.S<float>(s.ch), b.Wk<float>(wb.dch));
    }
    }
    t3_valid = fx_xc == 2;
  }
  }
  // B6 ---- channel-gate head
  bool wgbt = false;
  {
    const MatOp wcattT = b.WB(DGSCT_P_WCATT, dd, C), wbT = b.WB(DGSCT_P_WB, C, dd);
    // the chain's four launches (sigmoid', dq, dm1, its two consumers) as two products with the elementwise parts folded in
    const bool skf = skinny_fused_supported(ctx, B, dd, C, 0) && skinny_fused_supported(ctx, B, C, dd, 0) &&
                     (E != DT_BF16 || (wcattT.kmajor && wbT.kmajor));
    if (!skf) ew(ctx, EW_SIGMOID_BWD, b.Wk(wb.dpre_c), E, F32(b.Wk(wb.dch)), F32(b.S(s.ch)), NOARG, (long)B * C, 0.f, 1);
    Gemm g1 = mk(C, dd, B);                                      // dWcatt = dpre^T . q
    g1.A = mn(b.Wk(wb.dpre_c), C); g1.B = mn(b.S(s.q), dd);
    outF(g1, G(DGSCT_P_WCATT), dd);
    // (round 5) the four weight gradients that contract over the frames only -- dWcatt, dWb here, dWa1, dWa2 in B4 -- as ONE launch
    // (gemm_wgbt.hip) instead of four 2-3-workgroup launches of the tiled engine
    WgBtJob wj[WGBT_MAX] = {{b.Wk(wb.dpre_c), b.S(s.q), G(DGSCT_P_WCATT), C, dd, B, C, dd, dd},
                            {b.Wk(wb.dq), b.S(s.m1), G(DGSCT_P_WB), dd, C, B, dd, C, C},
                            {b.Wk(wb.dpa1), b.S(s.aE), G(DGSCT_P_WA1), C, C, B, C, C, C},
                            {b.Wk(wb.dpa2), b.S(s.aE), G(DGSCT_P_WA2), dd, C, B, dd, C, C}};
    wgbt = wgrad_bt_supported(side, wj, 4);
    if (!wgbt) defer([=, &side] { wgemm(side, g1); });
    if (skf) {                                                   // dq = (dpre . Wcatt) * (q > 0), dpre = dch ch (1 - ch) made (and stored) on the way in
      SkFuse f; f.M = B; f.N = dd; f.K = C;
      f.a_mode = 2; f.A = b.Wk(wb.dch); f.lda = C; f.a_mul = b.S<float>(s.ch); f.ld_mul = C; f.a_store = b.Wk(wb.dpre_c); f.ld_store = C;
      f.B = wcattT.p; f.ldb = wcattT.ld; f.b_kmajor = wcattT.kmajor;
      f.mask = b.S(s.q); f.ldmask = dd;
      f.D = b.Wk(wb.dq); f.ddt = E; f.ldd = dd;
      skinny_fused(ctx, f);
    } else {
      Gemm g2 = mk(B, dd, C);                                    // dq = (dpre . Wcatt) * (q > 0)
      g2.A = km(b.Wk(wb.dpre_c), C); g2.B = wcattT;
      g2.mask = b.S(s.q); g2.ldmask = dd;
      outE(g2, b.Wk(wb.dq), E, dd);
      gemm(ctx, g2);
    }
    Gemm g3 = mk(dd, C, B);                                      // dWb = dq^T . m1
    g3.A = mn(b.Wk(wb.dq), dd); g3.B = mn(b.S(s.m1), C);
    outF(g3, G(DGSCT_P_WB), C);
    if (!wgbt) defer([=, &side] { wgemm(side, g3); });
    else defer([=, &side] { if (!(SK & 1)) wgrad_bt(side, wj, 4); });      // (its operands dpa1 / dpa2 exist by the flush in B4)
    if (skf) {                                                   // dm1 = dq . Wb -> dpa1 = dm1 mvq1 (aq1 > 0), coef = dm1 aq1 from the epilogue
      SkFuse f; f.M = B; f.N = C; f.K = dd;
      f.A = b.Wk(wb.dq); f.lda = dd; f.B = wbT.p; f.ldb = wbT.ld; f.b_kmajor = wbT.kmajor;
      f.epi = 1; f.D = b.Wk(wb.dpa1); f.ddt = E; f.ldd = C; f.e_mul = b.S<float>(s.mvq1); f.ld_emul = C; f.e_q = b.S(s.aq1); f.ld_eq = C;
      f.D2 = b.Wk<float>(wb.coef); f.ldd2 = C;
      skinny_fused(ctx, f);
    } else {
      Gemm g4 = mk(B, C, dd);                                    // dm1 = dq . Wb
      g4.A = km(b.Wk(wb.dq), dd); g4.B = wbT;
      outF(g4, b.Wk<float>(wb.dm1), C);
      gemm(ctx, g4);
      ew2(ctx, EwCall{EW_MUL_MASK, b.Wk(wb.dpa1), E, F32(b.Wk(wb.dm1)), F32(b.S(s.mvq1)), Earg(b.S(s.aq1), E), (long)B * C, 0.f, 1},
          EwCall{EW_MUL, b.Wk(wb.coef), DT_F32, F32(b.Wk(wb.dm1)), Earg(b.S(s.aq1), E), NOARG, (long)B * C, 0.f, 1});
    }
  }
  // B5 ---- video query 1
  bool vq1_in_kernel_dw = false, fx_x1 = false;
  {
    if (vq1_fused_supported(ctx.mode, N, C) && !fp8) {
      // the forward kept no vq1: ReLU decisions recomputed from X1, dvq1 (into vq1's region, for dWv1 below), d bias, dX1 += dvq1 . Wv1
      PartJob pj;
      Ctx cl = ctx; cl.late = aux_stream ? &pj : nullptr;
      // "vq1fuse" = 3: dWv1 accumulated inside vq1_bwd (no dvq1 tensor, no product on the aux stream).  Measured (tools/call_overlap.py,
      // AB=vq1fuse=3): the stage-0 pair's backward 3340 -> 3417 us -- C x C accumulators cost the pass its second workgroup per CU and
      // the 32-deep contraction per block feeds the MFMAs badly; the aux-stream product it removes was overlapped anyway.  Off by default.
      vq1_in_kernel_dw = wb.vq1part >= 0 && vq1fuse_mode(-1) == 3;
      vq1_bwd(cl, b.S(s.X1), b.W(DGSCT_P_WV1), b.F(DGSCT_P_BV1), b.Wk<float>(wb.coef), B, N, C, 1.f / (float)N, dX1, b.S(s.vq1),
              G(DGSCT_P_BV1), b.Wk<float>(wb.rowpart_v1), row_part_floats(B, C), vq1_in_kernel_dw ? G(DGSCT_P_WV1) : nullptr,
              vq1_in_kernel_dw ? b.Wk<float>(wb.vq1part) : nullptr);
      if (pj.n) defer([=, &side] { part_reduce_run(side.stream, pj); });
    } else {
    // (round 5) dX1 += dvq1 . Wv1 with dvq1 = (vq1 > 0) * E(coef_b / N) formed in the product's staging (gemm_fx.hip); d bias(vq1) from
    // the positive counts the forward left: sum_b E(coef_b / N) * cnt1_b
    Gemm gx1 = mk((int)R, C, C);
    gx1.A = km(b.S(s.vq1), C); gx1.B = b.WB(DGSCT_P_WV1, C, C);
    resid(gx1, dX1, E, C);
    outE(gx1, dX1, E, C);
    GemmFx fx1;
    fx1.a_pro = APRO_MASKSCALE; fx1.rpf = N; fx1.a_cs = b.Wk(wb.coef); fx1.a_cs_dt = DT_F32; fx1.a_cs_ld = C; fx1.a_scale = 1.f / (float)N;
    fx1.a_store = b.Wk(wb.dvq1);
    fx_x1 = (gemmfx_mode(-1) & 8) && (C <= 256 || (gemmfx_mode(-1) & 16)) && gemm_fx_supported(ctx, gx1, fx1);   // (as for dXc above)
    if (fx_x1) {
      if (!(SK & 1024)) gemm_fx(ctx, gx1, fx1);
      defer([=, &side, &b] {                                     // t1 = E(coef / N) * cnt1: summed over the frames with the other bias gradients
        EwArg rdt; rdt.dt = E;
        ew(side, EW_RND_MUL, b.Wk(wb.t1), DT_F32, F32(b.Wk(wb.coef)), F32(b.S(s.cnt1)), rdt, (long)B * C, 1.f / (float)N, 1);
      });
      t1_valid = true;
    } else {
    {
      PartJob pj;                                                // d bias(vq1), likewise
      Ctx cl = ctx; cl.late = aux_stream ? &pj : nullptr;
      if (!(SK & 2)) relu_bwd_scale(cl, b.S(s.vq1), b.S(s.vq1), B, N, C, nullptr, b.Wk(wb.coef), DT_F32, nullptr, 1.f / (float)N,
                     G(DGSCT_P_BV1), b.Wk<float>(wb.rowpart_v1), row_part_floats(B, C));
      if (pj.n) defer([=, &side] { part_reduce_run(side.stream, pj); });
    }
    Gemm g1 = mk((int)R, C, C);                                  // dX1 += dvq1 . Wv1
    g1.A = km(b.S(s.vq1), C); g1.B = b.WB(DGSCT_P_WV1, C, C);
    resid(g1, dX1, E, C);
    outE(g1, dX1, E, C);
    if (!(SK & 1024)) gemm(ctx, g1);
    }
    }
    Gemm g2 = mk(C, C, (int)R);                                  // dWv1 = dvq1^T . X1
    g2.A = mn(fx_x1 ? b.Wk(wb.dvq1) : b.S(s.vq1), C); g2.B = mn(b.S(s.X1), C);
    outF(g2, G(DGSCT_P_WV1), C);
    atomic_out(g2);
    if (!vq1_in_kernel_dw) defer([=, &side] { wgemm(side, g2); });
  }
  // B4 ---- audio queries
  {
    Gemm g1 = mk(C, C, B);                                       // dWa1 = dpa1^T . a
    g1.A = mn(b.Wk(wb.dpa1), C); g1.B = mn(b.S(s.aE), C);
    outF(g1, G(DGSCT_P_WA1), C);
    if (!wgbt) defer([=, &side] { wgemm(side, g1); });
    Gemm g2 = mk(dd, C, B);                                      // dWa2 = dpa2^T . a
    g2.A = mn(b.Wk(wb.dpa2), dd); g2.B = mn(b.S(s.aE), C);
    outF(g2, G(DGSCT_P_WA2), C);
    defer([=, &side, &b] {
      if (!wgbt) wgemm(side, g2);
      // the four bias gradients of the gate MLPs + d fc_affine_v_s_att.weight: column sums of [BT][C] matrices, one launch
      ColsumSeg segs[7] = {{b.Wk(wb.dpre_c), E, B, C, G(DGSCT_P_BCATT)}, {b.Wk(wb.dq), E, B, dd, G(DGSCT_P_BB)},
                           {b.Wk(wb.dpa1), E, B, C, G(DGSCT_P_BA1)}, {b.Wk(wb.dpa2), E, B, dd, G(DGSCT_P_BA2)},
                           {b.Wk(wb.tmpBd), DT_F32, B, dd, G(DGSCT_P_WS)}};
      int nseg = 5;
      if (t3_valid) segs[nseg++] = ColsumSeg{b.Wk(wb.t3), DT_F32, B, dd, G(DGSCT_P_BV2)};     // d bias(vq2), d bias(vq1) of the fused engine
      if (t1_valid) segs[nseg++] = ColsumSeg{b.Wk(wb.t1), DT_F32, B, C, G(DGSCT_P_BV1)};
      colsum_multi(side, segs, nseg);
    });
    side_flush();                                                // dWcatt, dWb, dWv1, dWa1, dWa2 and their biases
    const MatOp wa1T = b.WB(DGSCT_P_WA1, C, C), wa2T = b.WB(DGSCT_P_WA2, C, dd);
    if (skinny_fused_supported(ctx, B, C, C, dd) && (E != DT_BF16 || (wa1T.kmajor && wa2T.kmajor))) {
      SkFuse f; f.M = B; f.N = C; f.K = C;                       // da = dpa1 . Wa1 + dpa2 . Wa2: both products into one tile
      f.A = b.Wk(wb.dpa1); f.lda = C; f.B = wa1T.p; f.ldb = wa1T.ld; f.b_kmajor = wa1T.kmajor;
      f.K2 = dd; f.A2 = b.Wk(wb.dpa2); f.lda2 = dd; f.B2 = wa2T.p; f.ldb2 = wa2T.ld; f.b2_kmajor = wa2T.kmajor;
      f.D = b.Wk(wb.da); f.ddt = DT_F32; f.ldd = C;
      skinny_fused(ctx, f);
    } else {
      Gemm g3 = mk(B, C, C);                                     // da = dpa1 . Wa1 + dpa2 . Wa2
      g3.A = km(b.Wk(wb.dpa1), C); g3.B = wa1T;
      outF(g3, b.Wk<float>(wb.da), C);
      gemm(ctx, g3);
      Gemm g4 = mk(B, C, dd);
      g4.A = km(b.Wk(wb.dpa2), dd); g4.B = wa2T;
      resid(g4, b.Wk(wb.da), DT_F32, C);
      outF(g4, b.Wk<float>(wb.da), C);
      gemm(ctx, g4);
    }
    if (d.temporal) {
      float* dtg = b.Wk<float>(wb.dtg);
      if (dTmap) ew(ctx, EW_ADD_BCAST, dtg, DT_F32, F32(dtg), F32(dTmap), NOARG, B, 1.f, 1);
      ew(ctx, EW_SIGMOID_BWD, b.Wk(wb.dpre_t), DT_F32, F32(dtg), F32(tg), NOARG, B, 0.f, 1);
      colsum_batched(ctx, b.S(s.aE), C, 0, 1, B, C, b.Wk<float>(wb.dpre_t), 0, 1.f, G(DGSCT_P_WT), 0);
      sum_batch(ctx, b.Wk<float>(wb.dpre_t), 1, B, 1, G(DGSCT_P_BT), 1.f, 1);
      ew(ctx, EW_OUTER_ACC, b.Wk(wb.da), DT_F32, F32(b.Wk(wb.dpre_t)), F32(b.F(DGSCT_P_WT)), NOARG, (long)B * C, 0.f, C);
    }
  }
  // B3 ---- X <- tokens attention: dX (output), dtok, d gate_av in one pass over X and dX1 (P2 recomputed)
  {
    if (wide)
      xattn_bwd_wide(ctx, X, dX1, s.tokhi >= 0 ? (const void*)b.S(s.tokhi) : (const void*)b.S(s.tok), b.F(DGSCT_P_GATE_AV), B, N, C, tk, dX,
                     skip_into_dx ? dOut : nullptr, b.Wk<float>(wb.dtokF), G(DGSCT_P_GATE_AV), b.S(s.P2), b.Wk<float>(wb.wdP), b.Wk(wb.wdS));
    else
    if (!(SK & (128 | 65536))) xattn_bwd(ctx, X, dX1, b.S<float>(s.tok), b.F(DGSCT_P_GATE_AV), B, N, C, tk, dX, skip_into_dx ? dOut : nullptr,
              b.Wk<float>(wb.dtokF), G(DGSCT_P_GATE_AV),      // fused skip (f2): out = X + adapter(X, Y) => dX += dOut
              s.tokpk >= 0 ? b.S(s.tokpk) : nullptr);
    if (pair && pair->dx_event) event_record(ctx, pair->dx_event);      // dX is complete
    if (d.remap == DGSCT_REMAP_CONV) {
      // d fc.bias = sum_{b,n} dYp[b,n,:].  Softmax rows sum to 1 and dS1 rows sum to 0, so this equals
      // sum_b (sum_t dtok[b,t,:] + da[b,:]) exactly -- computed from these two small fp32 tensors instead of
      // re-reducing the big bf16-rounded dYp (a cancellation-heavy sum: 30 % relative error in bf16 otherwise).
      if (bias_aux) {
        defer([=, &side, &b] {
          const ColsumSeg segs[2] = {{b.Wk(wb.dtokF), DT_F32, B * tk, C, G(DGSCT_P_BC)}, {b.Wk(wb.da), DT_F32, B, C, G(DGSCT_P_BC)}};
          colsum_multi(side, segs, 2);
        });
      } else {
        const ColsumSeg segs[2] = {{b.Wk(wb.dtokF), DT_F32, B * tk, C, G(DGSCT_P_BC)}, {b.Wk(wb.da), DT_F32, B, C, G(DGSCT_P_BC)}};
        colsum_multi(ctx, segs, 2);
      }
    }
  }
  // B2 ---- tokens <- remapped tokens attention: dYp and d my_tokens in one pass over Yp (P1 recomputed from lse)
  void* dYp = b.Wk(wb.dYp);
  {
    if (wide)
      tokattn_bwd_wide(ctx, b.S(s.Yp), prep_t0hi >= 0 ? (const void*)(b.prep + prep_t0hi) : (const void*)b.F(DGSCT_P_TOKENS), b.Wk<float>(wb.dtokF),
                       b.Wk<float>(wb.da), invN, B, N, C, tk, dYp, b.Wk<float>(wb.dT0b), b.S(s.P1), b.Wk<float>(wb.wdP), b.Wk(wb.wdS),
                       wb.dtokE >= 0 ? b.Wk(wb.dtokE) : nullptr, b.Wk<float>(wb.daN));
    else
    if (!(SK & (128 | 131072))) tokattn_bwd(ctx, b.S(s.Yp), b.F(DGSCT_P_TOKENS), b.S<float>(s.tok), b.S<float>(s.lse), b.Wk<float>(wb.dtokF),
                b.Wk<float>(wb.da), invN, B, N, C, tk, dYp, b.Wk<float>(wb.dT0b), b.Wk<float>(wb.Dtok),
                prep_t0pk >= 0 ? b.prep + prep_t0pk : nullptr, wb.dtokpk >= 0 ? b.Wk(wb.dtokpk) : nullptr);
    defer([=, &side, &b] {                                       // d my_tokens = sum_b (dtok + dS1 . Yp)
      const ColsumSeg segs[2] = {{b.Wk(wb.dtokF), DT_F32, B, tk * C, G(DGSCT_P_TOKENS)}, {b.Wk(wb.dT0b), DT_F32, B, tk * C, G(DGSCT_P_TOKENS)}};
      colsum_multi(side, segs, 2);
    });
  }
  // B1 ---- remap
  {
    const bool conv = d.remap == DGSCT_REMAP_CONV;
    // The two weight gradients go to the aux stream as soon as their operands exist; the remap's BIAS gradients (three
    // small reductions over dYp) stay on the main stream, after dY: the tail of the call is then ~balanced between the
    // two streams (they used to queue behind each other on aux while main sat in the final join).
    if (orderA) {
      Gemm g2 = mk(C, Co, (int)R);                               // dWc = dYp^T . T1
      g2.A = mn(dYp, C); g2.B = mn(b.S(s.T), Co);
      outF(g2, G(DGSCT_P_WC), Co);
      atomic_out(g2);
      defer([=, &side] { wgemm(side, g2); });
      side_flush();
      Gemm g1 = mk((int)R, Co, C);                               // dT1 = dYp . Wc
      g1.A = km(dYp, C); g1.B = b.WB(DGSCT_P_WC, Co, C);
      outE(g1, b.Wk(wb.dT), E, Co);
      if (!(SK & 2048)) gemm(ctx, g1);
      if (conv) {
        Gemm g4 = mk(N, No, Co);                                 // dWn = sum_b dT1[b] . Y[b]^T
        g4.KB = B;
        g4.A = km(b.Wk(wb.dT), Co, 0, (long)N * Co);
        g4.B = km(Y, Co, 0, (long)No * Co);
        outF(g4, G(DGSCT_P_WN), No);
        atomic_out(g4); g4.sole_writer = 1;
        defer([=, &side] { wgemm(side, g4); });
        side_flush();
      }
      if (!(SK & 2048) && phase != 1) dy_product();             // dY[b] = Wn^T . dT1[b]
    } else {
      if (conv) {
        Gemm g2 = mk(N, No, C);                                  // dWn = sum_b dYp[b] . T2[b]^T
        g2.KB = B;
        g2.A = km(dYp, C, 0, (long)N * C);
        g2.B = mn(b.S(s.T), Nop, 0, (long)C * Nop);
        outF(g2, G(DGSCT_P_WN), No);
        atomic_out(g2); g2.sole_writer = 1;
        defer([=, &side] { wgemm(side, g2); });
      }
      side_flush();
      Gemm g1 = mk(No, C, N, B);                                 // dT2[b] = Wn^T . dYp[b]     [No][C] token-major
      g1.A = mn(b.W(DGSCT_P_WN), No);                            //   (M = No, N = C: the 128x96 remap tile; the transposed
      g1.B = mn(dYp, C, (long)N * C);                            //    form M = C = 96 runs at half the rate)
      outE(g1, b.Wk(wb.dT), E, C, (long)No * C);
      if (!(SK & 2048)) gemm(ctx, g1);
      Gemm g4 = mk(C, Co, No);                                   // dWc = sum_b dT2[b]^T . Y[b]
      g4.KB = B;
      g4.A = mn(b.Wk(wb.dT), C, 0, (long)No * C);
      g4.B = mn(Y, Co, 0, (long)No * Co);
      outF(g4, G(DGSCT_P_WC), Co);
      atomic_out(g4);
      defer([=, &side] { wgemm(side, g4); });
      side_flush();
      if (!(SK & 2048) && phase != 1) dy_product();             // dY[b] = dT2[b] . Wc
    }
    // both bias-side reductions of dYp in one pass (they were rowdot -> sum_batch and colsum: two more reads of the cotangent)
    // (round 4: frame by frame -- contiguous rows -- with the per-frame row dots summed by a second small launch: 150 -> ~45 us
    //  at 655 360 x 96)
    auto bias_sums = [=, &b](const Ctx& c) {
      if (conv)                                                  // dbn[n] = sum dYp . colb;  d rowsum(Wc)[c] = sum rowb[n] dYp
        rowdot_colsum_frames(c, dYp, C, (long)N * C, B, N, C, b.colb(), b.rowb(), G(DGSCT_P_BN), b.Wk<float>(wb.dwcsum),
                             b.Wk<float>(wb.rowtmp), b.Wk<float>(wb.rowpart), row_part_floats(B, C));
      else
        rowdot_colsum_frames(c, dYp, C, (long)N * C, B, N, C, nullptr, b.rowb(), nullptr, G(DGSCT_P_BC), b.Wk<float>(wb.rowtmp),
                             b.Wk<float>(wb.rowpart), row_part_floats(B, C));
    };
    if (bias_aux)                                                // ... and the broadcast of d rowsum(Wc) into dWc right behind them (below)
      defer([=, &side, &b] {
        bias_sums(side);
        if (conv) ew(side, EW_ADD_BCAST, G(DGSCT_P_WC), DT_F32, F32(G(DGSCT_P_WC)), F32(b.Wk(wb.dwcsum)), NOARG, (long)C * Co, 1.f, Co);
      });
    else bias_sums(ctx);
  }
  side_flush();
  // d rowsum(Wc)[c] is broadcast over co into dWc, which the aux stream accumulates: after the join -- or, when the caller takes over the
  // ordering (DGSCT_BWD_NO_JOIN: it orders whatever reads `grads`, reuses `ws` or frees the inputs after the aux stream itself, so the
  // chain of the NEXT call never waits for this call's weight gradients), on the aux stream behind them
  if (!deferred && !(SK & 8192)) stream_join(ctx);
  if (d.remap == DGSCT_REMAP_CONV) {
    if (deferred && !bias_aux) { stream_fork(ctx); ew(side, EW_ADD_BCAST, G(DGSCT_P_WC), DT_F32, F32(G(DGSCT_P_WC)), F32(b.Wk(wb.dwcsum)), NOARG, (long)C * Co, 1.f, Co); }
    else if (!deferred) ew(ctx, EW_ADD_BCAST, G(DGSCT_P_WC), DT_F32, F32(G(DGSCT_P_WC)), F32(b.Wk(wb.dwcsum)), NOARG, (long)C * Co, 1.f, Co);
  }
  check_async("dgsct_adapter_backward");
  return has_error() ? 1 : 0;
}

}  // namespace dgsct
