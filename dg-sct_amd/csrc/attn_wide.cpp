// The two latent-token attentions for num_tokens > 32 (reference net_trans.py:572-589 and its autograd; the reference constructor's
// default is num_tk = 87, net_trans.py:437 -- no launcher passes more than 32).
//
// The fused kernels (attn.hip, attn2.hip) hold ONE 32-row MFMA tile of latent tokens per frame in registers / LDS: their P / dS
// images, the packed token fragments and the dtok accumulators are all sized by that tile (three token
// tiles would need 232 KB of LDS in xattn_bwd).  A frame with more latent tokens takes this path instead: the same arithmetic as a
// chain of per-frame batched products on the tiled engine (gemm.hip) with the row softmax kernels between them, the probabilities
// P1 [B][tk][N] and P2 [B][N][tk] SAVED in E by the forward instead of recomputed from log-sum-exp statistics.
//
// Plain C++ against prims.h, like plan.cpp: linked into libdgsct.so and into the host emulation of the CPU tests alike.
//
// Rounding points (bf16 mode), matching the fused kernels where it matters: the fp32 latent tokens enter the two LOGIT products as
// hi + lo bf16 pairs (two accumulating launches; the un-scaled logits amplify a bf16-rounded operand's error into the softmax),
// and as their hi part everywhere else; P1, P2, dS1, dS2 are stored in E; tok, dtok, dT0b and every logit / dP image are fp32.
#include "prims.h"

namespace dgsct {
namespace {
inline MatOp km(const void* p, long ld, long bs = 0) { MatOp m; m.p = p; m.ld = ld; m.kmajor = 1; m.bs = bs; return m; }
inline MatOp mn(const void* p, long ld, long bs = 0) { MatOp m; m.p = p; m.ld = ld; m.kmajor = 0; m.bs = bs; return m; }
inline Gemm mk(int M, int N, int K, int batch) { Gemm g; g.M = M; g.N = N; g.K = K; g.batch = batch; return g; }
inline void outF(Gemm& g, float* D, long ld, long dbs) { g.D = D; g.ddt = DT_F32; g.ldd = ld; g.dbs = dbs; }
inline void outE(Gemm& g, void* D, int E, long ld, long dbs) { g.D = D; g.ddt = E; g.ldd = ld; g.dbs = dbs; }
inline void resid(Gemm& g, const void* R, int rdt, long ld, long rbs) { g.R = R; g.rdt = rdt; g.ldr = ld; g.rbs = rbs; g.beta = 1.f; }
inline long rup8(long x) { return (x + 7) / 8 * 8; }
}  // namespace

long wide_attn_image_elems(int B, int N, int tk) {
  const long a = (long)B * tk * rup8(N), b = (long)B * N * rup8(tk);
  return a > b ? a : b;
}

// tok = T0 + softmax_N(T0 Yp^T) Yp;  a = mean_N Yp                                           net_trans.py:572-580, :592
void tokattn_fwd_wide(const Ctx& ctx, const void* Yp, const float* T0, const void* T0hi, const void* T0lo, int B, int N, int C, int tk,
                      float invN, float* tok, float* a, void* aE, float* L, void* P1) {
  const int E = ctx.mode;
  const long Np = rup8(N), fs = (long)tk * Np;
  Gemm g = mk(tk, N, C, B);                                      // L[b][t][n] = T0[t] . Yp[b][n]
  g.A = km(T0hi, C); g.B = km(Yp, C, (long)N * C);
  outF(g, L, Np, fs);
  gemm(ctx, g);
  if (T0lo) { g.A = km(T0lo, C); resid(g, L, DT_F32, Np, fs); gemm(ctx, g); }
  softmax_rows(ctx, L, Np, P1, E, Np, (long)B * tk, N, 0);
  Gemm h = mk(tk, C, N, B);                                      // tok[b] = T0 + P1[b] . Yp[b]
  h.A = km(P1, Np, fs); h.B = mn(Yp, C, (long)N * C);
  resid(h, T0, DT_F32, C, 0);
  outF(h, tok, C, (long)tk * C);
  gemm(ctx, h);
  colsum_batched(ctx, Yp, C, (long)N * C, B, N, C, nullptr, 0, invN, a, C);
  const CvtSeg seg{a, aE, (long)B * C, E, 0};
  cvt_multi(ctx, &seg, 1);
}

// X1 = X + gate_av * softmax_tk(X tok^T) tok                                                 net_trans.py:583-589
// tokhi / toklo: E [B][tk][C], WRITTEN here in bf16 mode (toklo == null in fp32 mode: tokhi is then `tok` itself)
void xattn_fwd_wide(const Ctx& ctx, const void* X, const float* tok, const float* gate_av, int B, int N, int C, int tk, void* X1,
                    void* tokhi, void* toklo, float* L, void* P2) {
  const int E = ctx.mode;
  const long tkp = rup8(tk), fs = (long)N * tkp;
  if (toklo) split_hilo(ctx, tok, (long)B * tk * C, tokhi, toklo);
  const void* th = toklo ? tokhi : (const void*)tok;
  Gemm g = mk(N, tk, C, B);                                      // L[b][n][t] = X[b][n] . tok[b][t]
  g.A = km(X, C, (long)N * C); g.B = km(th, C, (long)tk * C);
  outF(g, L, tkp, fs);
  gemm(ctx, g);
  if (toklo) { g.B = km(toklo, C, (long)tk * C); resid(g, L, DT_F32, tkp, fs); gemm(ctx, g); }
  softmax_rows(ctx, L, tkp, P2, E, tkp, (long)B * N, tk, 0);
  Gemm h = mk(N, C, tk, B);                                      // X1[b] = X[b] + gate_av * P2[b] . tok[b]
  h.A = km(P2, tkp, fs); h.B = mn(th, C, (long)tk * C);
  h.alpha_ptr = gate_av;
  resid(h, X, E, C, (long)N * C);
  outE(h, X1, E, C, (long)N * C);
  gemm(ctx, h);
}

// dX = dX1 + dS2 tok (+ R2);  dtok = gate_av P2^T dX1 + dS2^T X;  *dgate += sum P2 (dX1 tok^T),   dS2 = gate_av P2 (dP2 - rowsum(P2 dP2))
void xattn_bwd_wide(const Ctx& ctx, const void* X, const void* dX1, const void* tokhi, const float* gate_av, int B, int N, int C, int tk,
                    void* dX, const void* R2, float* dtok, float* dgate, const void* P2, float* dP, void* dS) {
  const int E = ctx.mode;
  const long tkp = rup8(tk), fs = (long)N * tkp;
  Gemm g = mk(N, tk, C, B);                                      // dP2[b][n][t] = dX1[b][n] . tok[b][t]
  g.A = km(dX1, C, (long)N * C); g.B = km(tokhi, C, (long)tk * C);
  outF(g, dP, tkp, fs);
  gemm(ctx, g);
  softmax_bwd_rows(ctx, P2, tkp, dP, tkp, dS, E, tkp, (long)B * N, tk, gate_av, dgate);
  Gemm h = mk(N, C, tk, B);                                      // dX[b] = dX1[b] + dS2[b] . tok[b] (+ R2[b])
  h.A = km(dS, tkp, fs); h.B = mn(tokhi, C, (long)tk * C);
  resid(h, dX1, E, C, (long)N * C);
  h.R2 = R2;
  outE(h, dX, E, C, (long)N * C);
  gemm(ctx, h);
  Gemm t = mk(tk, C, N, B);                                      // dtok[b] = gate_av * P2[b]^T . dX1[b]
  t.A = mn(P2, tkp, fs); t.B = mn(dX1, C, (long)N * C);
  t.alpha_ptr = gate_av;
  outF(t, dtok, C, (long)tk * C);
  gemm(ctx, t);
  Gemm u = mk(tk, C, N, B);                                      //          += dS2[b]^T . X[b]
  u.A = mn(dS, tkp, fs); u.B = mn(X, C, (long)N * C);
  resid(u, dtok, DT_F32, C, (long)tk * C);
  outF(u, dtok, C, (long)tk * C);
  gemm(ctx, u);
}

// dYp = P1^T dtok + dS1^T T0 + invN da[b];  dT0b = dS1 Yp,   dS1 = P1 (dP1 - rowsum(P1 dP1)),  dP1 = dtok Yp^T
// dtokE: E [B][tk][C] of scratch (bf16 mode; null in fp32 mode: dtok itself is the operand);  daN: B * C floats of scratch
void tokattn_bwd_wide(const Ctx& ctx, const void* Yp, const void* T0hi, const float* dtok, const float* da, float invN, int B, int N, int C,
                      int tk, void* dYp, float* dT0b, const void* P1, float* dP, void* dS, void* dtokE, float* daN) {
  const int E = ctx.mode;
  const long Np = rup8(N), fs = (long)tk * Np;
  const void* dt = dtok;
  if (dtokE) {
    const CvtSeg seg{dtok, dtokE, (long)B * tk * C, E, 0};
    cvt_multi(ctx, &seg, 1);
    dt = dtokE;
  }
  Gemm g = mk(tk, N, C, B);                                      // dP1[b][t][n] = dtok[b][t] . Yp[b][n]
  g.A = km(dt, C, (long)tk * C); g.B = km(Yp, C, (long)N * C);
  outF(g, dP, Np, fs);
  gemm(ctx, g);
  softmax_bwd_rows(ctx, P1, Np, dP, Np, dS, E, Np, (long)B * tk, N, nullptr, nullptr);
  EwArg src; src.p = da; src.dt = DT_F32;
  ew(ctx, EW_SCALE, daN, DT_F32, src, EwArg{}, EwArg{}, (long)B * C, invN, 1);
  Gemm h = mk(N, C, tk, B);                                      // dYp[b] = P1[b]^T . dtok[b] + invN da[b]
  h.A = mn(P1, Np, fs); h.B = mn(dt, C, (long)tk * C);
  h.bias_n = daN; h.bias_n_bs = C;
  outE(h, dYp, E, C, (long)N * C);
  gemm(ctx, h);
  Gemm k = mk(N, C, tk, B);                                      //         += dS1[b]^T . T0
  k.A = mn(dS, Np, fs); k.B = mn(T0hi, C, 0);
  resid(k, dYp, E, C, (long)N * C);
  outE(k, dYp, E, C, (long)N * C);
  gemm(ctx, k);
  Gemm w = mk(tk, C, N, B);                                      // dT0b[b] = dS1[b] . Yp[b]
  w.A = km(dS, Np, fs); w.B = mn(Yp, C, (long)N * C);
  outF(w, dT0b, C, (long)tk * C);
  gemm(ctx, w);
}

}  // namespace dgsct
