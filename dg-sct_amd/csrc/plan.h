// Buffer layout + kernel schedule of one adapter call (see plan.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/dgsct.h"

namespace dgsct {

struct Region { std::string name; int64_t offset, bytes; };

// The dY product split off the call (dgsct_adapter_backward_ex2): `phase` 1 = everything but the product that writes dY, 2 = only that
// product (reads the call's workspace, nothing else of phase 1's arguments), 0 = both.  dy_residual is added to dY in the product's
// epilogue; dx_event is recorded on the stream once dX is complete, dy_wait is waited for in front of the dY product.
struct BwdPair {
  int phase = 0;
  const void* dy_residual = nullptr;
  void* dx_event = nullptr;
  void* dy_wait = nullptr;
};

struct Plan {
  explicit Plan(const dgsct_adapter_desc& d, bool record_regions = false);
  bool record_regions_ = false;
  bool ok = false;
  dgsct_adapter_desc d;
  int B, N, C, No, Co, tk, g, dd, ds, E, Np, Nop, tkp;
  int64_t es, R;
  bool fp8 = false;   // DGSCT_BF16_FP8: fp8 operands for fc / fc_affine_video_1 / fc_affine_video_2 (forward)
  int64_t prep_w8[3] = {-1, -1, -1}, prep_w8scale = -1;   // fp8 copies of Wc, Wv1, Wv2 + their 3 inverse scales (+ 1 scratch word)
  bool xc_scratch = false;   // Xc = X1 (1 + ch) is scratch (the fused gate backward writes it), not a saved activation
  bool wide = false;   // num_tokens > 32: the latent-token attentions run as batched products + row softmax (attn_wide.cpp)
  bool orderA;   // remap association: (Wn.Y).Wc^T (A) or Wn.(Y.Wc^T) (B), whichever is cheaper

  // prep
  int64_t prep_w[DGSCT_P_COUNT], prep_wt[DGSCT_P_COUNT], wcols[DGSCT_P_COUNT], wnumel[DGSCT_P_COUNT], prep_rowb, prep_colb, prep_colb2, prep_t0pk, prep_t0hi, prep_t0lo, prep_bytes;
  // saved
  struct {
    int64_t a, mvq1, cnt1, bnacc1, bnacc2, zero_end;
    int64_t Yp, T, tok, tokpk, lse, aE, X1, aq1, aq2, vq1, m1, q, ch, Xc, vq2, sl, sg, map, tg, X3, mu_b, rstd_b, Zp, Z, Op,
        bn1, bn2, mu_p, rstd_p, P1, P2, tokhi;   // (P1 / P2 / tokhi: the wide path only)
  } s;
  std::vector<Region> saved_regions;
  int64_t saved_bytes;
  // forward / backward scratch
  struct { int64_t tokscr, Xc, wL, toklo; } wf;
  struct {
    int64_t bnsums2, bnsums1, dch, dtg, u, dwcsum, dtokF, dT0b, w2, zero_end;
    int64_t dO, dZ, dX3, dX1, dXc, Xc, dsg, dsl, tmpBd, dpre_c, dq, dm1, dpa1, dpa2, coef, da, dpre_t, Dtok, dtokpk, dYp, dT, rowtmp, rowpart, rowpart_v1, rowpart_v2, vq1part, dvq1, dvq2, dZp, t1, t3, wdP, wdS, dtokE, daN;
  } wb;
  int64_t ws_fwd_bytes, ws_bwd_bytes;
  // gradients
  int64_t grad_off[DGSCT_P_COUNT], grad_numel[DGSCT_P_COUNT], grad_floats;

  int prepare(float* const* params, void* prep, void* stream) const;
  int forward(float* const* params, const void* prep, const void* X, const void* Y, void* out, float* map, float* tmap,
              void* saved, void* ws, void* stream, const void* residual = nullptr, void* aux_stream = nullptr) const;
  int backward(float* const* params, const void* prep, const void* X, const void* Y, const void* saved, const void* dOut,
               const float* dMap, const float* dTmap, void* dX, void* dY, float* grads, void* ws, void* stream,
               void* aux_stream = nullptr, bool skip_into_dx = false, bool no_join = false, const BwdPair* pair = nullptr) const;

 private:
  bool validate();
  void layout();
};

// what-if timing switches of the schedule (plan.cpp; dgsct_test_tune "skip" / "skipminc"): set < 0 queries
int plan_skip_mode(int set);
int plan_skip_minc(int set);
int plan_skip_maxc(int set);

}  // namespace dgsct
