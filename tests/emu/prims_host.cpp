// TEST INFRASTRUCTURE ONLY -- never linked into libdgsct.so, never loaded by the product package.
//
// Plain host-loop implementation of csrc/prims.h.  tests/emu builds csrc/plan.cpp + csrc/capi.cpp
// against THIS file (g++, no HIP) into tests/emu/libdgsct_emu.so so that the kernel SCHEDULE
// (plan.cpp: operand roles, strides, offsets, gradient layout) can be checked against the oracle in
// the CPU-only build container.  The gfx950 kernels themselves are checked on the GPU against torch.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../dg-sct_amd/csrc/prims.h"
#include "../../dg-sct_amd/csrc/gemm_int.h"
#include "../../dg-sct_amd/csrc/err.h"

namespace dgsct {

static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
static inline uint16_t f2bf(float f) {
  uint32_t u; std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float ld(const void* p, int dt, long i) { return dt == DT_F32 ? ((const float*)p)[i] : bf2f(((const uint16_t*)p)[i]); }
static inline void st(void* p, int dt, long i, float v) { if (dt == DT_F32) ((float*)p)[i] = v; else ((uint16_t*)p)[i] = f2bf(v); }
static inline float sigm(float x) { return 1.f / (1.f + std::exp(-x)); }

void gemm_prof_enable(int) {}
void gemm_prof_collect(long* n, double* ms, double* fl) { if (n) *n = 0; if (ms) *ms = 0; if (fl) *fl = 0; }

void* stream_create(int) { return nullptr; }
void stream_destroy(void*) {}
void stream_fork(const Ctx&) {}
int call_prof_mode(int) { return 0; }
void* call_prof_begin(void*, int, int, int) { return nullptr; }
void call_prof_end(void*) {}
void call_prof_dump(const char*) {}
void part_reduce_run(void*, const PartJob&) {}   // (the host primitives never leave a second stage behind)
void stream_join(const Ctx&) {}
void event_record(const Ctx&, void*) {}
void event_wait(const Ctx&, void*) {}
void check_async(const char*) {}
void clear_async() {}
int gemm_skinny_mode(int) { return 0; }
static int g_wgbt = 1;
static int g_tfs8 = 1;
int tokattn_small8_mode(int set) { const int old = g_tfs8; if (set >= 0) g_tfs8 = set ? 1 : 0; return old; }      // (a kernel-shape switch: nothing to emulate)
int wgrad_bt_mode(int set) { const int old = g_wgbt; if (set >= 0) g_wgbt = set ? 1 : 0; return old; }
bool wgrad_bt_supported(const Ctx&, const WgBtJob*, int n) { return g_wgbt && n >= 1 && n <= WGBT_MAX; }
void wgrad_bt(const Ctx& ctx, const WgBtJob* jobs, int n) {          // (either element type: the host loops read through ld())
  for (int i = 0; i < n; ++i) {
    const WgBtJob& j = jobs[i];
    for (int m = 0; m < j.M; ++m)
      for (int c = 0; c < j.N; ++c) {
        double s = 0;
        for (int k = 0; k < j.K; ++k) s += (double)ld(j.A, ctx.mode, (long)k * j.lda + m) * (double)ld(j.B, ctx.mode, (long)k * j.ldb + c);
        j.D[(long)m * j.ldd + c] = (float)s;
      }
  }
}
int gemm_tall_mode(int) { return 0; }
int gemm8_mode(int) { return 0; }                 // (the 8-wave GEMM kernel is a device-side choice: nothing to emulate)
int gemm_noatomic_mode(int) { return 0; }
int gemm_cfgx_mode(int) { return 0; }
int gemm8_wg_target(int) { return 0; }
int gemm8_pipe_mode(int) { return 0; }
int gemm8_stag_mode(int) { return 0; }

void zero(const Ctx&, void* p, size_t bytes) { if (bytes) std::memset(p, 0, bytes); }

static void gemm_softmax(const Ctx& ctx, const Gemm& g) {      // ACT_SOFTMAX / ACT_SOFTMAX_BWD: column-wise, transposed out
  const int E = ctx.mode;
  const float alpha = g.alpha * (g.alpha_ptr ? *g.alpha_ptr : 1.f);
  const float sc = g.sm_scale ? *g.sm_scale : 1.f;
  std::vector<float> col(g.M);
  double tot = 0;
  for (int b = 0; b < g.batch; ++b)
    for (int n = 0; n < g.N; ++n) {
      for (int m = 0; m < g.M; ++m) {
        double acc = 0;
        for (int kb = 0; kb < g.KB; ++kb)
          for (int k = 0; k < g.K; ++k) {
            const long ao = (long)b * g.A.bs + (long)kb * g.A.kbs + (g.A.kmajor ? (long)m * g.A.ld + k : (long)k * g.A.ld + m);
            const long bo = (long)b * g.B.bs + (long)kb * g.B.kbs + (g.B.kmajor ? (long)n * g.B.ld + k : (long)k * g.B.ld + n);
            acc += (double)ld(g.A.p, E, ao) * (double)ld(g.B.p, E, bo);
          }
        col[m] = alpha * (float)acc;
      }
      const long o = (long)b * g.dbs + (long)n * g.ldd;
      const long po = (long)b * g.maskbs + (long)n * g.ldmask;
      if (g.act == ACT_SOFTMAX) {
        float mx = -INFINITY;
        for (int m = 0; m < g.M; ++m) mx = std::max(mx, col[m]);
        double s = 0;
        for (int m = 0; m < g.M; ++m) s += std::exp(col[m] - mx);
        for (int m = 0; m < g.M; ++m) st(g.D, g.ddt, o + m, (float)(std::exp(col[m] - mx) / s));
      } else {
        double pd = 0;
        for (int m = 0; m < g.M; ++m) pd += (double)ld(g.mask, E, po + m) * col[m];
        tot += pd;
        for (int m = 0; m < g.M; ++m) st(g.D, g.ddt, o + m, sc * ld(g.mask, E, po + m) * (col[m] - (float)pd));
      }
    }
  if (g.act == ACT_SOFTMAX_BWD && g.sm_dot) *g.sm_dot += (float)tot;
}
void zero2(const Ctx& c, void* a, size_t abytes, void* b, size_t bbytes) { zero(c, a, abytes); zero(c, b, bbytes); }

void gemm(const Ctx& ctx, const Gemm& g) {
  if (g.act == ACT_SOFTMAX || g.act == ACT_SOFTMAX_BWD) { gemm_softmax(ctx, g); return; }
  const int E = ctx.mode;
  const float alpha = g.alpha * (g.alpha_ptr ? *g.alpha_ptr : 1.f);
  for (int b = 0; b < g.batch; ++b)
    for (int m = 0; m < g.M; ++m)
      for (int n = 0; n < g.N; ++n) {
        double acc = 0;
        for (int kb = 0; kb < g.KB; ++kb)
          for (int k = 0; k < g.K; ++k) {
            const long ao = (long)b * g.A.bs + (long)kb * g.A.kbs + (g.A.kmajor ? (long)m * g.A.ld + k : (long)k * g.A.ld + m);
            const long bo = (long)b * g.B.bs + (long)kb * g.B.kbs + (g.B.kmajor ? (long)n * g.B.ld + k : (long)k * g.B.ld + n);
            acc += (double)ld(g.A.p, E, ao) * (double)ld(g.B.p, E, bo);
          }
        float v = alpha * (float)acc;
        const int mm = g.m_mod > 0 ? m % g.m_mod : m;
        if (g.bias_m) v += g.bias_m[mm];
        if (g.bias_n) v += g.bias_n[(long)b * g.bias_n_bs + n];
        if (g.r1_m) v += g.r1_m[mm] * g.r1_n[n];
        if (g.act == ACT_RELU) v = std::max(v, 0.f);
        else if (g.act == ACT_SIGMOID) v = sigm(v);
        if (g.mask && !(ld(g.mask, E, (long)b * g.maskbs + (long)m * g.ldmask + n) > 0.f)) v = 0.f;
        if (g.R) v += g.beta * ld(g.R, g.rdt, (long)b * g.rbs + (long)m * g.ldr + n);
        if (g.R2) v += ld(g.R2, g.rdt, (long)b * g.rbs + (long)m * g.ldr + n);
        const long o = (long)b * g.dbs + (long)m * g.ldd + n;
        if (g.atomic) ((float*)g.D)[o] += v;
        else st(g.D, g.ddt, o, v);
      }
}

static int g_vq1fuse = 1;
int vq1fuse_mode(int set) { const int old = g_vq1fuse; if (set >= 0) g_vq1fuse = set > 3 ? 1 : set; return old; }
long vq1_wpart_floats(int C) { return (long)C * C; }
bool vq1_fused_shape(int, int, int) { return true; }                   // (host loops: any shape, either element type)
bool vq1_fused_supported(int, int, int) { return g_vq1fuse != 0; }
void vq1sum_fwd(const Ctx& ctx, const void* X1, const void* Wv1, const float* bv1, int B, int N, int C, float invN, float* msum, void* vq1) {
  const int E = ctx.mode;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      double sum = 0;
      for (int n = 0; n < N; ++n) {
        double acc = 0;
        for (int k = 0; k < C; ++k) acc += (double)ld(X1, E, ((long)b * N + n) * C + k) * ld(Wv1, E, (long)c * C + k);
        sum += std::max((float)acc + bv1[c], 0.f);
        if (vq1) st(vq1, E, ((long)b * N + n) * C + c, std::max((float)acc + bv1[c], 0.f));
      }
      msum[(long)b * C + c] += invN * (float)sum;
    }
}
void vq1_bwd(const Ctx& ctx, const void* X1, const void* Wv1, const float* bv1, const float* coef, int B, int N, int C, float invN,
             void* dX1, void* dvq1_out, float* dbv1, float*, long, float* dWv1, float*) {
  const int E = ctx.mode;
  std::vector<float> dv(C);
  std::vector<char> one_row((size_t)C * 4);
  std::vector<double> dw(dWv1 ? (size_t)C * C : 0, 0.0);
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      const long row = (long)b * N + n;
      for (int c = 0; c < C; ++c) {
        double acc = 0;
        for (int k = 0; k < C; ++k) acc += (double)ld(X1, E, row * C + k) * ld(Wv1, E, (long)c * C + k);
        float v = (float)acc + bv1[c] > 0.f ? invN * coef[(long)b * C + c] : 0.f;
        st(one_row.data(), E, c, v);                         // rounded to E once, as the stored tensor would be
        dv[c] = ld(one_row.data(), E, c);
        if (!dWv1) st(dvq1_out, E, row * C + c, v);
        dbv1[c] += dv[c];
        if (dWv1) for (int k = 0; k < C; ++k) dw[(size_t)c * C + k] += (double)dv[c] * ld(X1, E, row * C + k);
      }
      for (int k = 0; k < C; ++k) {
        double acc = 0;
        for (int c = 0; c < C; ++c) acc += (double)dv[c] * ld(Wv1, E, (long)c * C + k);
        st(dX1, E, row * C + k, ld(dX1, E, row * C + k) + (float)acc);
      }
    }
  if (dWv1) for (size_t i = 0; i < dw.size(); ++i) dWv1[i] += (float)dw[i];
}

static int g_skfuse = 1;
int skfuse_mode(int set) { const int old = g_skfuse; if (set >= 0) g_skfuse = set ? 1 : 0; return old; }
bool skinny_fused_supported(const Ctx&, int M, int, int, int) { return g_skfuse && M <= 256; }   // (host loops: any width, either element type)
void skinny_fused(const Ctx& ctx, const SkFuse& p) {
  const int E = ctx.mode;
  std::vector<float> a((size_t)p.M * p.K);
  for (int m = 0; m < p.M; ++m)
    for (int k = 0; k < p.K; ++k) {
      float v;
      if (p.a_mode == 0) v = ld(p.A, E, (long)m * p.lda + k);
      else {
        const float mu = p.a_mul[(long)m * p.ld_mul + k];
        v = p.a_mode == 1 ? ld(p.A, E, (long)m * p.lda + k) * mu : ((const float*)p.A)[(long)m * p.lda + k] * mu * (1.f - mu);
        if (E == DT_BF16) v = bf2f(f2bf(v));                   // the operand is rounded to E before it is multiplied
        if (p.a_store) st(p.a_store, E, (long)m * p.ld_store + k, v);
      }
      a[(size_t)m * p.K + k] = v;
    }
  for (int m = 0; m < p.M; ++m)
    for (int n = 0; n < p.N; ++n) {
      double acc = 0;
      for (int k = 0; k < p.K; ++k) acc += (double)a[(size_t)m * p.K + k] * ld(p.B, E, p.b_kmajor ? (long)n * p.ldb + k : (long)k * p.ldb + n);
      for (int k = 0; k < p.K2; ++k)
        acc += (double)ld(p.A2, E, (long)m * p.lda2 + k) * ld(p.B2, E, p.b2_kmajor ? (long)n * p.ldb2 + k : (long)k * p.ldb2 + n);
      float v = (float)acc;
      if (p.bias_n) v += p.bias_n[n];
      if (p.act == ACT_RELU) v = std::max(v, 0.f);
      else if (p.act == ACT_SIGMOID) v = sigm(v);
      if (p.mask && !(ld(p.mask, E, (long)m * p.ldmask + n) > 0.f)) v = 0.f;
      if (p.epi == 1) {
        const float q = ld(p.e_q, E, (long)m * p.ld_eq + n);
        st(p.D, E, (long)m * p.ldd + n, q > 0.f ? v * p.e_mul[(long)m * p.ld_emul + n] : 0.f);
        p.D2[(long)m * p.ldd2 + n] = v * q;
      } else {
        st(p.D, p.ddt, (long)m * p.ldd + n, v);
      }
    }
}

void softmax_rows(const Ctx&, const float* in, long ld_in, void* out, int odt, long ld_out, long rows, int L, int pre_tanh) {
  for (long r = 0; r < rows; ++r) {
    float m = -INFINITY;
    for (int c = 0; c < L; ++c) { float v = in[r * ld_in + c]; if (pre_tanh) v = std::tanh(v); m = std::max(m, v); }
    double s = 0;
    for (int c = 0; c < L; ++c) { float v = in[r * ld_in + c]; if (pre_tanh) v = std::tanh(v); s += std::exp(v - m); }
    for (long c = 0; c < ld_out; ++c) {
      float o = 0.f;
      if (c < L) { float v = in[r * ld_in + c]; if (pre_tanh) v = std::tanh(v); o = (float)(std::exp(v - m) / s); }
      st(out, odt, r * ld_out + c, o);
    }
  }
}

void softmax_bwd_rows(const Ctx& ctx, const void* P, long ldp, const float* dP, long lddp, void* out, int odt, long ldo,
                      long rows, int L, const float* scale_ptr, float* dot_accum) {
  const float sc = scale_ptr ? *scale_ptr : 1.f;
  double tot = 0;
  for (long r = 0; r < rows; ++r) {
    double pd = 0;
    for (int c = 0; c < L; ++c) pd += (double)ld(P, ctx.mode, r * ldp + c) * dP[r * lddp + c];
    tot += pd;
    for (long c = 0; c < ldo; ++c)
      st(out, odt, r * ldo + c, c < L ? sc * ld(P, ctx.mode, r * ldp + c) * (dP[r * lddp + c] - (float)pd) : 0.f);
  }
  if (dot_accum) *dot_accum += (float)tot;
}

void colsum_batched(const Ctx& ctx, const void* x, long ldx, long bs, int B, int N, int C, const float* roww, long roww_bs,
                    float scale, float* out, long out_bs) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      double s = 0;
      for (int n = 0; n < N; ++n)
        s += (double)(roww ? roww[(long)b * roww_bs + n] : 1.f) * ld(x, ctx.mode, (long)b * bs + (long)n * ldx + c);
      out[(long)b * out_bs + c] += scale * (float)s;
    }
}

void colsum_batched_pos(const Ctx& ctx, const void* x, long ldx, long bs, int B, int N, int C, const float* roww, long roww_bs,
                        float scale, float* out, long out_bs, float* out_pos, long out_pos_bs) {
  colsum_batched(ctx, x, ldx, bs, B, N, C, roww, roww_bs, scale, out, out_bs);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      double s = 0;
      for (int n = 0; n < N; ++n)
        if (ld(x, ctx.mode, (long)b * bs + (long)n * ldx + c) > 0.f) s += roww ? roww[(long)b * roww_bs + n] : 1.f;
      out_pos[(long)b * out_pos_bs + c] += (float)s;
    }
}

// ---- fused GEMM hooks (csrc/gemm_fx.hip): the same math in host loops, either element type, either B layout ------------------------
static int g_gemmfx = 31 + 64 + 128;      // (all call sites, prologues at every width: the host loops have no tile economics)
int gemmfx_mode(int set) { const int old = g_gemmfx; if (set >= 0) g_gemmfx = set & 255; return old; }
bool gemm_fx_supported(const Ctx&, const Gemm& g, const GemmFx& fx) {
  if (!g_gemmfx || !g.A.kmajor || g.KB != 1 || g.atomic) return false;
  const bool frames = fx.a_pro == APRO_MASKSCALE || fx.epi == EPI_XCBWD || fx.epi == EPI_COLSUM;
  if (frames && (fx.rpf <= 0 || g.batch != 1 || g.M % fx.rpf)) return false;
  return true;
}
void gemm_fx(const Ctx& ctx, const Gemm& g, const GemmFx& fx) {
  const int E = ctx.mode;
  auto rnd = [E](float v) { return E == DT_BF16 ? bf2f(f2bf(v)) : v; };
  std::vector<float> Ap((size_t)g.batch * g.M * g.K);
  for (int b = 0; b < g.batch; ++b)
    for (int m = 0; m < g.M; ++m)
      for (int k = 0; k < g.K; ++k) {
        const long ao = (long)b * g.A.bs + (long)m * g.A.ld + k;
        float a = ld(g.A.p, E, ao);
        if (fx.a_pro == APRO_MASKSCALE) {
          const int f = m / fx.rpf;
          float v = (fx.a_rs ? fx.a_rs[m] : 1.f) * fx.a_scale;
          v *= ld(fx.a_cs, fx.a_cs_dt, (long)f * fx.a_cs_ld + k) * (fx.a_cs2 ? fx.a_cs2[k] : 1.f);
          a = a > 0.f ? rnd(v) : 0.f;
        } else if (fx.a_pro == APRO_BNBWD) {
          const int c = b * g.K + k;
          const float x = ld(fx.a2, E, ao), sc = fx.bn_sc[c];
          float k2 = 0.f, k3 = 0.f;
          if (fx.bn_training) {
            const float inv = 1.f / (float)fx.bn_rows;
            k3 = sc * fx.bn_rstd[c] * fx.bn_sums[fx.bn_C + c] * inv;
            k2 = sc * fx.bn_sums[c] * inv - fx.bn_mean[c] * k3;
          }
          float gg = a;
          if (fx.bn_relu && !(x * sc + fx.bn_sh[c] > 0.f)) gg = 0.f;
          a = rnd(sc * gg - k2 - x * k3);
        }
        Ap[((size_t)b * g.M + m) * g.K + k] = a;
      }
  if (fx.a_store)
    for (int b = 0; b < g.batch; ++b)
      for (int m = 0; m < g.M; ++m)
        for (int k = 0; k < g.K; ++k) st(fx.a_store, E, (long)b * g.A.bs + (long)m * g.A.ld + k, Ap[((size_t)b * g.M + m) * g.K + k]);
  std::vector<double> cs1(fx.epi == EPI_COLSTATS ? (size_t)g.batch * g.N : 0, 0.0), cs2(cs1.size(), 0.0);
  for (int b = 0; b < g.batch; ++b)
    for (int m = 0; m < g.M; ++m)
      for (int n = 0; n < g.N; ++n) {
        double acc = 0;
        for (int k = 0; k < g.K; ++k) {
          const long bo = (long)b * g.B.bs + (g.B.kmajor ? (long)n * g.B.ld + k : (long)k * g.B.ld + n);
          acc += (double)Ap[((size_t)b * g.M + m) * g.K + k] * (double)ld(g.B.p, E, bo);
        }
        float v = (float)acc + (g.bias_n ? g.bias_n[n] : 0.f);
        if (g.act == ACT_RELU) v = std::max(v, 0.f);
        const long o = (long)b * g.dbs + (long)m * g.ldd + n;
        if (fx.epi == EPI_COLSTATS) {
          const double vr = rnd(v);                       // (double partial sums: the host loops are the ideal evaluation of the schedule)
          cs1[(size_t)b * g.N + n] += vr;
          cs2[(size_t)b * g.N + n] += vr * vr;
        } else if (fx.epi == EPI_COLSUM) {
          const int f = m / fx.rpf;
          const float vr = rnd(v);
          fx.e_acc[(long)f * fx.e_ld + n] += fx.e_scale * vr;
          if (vr > 0.f) fx.e_acc2[(long)f * fx.e_ld + n] += 1.f;
        }
        if (fx.epi == EPI_XCBWD) {
          const int f = m / fx.rpf;
          const float vr = rnd(v);
          fx.e_acc[(long)f * fx.e_ld + n] += vr * ld(fx.e_x, E, o);
          v = ld(g.R, g.rdt, (long)b * g.rbs + (long)m * g.ldr + n) + vr * (1.f + fx.e_cs[(long)f * fx.e_ld + n]);
        } else if (g.R) {
          v += g.beta * ld(g.R, g.rdt, (long)b * g.rbs + (long)m * g.ldr + n);
        }
        st(g.D, g.ddt, o, v);
      }
  for (size_t i = 0; i < cs1.size(); ++i) { fx.e_acc[i] += (float)cs1[i]; fx.e_acc2[i] += (float)cs2[i]; }
}

void rowdot_batched(const Ctx& ctx, const void* x, long ldx, long bs, int B, int N, int C, const void* w, int wdt, long w_bs,
                    const float* w2, const float* bias, float* out) {
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int c = 0; c < C; ++c)
        s += (double)ld(x, ctx.mode, (long)b * bs + (long)n * ldx + c) * ld(w, wdt, (long)b * w_bs + c) * (w2 ? w2[c] : 1.f);
      out[(long)b * N + n] = (float)s + (bias ? *bias : 0.f);
    }
}

void rowdot_colsum(const Ctx& ctx, const void* x, long ldx, long bs, int B, int N, int C, const float* w, const float* roww,
                   float* out_row, float* out_col, float*, long) {
  if (out_row)
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) s += (double)ld(x, ctx.mode, (long)b * bs + (long)n * ldx + c) * w[c];
      out_row[n] += (float)s;
    }
  if (out_col)
    for (int c = 0; c < C; ++c) {
      double s = 0;
      for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) s += (double)(roww ? roww[n] : 1.f) * ld(x, ctx.mode, (long)b * bs + (long)n * ldx + c);
      out_col[c] += (float)s;
    }
}
void rowdot_colsum_frames(const Ctx& ctx, const void* x, long ld, long bs, int B, int N, int C, const float* w, const float* roww,
                          float* out_row, float* out_col, float*, float*, long) {
  rowdot_colsum(ctx, x, ld, bs, B, N, C, w, roww, out_row, out_col);
}

void sum_batch(const Ctx&, const float* in, long bs, int B, long n, float* out, float scale, int accumulate) {
  for (long i = 0; i < n; ++i) {
    double s = 0;
    for (int b = 0; b < B; ++b) s += in[(long)b * bs + i];
    out[i] = (accumulate ? out[i] : 0.f) + scale * (float)s;
  }
}

void scale_cols(const Ctx& ctx, const void* x, void* y, int B, int N, int C, const float* colw, float add) {
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < C; ++c) {
        const long o = ((long)b * N + n) * C + c;
        st(y, ctx.mode, o, ld(x, ctx.mode, o) * (add + colw[(long)b * C + c]));
      }
}

long row_part_floats(int B, int C) { return 0 * (long)B * C; }

void outer_rows(const Ctx& ctx, const float* roww, const float* colw, int B, int N, int C, void* y) {
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < C; ++c) st(y, ctx.mode, ((long)b * N + n) * C + c, roww[(long)b * N + n] * colw[(long)b * C + c]);
}

void relu_bwd_scale(const Ctx& ctx, const void* x, void* y, int B, int N, int C, const float* roww, const void* colw, int cdt,
                    const float* colw2, float scale, float* colsum_out, float*, long) {
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < C; ++c) {
        const long o = ((long)b * N + n) * C + c;
        float v = 0.f;
        if (ld(x, ctx.mode, o) > 0.f)
          v = (roww ? roww[(long)b * N + n] : 1.f) * scale * ld(colw, cdt, (long)b * C + c) * (colw2 ? colw2[c] : 1.f);
        st(y, ctx.mode, o, v);
        if (colsum_out) colsum_out[c] += ld(y, ctx.mode, o);
      }
}

void xc_bwd(const Ctx& ctx, const void* dXc, const void* X1, void* dX1, int B, int N, int C, const float* ch, float* dch) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      double s = 0;
      for (int n = 0; n < N; ++n) {
        const long o = ((long)b * N + n) * C + c;
        const float g = ld(dXc, ctx.mode, o);
        s += (double)g * ld(X1, ctx.mode, o);
        st(dX1, ctx.mode, o, ld(dX1, ctx.mode, o) + g * (1.f + ch[(long)b * C + c]));
      }
      dch[(long)b * C + c] += (float)s;
    }
}

void spatial_fwd(const Ctx&, const float* sl, int B, int N, float* sg, float* map, float* map2) {
  for (int b = 0; b < B; ++b) {
    const float* x = sl + (long)b * N;
    float m = -INFINITY;
    for (int n = 0; n < N; ++n) m = std::max(m, std::tanh(x[n]));
    double s = 0;
    for (int n = 0; n < N; ++n) s += std::exp(std::tanh(x[n]) - m);
    for (int n = 0; n < N; ++n) {
      sg[(long)b * N + n] = sigm(x[n]);
      map[(long)b * N + n] = (float)(std::exp(std::tanh(x[n]) - m) / s);
      if (map2) map2[(long)b * N + n] = map[(long)b * N + n];
    }
  }
}

void spatial_bwd(const Ctx&, const float* sl, const float* sg, const float* map, const float* dsg, const float* dMap,
                 int B, int N, float* dsl, float* dbs) {
  double tot = 0;
  for (int b = 0; b < B; ++b) {
    const long o = (long)b * N;
    double pd = 0;
    if (dMap) for (int n = 0; n < N; ++n) pd += (double)map[o + n] * dMap[o + n];
    for (int n = 0; n < N; ++n) {
      const float s = sg[o + n];
      float d = dsg[o + n] * s * (1.f - s);
      if (dMap) { const float t = std::tanh(sl[o + n]); d += map[o + n] * (dMap[o + n] - (float)pd) * (1.f - t * t); }
      dsl[o + n] = d;
      tot += d;
    }
  }
  *dbs += (float)tot;
}

static void ln_row(std::vector<float>& x, const float* w, const float* b, float eps, float* mu, float* rstd) {
  const int C = (int)x.size();
  double s = 0; for (float v : x) s += v;
  const float mean = (float)(s / C);
  double q = 0; for (float v : x) q += (double)(v - mean) * (v - mean);
  const float rs = 1.f / std::sqrt((float)(q / C) + eps);
  for (int c = 0; c < C; ++c) x[c] = (x[c] - mean) * rs * w[c] + b[c];
  *mu = mean; *rstd = rs;
}

void modln_fwd(const Ctx& ctx, const void* X1, const float* ch, const float* sg, const float* tg, float alpha, float beta,
               float gamma, const float* lnw, const float* lnb, float eps, int B, int N, int C, void* X3, float* mu, float* rstd) {
  std::vector<float> x(C);
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      const long row = (long)b * N + n;
      for (int c = 0; c < C; ++c)
        x[c] = ld(X1, ctx.mode, row * C + c) * (alpha * ch[(long)b * C + c] + beta * sg[row] + (tg ? gamma * tg[b] : 0.f) + 1.f - alpha);
      if (lnw) ln_row(x, lnw, lnb, eps, mu + row, rstd + row);
      for (int c = 0; c < C; ++c) st(X3, ctx.mode, row * C + c, x[c]);
    }
}

void modln_bwd(const Ctx& ctx, const void* dX3, const void* X1, const float* ch, const float* sg, const float* tg, float alpha,
               float beta, float gamma, const float* lnw, const float* mu, const float* rstd, int B, int N, int C,
               void* dX1, float* dlnw, float* dlnb, float* dch, float* dsg, float* dtg, float*, long) {
  std::vector<float> g(C), xh(C), md(C), x1(C);
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      const long row = (long)b * N + n;
      double s1 = 0, s2 = 0;
      for (int c = 0; c < C; ++c) {
        md[c] = alpha * ch[(long)b * C + c] + beta * sg[row] + (tg ? gamma * tg[b] : 0.f) + 1.f - alpha;
        x1[c] = ld(X1, ctx.mode, row * C + c);
        g[c] = ld(dX3, ctx.mode, row * C + c);
        if (lnw) {
          xh[c] = (x1[c] * md[c] - mu[row]) * rstd[row];
          dlnw[c] += g[c] * xh[c];
          dlnb[c] += g[c];
          g[c] *= lnw[c];
          s1 += g[c]; s2 += (double)g[c] * xh[c];
        }
      }
      double rsum = 0;
      for (int c = 0; c < C; ++c) {
        const float dx2 = lnw ? rstd[row] * (g[c] - (float)(s1 / C) - xh[c] * (float)(s2 / C)) : g[c];
        const float dm = dx2 * x1[c];
        dch[(long)b * C + c] += alpha * dm;
        rsum += dm;
        st(dX1, ctx.mode, row * C + c, dx2 * md[c]);
      }
      dsg[row] = beta * (float)rsum;
      if (tg && dtg) dtg[b] += gamma * (float)rsum;
    }
}

bool gproj_supported(int mode, int C, int ds, int g) {
  if (getenv("DGSCT_NO_GPROJ") && atoi(getenv("DGSCT_NO_GPROJ"))) return false;
  if (g < 1 || C % g || ds % g) return false;
  const int ve = mode == DT_BF16 ? 8 : 4, cg = C / g, dg = ds / g;
  if (dg < 1 || dg > 16 || (dg & 1) || cg % ve) return false;
  int gs = 1;
  while (gs < C / ve) gs <<= 1;
  return gs <= 64 && ds <= gs && C <= 512 && (long)ds * cg <= 4096;
}
void gproj_narrow(const Ctx& ctx, const void* x, long rows, int C, int ds, int g, const float* W, long sg, long sj, long sc, void* y);
void gproj_narrow_bnb(const Ctx& ctx, const void* dy, const void* xv, void* dx, long rows, int C, int ds, int g, const float* W, long sg,
                      long sj, long sc, void* y, const float* mean, const float* rstd, const float* bsc, const float* bsh, const float* sums, int training) {
  bn_bwd_apply(ctx, dy, xv, dx, rows, C, mean, rstd, bsc, bsh, sums, 0, 1, training);
  gproj_narrow(ctx, dx, rows, C, ds, g, W, sg, sj, sc, y);
}
void gproj_narrow(const Ctx& ctx, const void* x, long rows, int C, int ds, int g, const float* W, long sg, long sj, long sc,
                  void* y) {
  const int cg = C / g, dg = ds / g;
  for (long r = 0; r < rows; ++r)
    for (int gi = 0; gi < g; ++gi)
      for (int jl = 0; jl < dg; ++jl) {
        double s = 0;
        for (int cl = 0; cl < cg; ++cl) s += (double)ld(x, ctx.mode, r * C + gi * cg + cl) * W[gi * sg + jl * sj + cl * sc];
        st(y, ctx.mode, r * ds + gi * dg + jl, (float)s);
      }
}
static int g_rowfuse = 1;
int rowfuse_mode(int set) { const int old = g_rowfuse; if (set >= 0) g_rowfuse = set ? 1 : 0; return old; }
bool modln_gproj_supported(int mode, int C, int ds, int g) { return g_rowfuse && gproj_supported(mode, C, ds, g); }

void modln_gproj(const Ctx& ctx, const void* X1, const float* ch, const float* sg, const float* tg, float alpha, float beta, float gamma,
                 const float* lnw, const float* lnb, float eps, int B, int N, int C, int ds, int g, const float* W, long wsg, long wsj,
                 long wsc, void* X3, float* mu, float* rstd, void* y, float* stats) {
  modln_fwd(ctx, X1, ch, sg, tg, alpha, beta, gamma, lnw, lnb, eps, B, N, C, X3, mu, rstd);
  gproj_narrow(ctx, X3, (long)B * N, C, ds, g, W, wsg, wsj, wsc, y);
  if (stats) bn_stats(ctx, y, (long)B * N, ds, stats);
}

// ---- fused gate / bottleneck passes (csrc/fused_gate.hip): host loops with the same rounding points (stored tensors rounded to E)
static int g_gatefuse = 1;
int gatefuse_mode(int set) { const int old = g_gatefuse; if (set >= 0) g_gatefuse = set ? 1 : 0; return old; }
bool gate_fused_supported(int, int, int, int ds, int g) { return g_gatefuse && ds % g == 0; }      // the emulation takes every shape
void gatemod_fwd(const Ctx& ctx, const void* X1, const float* ch, const void* aq2, const float* Wv2, const float* bv2, const float* ws,
                 const float* bs, const float* tg, float alpha, float beta, float gamma, const float* lnw, const float* lnb, float eps,
                 int B, int N, int C, int ds, int g, const float* Wd, float* sl, void* X3, float* mu, float* rstd, void* Zp,
                 float* stats, void* vq2) {
  const int dd = C / 2, cg = C / g, dg = ds / g;
  std::vector<double> x2(C);
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      const long row = (long)b * N + n;
      double s = 0;
      for (int j = 0; j < dd; ++j) {
        double a = bv2[j];
        for (int c = 0; c < C; ++c) a += (double)ld(X1, ctx.mode, row * C + c) * (1.0 + ch[(long)b * C + c]) * Wv2[(long)j * C + c];
        const double v = a > 0 ? a : 0;
        if (vq2) st(vq2, ctx.mode, row * dd + j, (float)v);
        s += v * ld(aq2, ctx.mode, (long)b * dd + j) * ws[j];
      }
      s += *bs;
      sl[row] = (float)s;
      const double sgv = 1.0 / (1.0 + std::exp(-s));
      double m1 = 0;
      for (int c = 0; c < C; ++c) {
        x2[c] = (double)ld(X1, ctx.mode, row * C + c) * (alpha * ch[(long)b * C + c] + beta * sgv + (tg ? gamma * tg[b] : 0.0) + 1.0 - alpha);
        m1 += x2[c];
      }
      if (lnw) {
        m1 /= C;
        double v = 0;
        for (int c = 0; c < C; ++c) v += (x2[c] - m1) * (x2[c] - m1);
        const double rs = 1.0 / std::sqrt(v / C + eps);
        mu[row] = (float)m1; rstd[row] = (float)rs;
        for (int c = 0; c < C; ++c) x2[c] = (x2[c] - m1) * rs * lnw[c] + lnb[c];
      }
      for (int c = 0; c < C; ++c) st(X3, ctx.mode, row * C + c, (float)x2[c]);
      for (int jz = 0; jz < ds; ++jz) {
        const int gi = jz / dg;
        double a = 0;
        for (int cl = 0; cl < cg; ++cl) a += (double)ld(X3, ctx.mode, row * C + gi * cg + cl) * Wd[(long)jz * cg + cl];
        st(Zp, ctx.mode, row * ds + jz, (float)a);
      }
    }
  if (stats) bn_stats(ctx, Zp, (long)B * N, ds, stats);
}

bool gate_bwd_fused_shape(int, int, int, int ds, int g) { return ds % g == 0; }
bool gate_bwd_fused_supported(int mode, int N, int C, int ds, int g) { return gate_fused_supported(mode, N, C, ds, g); }
long gate_bwd_part_floats(int, int) { return 0; }
// composition of the host primitives the device kernel replaces (same order as the unfused schedule)
void gatemod_bwd(const Ctx& ctx, const void* X1, const float* ch, const void* aq2, const float* Wv2, const float* bv2, const float* ws,
                 const float* tg, float alpha, float beta, float gamma, const float* lnw, const float* mu, const float* rstd,
                 const float* sl, const float* sg, const float* map, const float* dMap, int B, int N, int C, int ds, int g, const float* Wd,
                 void* dZ, const void* Zp, const float* bn_mean, const float* bn_rstd, const float* bn_sc, const float* bn_sh,
                 const float* bn_sums, int has_bn, int training, void* dX1, void* dvq2, void* Xc, float* dch, float* u, float* dtg,
                 float* dlnw, float* dlnb, float* dbv2, float* dbs, float*, float*, long) {
  const long R = (long)B * N;
  const int dd = C / 2, cg = C / g, dg = ds / g;
  const size_t es = dt_size(ctx.mode);
  bn_bwd_apply(ctx, dZ, Zp, dZ, R, ds, bn_mean, bn_rstd, bn_sc, bn_sh, bn_sums, 1, has_bn, training);
  std::vector<char> dX3((size_t)R * C * es), dXc((size_t)R * C * es);
  std::vector<float> dsg(R), dsl(R);
  gproj_wide(ctx, dZ, R, C, ds, g, Wd, (long)dg * cg, cg, 1, dX3.data(), nullptr);
  modln_bwd(ctx, dX3.data(), X1, ch, sg, tg, alpha, beta, gamma, lnw, mu, rstd, B, N, C, dX1, dlnw, dlnb, dch, dsg.data(), dtg);
  spatial_bwd(ctx, sl, sg, map, dsg.data(), dMap, B, N, dsl.data(), dbs);
  scale_cols(ctx, X1, Xc, B, N, C, ch, 1.f);
  for (long r = 0; r < R; ++r)                               // vq2 recomputed: relu(Xc Wv2^T + bv2)
    for (int j = 0; j < dd; ++j) {
      double a = bv2[j];
      for (int c = 0; c < C; ++c) a += (double)ld(Xc, ctx.mode, r * C + c) * Wv2[(long)j * C + c];
      st(dvq2, ctx.mode, r * dd + j, a > 0 ? (float)a : 0.f);
    }
  colsum_batched(ctx, dvq2, dd, (long)N * dd, B, N, dd, dsl.data(), N, 1.f, u, dd);
  relu_bwd_scale(ctx, dvq2, dvq2, B, N, dd, dsl.data(), aq2, ctx.mode, ws, 1.f, dbv2);
  for (long r = 0; r < R; ++r)                               // dXc = dvq2 . Wv2
    for (int c = 0; c < C; ++c) {
      double a = 0;
      for (int j = 0; j < dd; ++j) a += (double)ld(dvq2, ctx.mode, r * dd + j) * Wv2[(long)j * C + c];
      st(dXc.data(), ctx.mode, r * C + c, (float)a);
    }
  xc_bwd(ctx, dXc.data(), X1, dX1, B, N, C, ch, dch);
}

void gproj_wide(const Ctx& ctx, const void* x, long rows, int C, int ds, int g, const float* W, long sg, long sj, long sc,
                void* y, float* stats) {
  const int cg = C / g, dg = ds / g;
  for (long r = 0; r < rows; ++r)
    for (int gi = 0; gi < g; ++gi)
      for (int cl = 0; cl < cg; ++cl) {
        double s = 0;
        for (int jl = 0; jl < dg; ++jl) s += (double)ld(x, ctx.mode, r * ds + gi * dg + jl) * W[gi * sg + jl * sj + cl * sc];
        st(y, ctx.mode, r * C + gi * cg + cl, (float)s);
      }
  if (stats) bn_stats(ctx, y, rows, C, stats);
}

void bn_stats(const Ctx& ctx, const void* x, long rows, int C, float* acc) {
  for (int c = 0; c < C; ++c) {
    const float sft = ld(x, ctx.mode, c);
    double s1 = 0, s2 = 0;
    for (long r = 0; r < rows; ++r) { const double d = ld(x, ctx.mode, r * C + c) - sft; s1 += d; s2 += d * d; }
    acc[c] = sft; acc[C + c] += (float)s1; acc[2 * C + c] += (float)s2;
  }
}

void bn_finalize(const Ctx&, const float* acc, long rows, int C, const float* w, const float* b, float* run_mean,
                 float* run_var, float momentum, float eps, int training, float* mean, float* rstd, float* sc, float* sh) {
  for (int c = 0; c < C; ++c) {
    float m, v;
    if (training) {
      const float s1 = acc[C + c] / rows, s2 = acc[2 * C + c] / rows;
      m = acc[c] + s1;
      v = std::max(s2 - s1 * s1, 0.f);
      const float unb = rows > 1 ? v * ((float)rows / (float)(rows - 1)) : v;
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * m;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * unb;
    } else { m = run_mean[c]; v = run_var[c]; }
    const float rs = 1.f / std::sqrt(v + eps);
    mean[c] = m; rstd[c] = rs; sc[c] = w[c] * rs; sh[c] = b[c] - m * sc[c];
  }
}

static int g_bnfold = 1;
int bnfold_mode(int set) { const int old = g_bnfold; if (set >= 0) g_bnfold = set ? 1 : 0; return old; }
void affine_act(const Ctx& ctx, const void* x, void* y, long rows, int C, const float* sc, const float* sh, int relu);
void affine_act_bn(const Ctx& ctx, const void* x, void* y, long rows, int C, const BnFin& f, int relu) {
  bn_finalize(ctx, f.acc, f.rows, C, f.w, f.b, f.run_mean, f.run_var, f.momentum, f.eps, f.training, f.mean, f.rstd, f.sc, f.sh);
  affine_act(ctx, x, y, rows, C, f.sc, f.sh, relu);
}

void affine_act(const Ctx& ctx, const void* x, void* y, long rows, int C, const float* sc, const float* sh, int relu) {
  for (long r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) {
      float v = ld(x, ctx.mode, r * C + c);
      if (sc) v = v * sc[c] + sh[c];
      if (relu) v = std::max(v, 0.f);
      st(y, ctx.mode, r * C + c, v);
    }
}

void bn_bwd_stats(const Ctx& ctx, const void* dy, const void* x, long rows, int C, const float* mean, const float* rstd,
                  const float* sc, const float* sh, int relu, float* sums, float*, long) {
  for (int c = 0; c < C; ++c) {
    double s0 = 0, s1 = 0;
    for (long r = 0; r < rows; ++r) {
      const float t = ld(x, ctx.mode, r * C + c);
      float g = ld(dy, ctx.mode, r * C + c);
      if (relu && !(t * sc[c] + sh[c] > 0.f)) g = 0.f;
      s0 += g; s1 += (double)g * (t - mean[c]) * rstd[c];
    }
    sums[c] += (float)s0; sums[C + c] += (float)s1;
  }
}

void bn_bwd_apply(const Ctx& ctx, const void* dy, const void* x, void* dx, long rows, int C, const float* mean,
                  const float* rstd, const float* sc, const float* sh, const float* sums, int relu, int has_bn, int training) {
  for (long r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) {
      const float t = ld(x, ctx.mode, r * C + c);
      float g = ld(dy, ctx.mode, r * C + c);
      if (has_bn) {
        if (relu && !(t * sc[c] + sh[c] > 0.f)) g = 0.f;
        if (training) g = sc[c] * (g - sums[c] / rows - (t - mean[c]) * rstd[c] * sums[C + c] / rows);
        else g = sc[c] * g;
      } else if (relu && !(t > 0.f)) g = 0.f;
      st(dx, ctx.mode, r * C + c, g);
    }
}

void tail_fwd(const Ctx& ctx, const void* Op, const float* sc2, const float* sh2, const float* lnw, const float* lnb,
              const float* gate, int gate_first, float eps, long rows, int C, void* out, float* mu, float* rstd,
              const void* residual, const BnFin* f) {
  if (f) bn_finalize(ctx, f->acc, f->rows, C, f->w, f->b, f->run_mean, f->run_var, f->momentum, f->eps, f->training, f->mean, f->rstd, f->sc, f->sh);
  std::vector<float> x(C);
  const float gv = gate ? *gate : 1.f;
  for (long r = 0; r < rows; ++r) {
    for (int c = 0; c < C; ++c) {
      float v = ld(Op, ctx.mode, r * C + c);
      if (sc2) v = v * sc2[c] + sh2[c];
      if (gate_first) v *= gv;
      x[c] = v;
    }
    if (lnw) ln_row(x, lnw, lnb, eps, mu + r, rstd + r);
    for (int c = 0; c < C; ++c)
      st(out, ctx.mode, r * C + c, (gate_first ? x[c] : x[c] * gv) + (residual ? ld(residual, ctx.mode, r * C + c) : 0.f));
  }
}

void tail_bwd(const Ctx& ctx, const void* dOut, const void* Op, const float* sc2, const float* sh2, const float* mean2,
              const float* rstd2, const float* lnw, const float* lnb, const float* gate, int gate_first, const float* mu,
              const float* rstd, long rows, int C, void* dO, float* dlnw, float* dlnb, float* dgate, float* bnsums, float eps, float*, long) {
  std::vector<float> g(C), o(C), xh(C), op(C);
  const float gv = gate ? *gate : 1.f;
  const bool closed = gate_first && gate && lnw && gv != 0.f;      // see tail_bwd_body (prims_hip.hip)
  double gsum = 0;
  for (long r = 0; r < rows; ++r) {
    double s1 = 0, s2 = 0;
    for (int c = 0; c < C; ++c) {
      op[c] = ld(Op, ctx.mode, r * C + c);
      o[c] = sc2 ? op[c] * sc2[c] + sh2[c] : op[c];
      g[c] = ld(dOut, ctx.mode, r * C + c);
      if (gate_first) {
        if (lnw) { xh[c] = (o[c] * gv - mu[r]) * rstd[r]; dlnw[c] += g[c] * xh[c]; dlnb[c] += g[c]; g[c] *= lnw[c]; s1 += g[c]; s2 += (double)g[c] * xh[c]; }
      } else {
        float L;
        if (lnw) { xh[c] = (o[c] - mu[r]) * rstd[r]; L = xh[c] * lnw[c] + lnb[c]; } else L = o[c];
        if (gate) gsum += (double)g[c] * L;
        g[c] *= gv;
        if (lnw) { dlnw[c] += g[c] * xh[c]; dlnb[c] += g[c]; g[c] *= lnw[c]; s1 += g[c]; s2 += (double)g[c] * xh[c]; }
      }
    }
    for (int c = 0; c < C; ++c) {
      float t = lnw ? rstd[r] * (g[c] - (float)(s1 / C) - xh[c] * (float)(s2 / C)) : g[c];
      if (gate_first) { if (gate && !closed) gsum += (double)t * o[c]; t *= gv; }
      st(dO, ctx.mode, r * C + c, t);
      if (sc2 && bnsums) { bnsums[c] += t; bnsums[C + c] += t * (op[c] - mean2[c]) * rstd2[c]; }
    }
    if (closed) gsum += (double)C * eps / gv * rstd[r] * rstd[r] * (s2 / C);
  }
  if (gate) *dgate += (float)gsum;
}

void ew(const Ctx&, int op, void* o, int odt, EwArg a, EwArg b, EwArg c, long n, float s, long div) {
  if (div < 1) div = 1;
  for (long i = 0; i < n; ++i) {
    float r;
    switch (op) {
      case EW_MUL: r = ld(a.p, a.dt, i) * ld(b.p, b.dt, i); break;
      case EW_MUL_MASK: r = ld(c.p, c.dt, i) > 0.f ? ld(a.p, a.dt, i) * ld(b.p, b.dt, i) : 0.f; break;
      case EW_SIGMOID_BWD: { const float y = ld(b.p, b.dt, i); r = ld(a.p, a.dt, i) * y * (1.f - y); break; }
      case EW_SCALE: r = s * ld(a.p, a.dt, i); break;
      case EW_ADD_BCAST: r = ld(a.p, a.dt, i) + s * ld(b.p, b.dt, i / div); break;
      case EW_OUTER_ACC: r = ld(o, odt, i) + ld(a.p, a.dt, i / div) * ld(b.p, b.dt, i % div); break;
      case EW_MULB_MASK: r = ld(c.p, c.dt, i) > 0.f ? ld(a.p, a.dt, i) * ld(b.p, b.dt, i % div) : 0.f; break;
      case EW_RND_MUL: { float t = s * ld(a.p, a.dt, i); if (c.dt == DT_BF16) t = bf2f(f2bf(t)); r = t * ld(b.p, b.dt, i); break; }
      case EW_MUL3B: r = ld(a.p, a.dt, i) * ld(b.p, b.dt, i) * ld(c.p, c.dt, i % div); break;
      default: r = ld(a.p, a.dt, i); break;
    }
    st(o, odt, i, r);
  }
}

void ew2(const Ctx& c, EwCall p, EwCall q) {
  ew(c, p.op, p.o, p.odt, p.a, p.b, p.c, p.n, p.s, p.div);
  ew(c, q.op, q.o, q.odt, q.a, q.b, q.c, q.n, q.s, q.div);
}

void temporal_fwd(const Ctx&, const float* a, const float* wt, const float* bt, int B, int C, float* tg) {
  for (int b = 0; b < B; ++b) {
    double s = 0;
    for (int c = 0; c < C; ++c) s += (double)a[(long)b * C + c] * wt[c];
    tg[b] = sigm((float)s + bt[0]);
  }
}

void cvt(const Ctx&, const float* in, void* out, int odt, long n) { for (long i = 0; i < n; ++i) st(out, odt, i, in[i]); }

void cvt_multi(const Ctx&, const CvtSeg* segs, int nseg) {
  for (int s = 0; s < nseg; ++s)
    for (long i = 0; i < segs[s].n; ++i) {
      long si = i;
      if (segs[s].tr_cols > 0) { const long rows = segs[s].n / segs[s].tr_cols, j = i / rows; si = (i - j * rows) * segs[s].tr_cols + j; }
      st(segs[s].dst, segs[s].odt, i, segs[s].src[si]);
    }
}

void split_hilo(const Ctx&, const float* src, long n, void* hi, void* lo) {
  for (long i = 0; i < n; ++i) {
    st(hi, DT_BF16, i, src[i]);
    st(lo, DT_BF16, i, src[i] - ld(hi, DT_BF16, i));
  }
}

void rowsum_f32(const Ctx&, const float* W, int R, int C, float* out) {
  for (int r = 0; r < R; ++r) { double s = 0; for (int c = 0; c < C; ++c) s += W[(long)r * C + c]; out[r] = (float)s; }
}

}  // namespace dgsct

// ---- fused latent-token attention (host loops; same rounding points as csrc/attn.hip: probabilities / dS stored in E,
// fp32 latent tokens enter the logit products as hi + lo (here: exactly), hi only where attn.hip uses the hi part) ------
namespace dgsct {
static inline float rnd(float v, int dt) { return dt == DT_F32 ? v : bf2f(f2bf(v)); }
long tokattn_scratch_floats(int, int, int) { return 64; }
long tok_pack_elems(int nb, int C) { return (long)nb * 96 * C; }
void tok_pack(const Ctx&, const float*, int, int, int, void*, const float*, const float*, float*) {}
void tokattn_fwd(const Ctx& ctx, const void* Yp, const float* T0, int B, int N, int C, int tk, float* tok, float* lse, float* a,
                 void* aE, float*, void*, const void*) {
  const int E = ctx.mode;
  std::vector<double> S(N);
  for (int b = 0; b < B; ++b) {
    for (int t = 0; t < tk; ++t) {
      double mx = -INFINITY;
      for (int n = 0; n < N; ++n) {
        double s = 0;
        for (int c = 0; c < C; ++c) s += (double)T0[(long)t * C + c] * ld(Yp, E, ((long)b * N + n) * C + c);
        S[n] = s; mx = std::max(mx, s);
      }
      double l = 0;
      for (int n = 0; n < N; ++n) l += std::exp(S[n] - mx);
      lse[(long)b * tk + t] = (float)(mx + std::log(l));
      for (int c = 0; c < C; ++c) {
        double o = 0;
        for (int n = 0; n < N; ++n) o += (double)rnd((float)std::exp(S[n] - mx), E) * ld(Yp, E, ((long)b * N + n) * C + c);
        tok[((long)b * tk + t) * C + c] = T0[(long)t * C + c] + (float)(o / l);
      }
    }
    for (int c = 0; c < C; ++c) {
      double s = 0;
      for (int n = 0; n < N; ++n) s += ld(Yp, E, ((long)b * N + n) * C + c);
      const float av = (float)(s / N);
      a[(long)b * C + c] = av;
      if (aE) st(aE, E, (long)b * C + c, av);
    }
  }
}
static void p2_row(const void* X, int E, const float* tokb, long xo, int C, int tk, std::vector<double>& P) {
  double mx = -INFINITY;
  for (int t = 0; t < tk; ++t) {
    double s = 0;
    for (int c = 0; c < C; ++c) s += (double)ld(X, E, xo + c) * tokb[(long)t * C + c];
    P[t] = s; mx = std::max(mx, s);
  }
  double l = 0;
  for (int t = 0; t < tk; ++t) { P[t] = std::exp(P[t] - mx); l += P[t]; }
  for (int t = 0; t < tk; ++t) P[t] /= l;
}
void xattn_fwd(const Ctx& ctx, const void* X, const float* tok, const float* gate_av, int B, int N, int C, int tk, void* X1,
               const void*) {
  const int E = ctx.mode;
  std::vector<double> P(tk);
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      const long xo = ((long)b * N + n) * C;
      const float* tokb = tok + (long)b * tk * C;
      p2_row(X, E, tokb, xo, C, tk, P);
      for (int c = 0; c < C; ++c) {
        double o = 0;
        for (int t = 0; t < tk; ++t) o += (double)rnd((float)P[t], E) * tokb[(long)t * C + c];
        st(X1, E, xo + c, ld(X, E, xo + c) + *gate_av * (float)o);
      }
    }
}
void xattn_bwd(const Ctx& ctx, const void* X, const void* dX1, const float* tok, const float* gate_av, int B, int N, int C, int tk,
               void* dX, const void* R2, float* dtok, float* dgate, const void*) {
  const int E = ctx.mode;
  const float g = *gate_av;
  std::vector<double> P(tk), U(tk);
  std::vector<float> Pe(tk), dSe(tk);
  double dg = 0;
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      const long xo = ((long)b * N + n) * C;
      const float* tokb = tok + (long)b * tk * C;
      p2_row(X, E, tokb, xo, C, tk, P);
      double dot = 0;
      for (int t = 0; t < tk; ++t) {
        double u = 0;
        for (int c = 0; c < C; ++c) u += (double)ld(dX1, E, xo + c) * tokb[(long)t * C + c];
        U[t] = u; dot += P[t] * u;
      }
      dg += dot;
      for (int t = 0; t < tk; ++t) { Pe[t] = rnd((float)P[t], E); dSe[t] = rnd((float)(g * P[t] * (U[t] - dot)), E); }
      for (int c = 0; c < C; ++c) {
        double o = 0;
        for (int t = 0; t < tk; ++t) o += (double)dSe[t] * rnd(tokb[(long)t * C + c], E);
        float v = rnd(ld(dX1, E, xo + c) + (float)o, E);
        if (R2) v += ld(R2, E, xo + c);
        st(dX, E, xo + c, v);
      }
      for (int t = 0; t < tk; ++t)
        for (int c = 0; c < C; ++c)
          dtok[((long)b * tk + t) * C + c] += g * Pe[t] * ld(dX1, E, xo + c) + dSe[t] * ld(X, E, xo + c);
    }
  if (dgate) *dgate += (float)dg;
}
void tokattn_bwd(const Ctx& ctx, const void* Yp, const float* T0, const float* tok, const float* lse, const float* dtok,
                 const float* da, float invN, int B, int N, int C, int tk, void* dYp, float* dT0b, float*, const void*, void*) {
  const int E = ctx.mode;
  std::vector<float> P1((size_t)tk * N), dS1((size_t)tk * N);
  for (int b = 0; b < B; ++b) {
    for (int t = 0; t < tk; ++t) {
      double D = 0;
      for (int c = 0; c < C; ++c) D += (double)dtok[((long)b * tk + t) * C + c] * (tok[((long)b * tk + t) * C + c] - T0[(long)t * C + c]);
      for (int n = 0; n < N; ++n) {
        double s = 0, dp = 0;
        for (int c = 0; c < C; ++c) {
          const double y = ld(Yp, E, ((long)b * N + n) * C + c);
          s += (double)T0[(long)t * C + c] * y;
          dp += (double)rnd(dtok[((long)b * tk + t) * C + c], E) * y;
        }
        const double p = std::exp(s - lse[(long)b * tk + t]);
        P1[(size_t)t * N + n] = rnd((float)p, E);
        dS1[(size_t)t * N + n] = rnd((float)(p * (dp - D)), E);
      }
    }
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < C; ++c) {
        double o = 0;
        for (int t = 0; t < tk; ++t)
          o += (double)P1[(size_t)t * N + n] * rnd(dtok[((long)b * tk + t) * C + c], E) + (double)dS1[(size_t)t * N + n] * rnd(T0[(long)t * C + c], E);
        st(dYp, E, ((long)b * N + n) * C + c, (float)o + da[(long)b * C + c] * invN);
      }
    for (int t = 0; t < tk; ++t)
      for (int c = 0; c < C; ++c) {
        double o = 0;
        for (int n = 0; n < N; ++n) o += (double)dS1[(size_t)t * N + n] * ld(Yp, E, ((long)b * N + n) * C + c);
        dT0b[((long)b * tk + t) * C + c] += (float)o;
      }
  }
}
void colsum_multi(const Ctx&, const ColsumSeg* segs, int nseg) {
  for (int s = 0; s < nseg; ++s)
    for (int c = 0; c < segs[s].C; ++c) {
      double a = 0;
      for (int r = 0; r < segs[s].rows; ++r) a += ld(segs[s].x, segs[s].dt, (long)r * segs[s].C + c);
      segs[s].out[c] += (float)a;
    }
}
void frame_scale_fwd(const Ctx& ctx, int rows, long inner, float gamma, const void* x, const float* g, void* y) {
  for (int r = 0; r < rows; ++r)
    for (long i = 0; i < inner; ++i) st(y, ctx.mode, (long)r * inner + i, ld(x, ctx.mode, (long)r * inner + i) * (1.f + gamma * g[r]));
}
void frame_scale_bwd(const Ctx& ctx, int rows, long inner, float gamma, const void* x, const float* g, const void* dy, void* dx, float* dg) {
  for (int r = 0; r < rows; ++r) {
    double acc = 0;
    for (long i = 0; i < inner; ++i) {
      const float a = ld(dy, ctx.mode, (long)r * inner + i);
      acc += (double)a * ld(x, ctx.mode, (long)r * inner + i);
      if (dx) st(dx, ctx.mode, (long)r * inner + i, a * (1.f + gamma * g[r]));
    }
    if (dg) dg[r] = (float)(gamma * acc);
  }
}
void temporal_gate_fwd(const Ctx&, int R, int D, float gamma, const float* akv, const float* vkv, const float* vq, const float* aq,
                       const float* wa, const float* ba, const float* wv, const float* bv, float* out_v, float* out_a, float* gate,
                       float* ga, float* gv) {
  for (int r = 0; r < R; ++r) {
    double sa = *ba, sv = *bv;
    for (int c = 0; c < D; ++c) { sa += (double)akv[(long)r * D + c] * wa[c]; sv += (double)vkv[(long)r * D + c] * wv[c]; }
    const float a = sigm((float)sa), v = sigm((float)sv);
    ga[r] = a; gv[r] = v; gate[r] = a * v;
    for (int c = 0; c < D; ++c) { out_v[(long)r * D + c] = vq[(long)r * D + c] * (1.f + gamma * a); out_a[(long)r * D + c] = aq[(long)r * D + c] * (1.f + gamma * v); }
  }
}
void temporal_gate_bwd(const Ctx&, int R, int D, float gamma, const float* akv, const float* vkv, const float* vq, const float* aq,
                       const float* wa, const float* wv, const float* ga, const float* gv, const float* dOv, const float* dOa,
                       const float* dg, float* dakv, float* dvkv, float* dvq, float* daq, float* dwa, float* dba, float* dwv, float* dbv) {
  std::vector<double> A(D, 0.0), V(D, 0.0);
  double sa = 0, sv = 0;
  for (int r = 0; r < R; ++r) {
    double da = 0, dv = 0;
    for (int c = 0; c < D; ++c) {
      const long o = (long)r * D + c;
      da += (double)dOv[o] * vq[o]; dv += (double)dOa[o] * aq[o];
      dvq[o] = dOv[o] * (1.f + gamma * ga[r]); daq[o] = dOa[o] * (1.f + gamma * gv[r]);
    }
    const double dgr = dg ? dg[r] : 0.0;
    const double dpa = (gamma * da + dgr * gv[r]) * ga[r] * (1.0 - ga[r]), dpv = (gamma * dv + dgr * ga[r]) * gv[r] * (1.0 - gv[r]);
    sa += dpa; sv += dpv;
    for (int c = 0; c < D; ++c) {
      const long o = (long)r * D + c;
      dakv[o] = (float)(dpa * wa[c]); dvkv[o] = (float)(dpv * wv[c]);
      A[c] += dpa * akv[o]; V[c] += dpv * vkv[o];
    }
  }
  for (int c = 0; c < D; ++c) { dwa[c] = (float)A[c]; dwv[c] = (float)V[c]; }
  *dba = (float)sa; *dbv = (float)sv;
}
// ---- fp8 e4m3 (OCP "fn": bias 7, max 448, no infinities) host model of v_cvt_pk_fp8_f32 + the fp8 projections -------------
static inline float e4m3_round(float x) {
  if (!(x == x)) return x;
  float a = std::fabs(x);
  if (a > 448.f) a = 448.f;
  if (a == 0.f) return x;
  int e; std::frexp(a, &e); e -= 1;                 // a = m * 2^e, m in [1, 2)
  if (e < -6) e = -6;                                // subnormal range shares the exponent of the smallest normal
  const float step = std::ldexp(1.f, e - 3);
  float q = std::nearbyint(a / step) * step;         // round-to-nearest-even (default rounding mode)
  if (q > 448.f) q = 448.f;
  return x < 0 ? -q : q;
}
static inline unsigned char e4m3_encode(float q) {   // q already representable
  unsigned char s = q < 0 ? 0x80 : 0;
  float a = std::fabs(q);
  if (a == 0.f) return s;
  int e; float m = std::frexp(a, &e); e -= 1; m *= 2.f;        // m in [1,2)
  if (e < -6) { return s | (unsigned char)std::nearbyint(a / std::ldexp(1.f, -9)); }
  return s | (unsigned char)(((e + 7) << 3) | (int)std::nearbyint((m - 1.f) * 8.f));
}
static inline float e4m3_decode(unsigned char b) {
  const int e = (b >> 3) & 15, m = b & 7;
  const float v = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.f + m / 8.f, e - 7);
  return (b & 0x80) ? -v : v;
}
void fp8_quantize(const Ctx&, const float* w, long n, void* out8, float* inv_scale, void*) {
  float amax = 0.f;
  for (long i = 0; i < n; ++i) amax = std::max(amax, std::fabs(w[i]));
  const float scale = amax > 0.f ? 448.f / amax : 1.f;
  *inv_scale = 1.f / scale;
  for (long i = 0; i < n; ++i) ((unsigned char*)out8)[i] = e4m3_encode(e4m3_round(w[i] * scale));
}
void gemm_fp8(const Ctx&, int M, int N, int K, const void* A, long lda, const void* W8, const float* inv_scale, const float* bias,
              int relu, void* D, long ldd, const float* r1_m, const float* r1_n, int m_mod) {
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double acc = 0;
      for (int k = 0; k < K; ++k)
        acc += (double)e4m3_round(ld(A, DT_BF16, (long)m * lda + k)) * e4m3_decode(((const unsigned char*)W8)[(long)n * K + k]);
      float v = *inv_scale * (float)acc + (bias ? bias[n] : 0.f);
      if (r1_m) v += r1_m[m % (m_mod > 0 ? m_mod : 1)] * r1_n[n];
      if (relu) v = std::max(v, 0.f);
      st(D, DT_BF16, (long)m * ldd + n, v);
    }
}
}  // namespace dgsct

// ---- fused window attention of the frozen blocks (wattn.hip): plain loops, same addressing, bf16 storage ------------------------------
namespace dgsct {
namespace {
struct WinGeom { int B, H, W, ws, shift, heads, hd, nwm, n, nwx, nW, L, C; };
static bool win_geom(WinGeom& g, int B, int H, int W, int ws, int shift, int heads, int hd, int nwm) {
  if (B < 1 || ws < 1 || H % ws || W % ws || ws * ws > 144 || (ws * ws) % 4 || hd % 8 || hd > 32 || hd < 8 || shift < 0 || shift >= ws || heads < 1) {
    set_error("window attention (host emulation): unsupported geometry"); return false;
  }
  g = WinGeom{B, H, W, ws, shift, heads, hd, nwm, ws * ws, W / ws, (H / ws) * (W / ws), H * W, heads * hd};
  if (nwm != 1 && nwm != g.nW) { set_error("window attention (host emulation): bias/mask table with %d window types", nwm); return false; }
  return true;
}
static int win_row(const WinGeom& g, int w, int t) {
  const int wy = w / g.nwx, wx = w % g.nwx, ly = t / g.ws, lx = t % g.ws;
  return ((wy * g.ws + ly + g.shift) % g.H) * g.W + (wx * g.ws + lx + g.shift) % g.W;
}
}  // namespace
// q / k rows of one (frame, window, head) as the kernel holds them in LDS: raw bf16, or (cosine) x / max(|x|, 1e-12) rounded to bf16
static void win_rows(const WinGeom& g, const uint16_t* qkv, int b, int w, int h, int which, int cosine, std::vector<float>& x, std::vector<float>& inv) {
  x.assign((size_t)g.n * g.hd, 0.f); inv.assign(g.n, 1.f);
  for (int t = 0; t < g.n; ++t) {
    const uint16_t* r = qkv + ((long)b * g.L + win_row(g, w, t)) * 3 * g.C + which * g.C + h * g.hd;
    float ss = 0.f;
    for (int d = 0; d < g.hd; ++d) { x[(size_t)t * g.hd + d] = bf2f(r[d]); ss += bf2f(r[d]) * bf2f(r[d]); }
    if (cosine && which < 2) {
      inv[t] = 1.f / std::max(std::sqrt(ss), 1e-12f);
      for (int d = 0; d < g.hd; ++d) x[(size_t)t * g.hd + d] = bf2f(f2bf(x[(size_t)t * g.hd + d] * inv[t]));
    }
  }
}
int window_attn_forward(void*, int B, int H, int W, int ws, int shift, int heads, int hd, int nwm, const void* qkv_, const float* bm,
                        const float* scale, void* out_, float* lse, int cosine) {
  WinGeom g;
  if (!win_geom(g, B, H, W, ws, shift, heads, hd, nwm)) return 2;
  const uint16_t* qkv = (const uint16_t*)qkv_; uint16_t* out = (uint16_t*)out_;
  const int n = g.n;
  std::vector<float> s(n), Q, K, V, iq, ik, iv;
  for (int b = 0; b < B; ++b) for (int w = 0; w < g.nW; ++w) for (int h = 0; h < heads; ++h) {
    const float* bmh = bm + ((long)(w % nwm) * heads + h) * n * n;
    win_rows(g, qkv, b, w, h, 0, cosine, Q, iq); win_rows(g, qkv, b, w, h, 1, cosine, K, ik); win_rows(g, qkv, b, w, h, 2, 0, V, iv);
    for (int i = 0; i < n; ++i) {
      float mx = -INFINITY;
      for (int j = 0; j < n; ++j) {
        float a = 0.f;
        for (int d = 0; d < hd; ++d) a += Q[(size_t)i * hd + d] * K[(size_t)j * hd + d];
        s[j] = a * scale[h] + bmh[(long)i * n + j];
        mx = std::max(mx, s[j]);
      }
      float sum = 0.f;
      for (int j = 0; j < n; ++j) { s[j] = std::exp(s[j] - mx); sum += s[j]; }
      uint16_t* o = out + ((long)b * g.L + win_row(g, w, i)) * g.C + h * hd;
      for (int d = 0; d < hd; ++d) {
        float a = 0.f;
        for (int j = 0; j < n; ++j) a += bf2f(f2bf(s[j])) * V[(size_t)j * hd + d];   // (bf16 probabilities, as on the device)
        o[d] = f2bf(a / sum);
      }
      lse[(((long)b * g.nW + w) * heads + h) * n + i] = mx + std::log(sum);
    }
  }
  return 0;
}
int window_attn_backward(void*, int B, int H, int W, int ws, int shift, int heads, int hd, int nwm, const void* qkv_, const float* bm,
                         const float* scale, const void* out_, const float* lse, const void* dout_, void* dqkv_, int cosine) {
  WinGeom g;
  if (!win_geom(g, B, H, W, ws, shift, heads, hd, nwm)) return 2;
  const uint16_t* qkv = (const uint16_t*)qkv_; const uint16_t* out = (const uint16_t*)out_; const uint16_t* dout = (const uint16_t*)dout_;
  uint16_t* dqkv = (uint16_t*)dqkv_;
  const int n = g.n;
  std::vector<float> P((size_t)n * n), dS((size_t)n * n), Dv(n), Q, K, V, iq, ik, iv, gq(hd), gk(hd);
  for (int b = 0; b < B; ++b) for (int w = 0; w < g.nW; ++w) for (int h = 0; h < heads; ++h) {
    const float* bmh = bm + ((long)(w % nwm) * heads + h) * n * n;
    auto row = [&](int t) { return (long)b * g.L + win_row(g, w, t); };
    win_rows(g, qkv, b, w, h, 0, cosine, Q, iq); win_rows(g, qkv, b, w, h, 1, cosine, K, ik); win_rows(g, qkv, b, w, h, 2, 0, V, iv);
    for (int i = 0; i < n; ++i) {
      float d = 0.f;
      for (int c = 0; c < hd; ++c) d += bf2f(out[row(i) * g.C + h * hd + c]) * bf2f(dout[row(i) * g.C + h * hd + c]);
      Dv[i] = d;
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
      float a = 0.f, dp = 0.f;
      for (int c = 0; c < hd; ++c) {
        a += Q[(size_t)i * hd + c] * K[(size_t)j * hd + c];
        dp += bf2f(dout[row(i) * g.C + h * hd + c]) * V[(size_t)j * hd + c];
      }
      const float p = std::exp(a * scale[h] + bmh[(long)i * n + j] - lse[(((long)b * g.nW + w) * heads + h) * n + i]);
      P[(size_t)i * n + j] = bf2f(f2bf(p));
      dS[(size_t)i * n + j] = bf2f(f2bf(p * (dp - Dv[i])));
    }
    // the gradient of a (normalised) row g -> the raw row: inv * (g - xn (xn . g))   (cosine mode; identity otherwise)
    auto through_norm = [&](std::vector<float>& gr, const std::vector<float>& X, const std::vector<float>& inv, int t) {
      if (!cosine) return;
      float dot = 0.f;
      for (int c = 0; c < hd; ++c) dot += gr[c] * X[(size_t)t * hd + c];
      for (int c = 0; c < hd; ++c) gr[c] = inv[t] * (gr[c] - X[(size_t)t * hd + c] * dot);
    };
    for (int i = 0; i < n; ++i) {
      for (int c = 0; c < hd; ++c) {
        float a = 0.f;
        for (int j = 0; j < n; ++j) a += dS[(size_t)i * n + j] * K[(size_t)j * hd + c];
        gq[c] = a * scale[h];
      }
      through_norm(gq, Q, iq, i);
      for (int c = 0; c < hd; ++c) dqkv[row(i) * 3 * g.C + h * hd + c] = f2bf(gq[c]);
    }
    for (int j = 0; j < n; ++j) {
      for (int c = 0; c < hd; ++c) {
        float ak = 0.f, av = 0.f;
        for (int i = 0; i < n; ++i) {
          ak += dS[(size_t)i * n + j] * Q[(size_t)i * hd + c];
          av += P[(size_t)i * n + j] * bf2f(dout[row(i) * g.C + h * hd + c]);
        }
        gk[c] = ak * scale[h];
        dqkv[row(j) * 3 * g.C + 2 * g.C + h * hd + c] = f2bf(av);
      }
      through_norm(gk, K, ik, j);
      for (int c = 0; c < hd; ++c) dqkv[row(j) * 3 * g.C + g.C + h * hd + c] = f2bf(gk[c]);
    }
  }
  return 0;
}
}  // namespace dgsct
