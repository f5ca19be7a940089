// Grouped projections with a tiny per-group width (bottleneck adapters at the early stages: C = 96/128, ds = C/8 = 12/16,
// g = 2 -> dg = ds/g = 6/8).  As MFMA GEMMs these have N or K = 6..8: a 32-wide tile is 80 % padding and the launch is
// a pure HBM stream that the GEMM engine runs at ~1.1 TB/s.  Here they are row kernels on the vector units: a
// power-of-two group of lanes owns one token row (16 B per lane), the dg x VE weight block of the lane lives in
// registers, and the row never leaves the register file between load, dot products and store.
//   narrow: y[r][gi*dg + jl] = sum_cl x[r][gi*cg + cl] * W(gi, jl, cl)     [rows][C]  -> [rows][ds]
//   wide:   y[r][gi*cg + cl] = sum_jl x[r][gi*dg + jl] * W(gi, jl, cl)     [rows][ds] -> [rows][C]  (+ BatchNorm sums)
// W(gi, jl, cl) = W[gi*sg + jl*sj + cl*sc] addresses the fp32 master weight in either role (forward: down/up sampler,
// backward: the same tensors transposed), so no transposed copy exists.   Reference: net_trans.py:629-643.
#include <atomic>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include "prims.h"
#include "device_util.h"
#include "err.h"

namespace dgsct {

#define STREAM(ctx) ((hipStream_t)(ctx).stream)
static inline long cdivl(long a, long b) { return (a + b - 1) / b; }

// The dg (6, 8, 12 or 16) bf16 values of one narrow row's channel group as packed dwords, elements [0, dg) in w[0 .. dg/2), zero
// above: ONE 16-byte (dg = 8, 16) or 12-byte (dg = 6, 12) load per 8 / 6 values.  As dg/2 separate dword loads (the first
// version) every row cost 3-4 vector-memory instructions per lane and the kernels ran at the CU's address-unit rate, not at HBM
// speed: gproj_wide 68 / 100 us for streams whose HBM time is 32 us (49 / 51 us with this).
struct __attribute__((packed, aligned(4))) ProjU3 { unsigned a, b, c; };
template <int ZW>
__device__ __forceinline__ void ld_narrow_bf16(const unsigned short* p, int dg, unsigned (&w)[ZW]) {
#pragma unroll
  for (int j = 0; j < ZW; ++j) w[j] = 0u;
  if (dg * 2 == ZW * 4) {
#pragma unroll
    for (int q = 0; q < ZW / 4; ++q) {
      const uint4 t = reinterpret_cast<const uint4*>(p)[q];
      w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
    }
  } else if (dg * 2 == ZW * 3) {
#pragma unroll
    for (int q = 0; q < ZW / 4; ++q) {
      const ProjU3 t = reinterpret_cast<const ProjU3*>(p)[q];
      w[3 * q] = t.a; w[3 * q + 1] = t.b; w[3 * q + 2] = t.c;
    }
  } else {
#pragma unroll
    for (int j = 0; j < ZW; ++j) w[j] = reinterpret_cast<const unsigned*>(p)[2 * j < dg ? j : 0];
  }
}

constexpr int PROJ_DG_MAX = 16;     // max per-group narrow width (weight block of a lane in registers: DG x VE floats)

struct ProjGeom { int ve, gs, rpp, rpc, chunks; };
static bool proj_geom(int mode, int C, int ds, int g, long rows, long target_wgs, ProjGeom& pg) {
  if (g < 1 || C % g || ds % g) return false;
  const int cg = C / g, dg = ds / g;
  pg.ve = mode == DT_BF16 ? 8 : 4;
  if (dg < 1 || dg > PROJ_DG_MAX || cg % pg.ve || (dg & 1)) return false;
  const int nvec = C / pg.ve;
  pg.gs = 1;
  while (pg.gs < nvec) pg.gs <<= 1;
  if (pg.gs > 64 || ds > pg.gs || (long)ds * cg > 4096) return false;
  pg.rpp = 256 / pg.gs;
  long chunks = target_wgs;
  const long maxc = cdivl(rows, pg.rpp);
  if (chunks > maxc) chunks = maxc;
  if (chunks < 1) chunks = 1;
  pg.rpc = (int)(cdivl(cdivl(rows, chunks), pg.rpp) * pg.rpp);
  pg.chunks = (int)cdivl(rows, pg.rpc);
  return true;
}

bool gproj_supported(int mode, int C, int ds, int g) {
  static const int off = getenv("DGSCT_NO_GPROJ") ? atoi(getenv("DGSCT_NO_GPROJ")) : 0;
  ProjGeom pg;
  return !off && proj_geom(mode, C, ds, g, 1 << 20, 1024, pg);
}

// ---- narrow ---------------------------------------------------------------------------------------------------------
// Per row: every lane forms its dg partial dot products; the lanes of a channel group are combined through LDS (the
// row group sits inside one wavefront, whose DS instructions execute in order: no workgroup barrier).
// UNR rows per lane per trip: all loads are issued before the first use (latency hiding).  DG = 8 (stage 0: dg = 6, 8)
// keeps 4 rows in flight; DG = 16 (stage 1: dg = 12, 16) has a 128-register weight block and runs 2 rows at 2 WG/CU.

// BNB: the BatchNorm backward of the wide tensor (bn_bwd_apply: dx = k1 dy - k2 - xv k3 per channel) applied to the row on its way in
// and the result stored (the dWu product of the aux stream reads it) -- one pass over dOp less, one launch less on the chain.  The
// projection multiplies the value AS STORED (rounded to E): the numbers are the two-launch path's.
struct BnbArgs { const void* xv; void* dx; const float* mean; const float* rstd; const float* sc; const float* sums; float inv_rows; int training; };
template <int DT, int VE, int PROJ_DG, int PROJ_UNR, bool BNB>
__global__ __launch_bounds__(256, PROJ_DG > 8 ? 2 : 3) void gproj_narrow_k(const void* x, long rows, int C, int ds, int g, const float* W, long sg,
                                                      long sj, long sc, int gs, int rpc, void* y, const BnbArgs bn) {
  // [UNR][row slot][lane][DG] partial dot products (+8 floats per row slot: spreads the slots over the banks)
  __shared__ __attribute__((aligned(16))) float lds[PROJ_UNR * (256 * PROJ_DG + 128 * 8)];
  const int gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const int cg = C / g, dg = ds / g, lpg = cg / VE;
  const int col = gl * VE;
  const bool valid = col < C;
  const int colc = valid ? col : 0;
  const int gi = colc / cg, cl0 = colc - gi * cg;
  // weights: staged through LDS by the whole workgroup (ds*cg <= 4096 floats), then each lane keeps its dg x VE block
  // in registers.  (Per-lane global loads of 64 scattered floats compile to 64 serial branch+load+wait round trips.)
  for (int i = threadIdx.x; i < ds * cg; i += 256) {
    const int cl = i % cg, j = i / cg, gq = j / dg, jl = j - gq * dg;
    lds[i] = W[gq * sg + jl * sj + (long)cl * sc];
  }
  __syncthreads();
  float w[PROJ_DG][VE];
#pragma unroll
  for (int jl = 0; jl < PROJ_DG; ++jl)
#pragma unroll
    for (int e = 0; e < VE; ++e) w[jl][e] = (valid && jl < dg) ? lds[(gi * dg + jl) * cg + cl0 + e] : 0.f;
  __syncthreads();                                    // lds is reused for the partial sums below
  float k1[BNB ? VE : 1], k2[BNB ? VE : 1], k3[BNB ? VE : 1];
  if (BNB) {
    float a[VE], rs[VE], mn[VE], s0[VE], s1[VE];
    ldf<VE>(bn.sc, colc, a); ldf<VE>(bn.rstd, colc, rs); ldf<VE>(bn.mean, colc, mn); ldf<VE>(bn.sums, colc, s0); ldf<VE>(bn.sums, (long)C + colc, s1);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      k1[e] = a[e];
      k3[e] = bn.training ? a[e] * rs[e] * s1[e] * bn.inv_rows : 0.f;
      k2[e] = bn.training ? a[e] * s0[e] * bn.inv_rows - mn[e] * k3[e] : 0.f;
    }
  }
  const int slot_stride = gs * PROJ_DG + 8;
  float* const slot0 = lds + sub * slot_stride;
  const int unr_stride = 256 * PROJ_DG + 128 * 8;
  const int oq = gl / dg, ojl = gl - oq * dg;           // the output element this lane reduces (gl < ds)
  const long r0 = (long)blockIdx.x * rpc;
  const long r_end = lmin_d(rows, r0 + rpc);
  for (long rb = r0; rb < r_end; rb += (long)rpp * PROJ_UNR) {
    // Unconditional loads from clamped addresses: a per-element "load or zero" makes hipcc branch around every load and
    // wait vmcnt(0) after each one, which serialises the four HBM round trips of a trip.
    float t[PROJ_UNR][VE];
    float xv[BNB ? PROJ_UNR : 1][VE];
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      const long row = rb + (long)u * rpp + sub;
      ldv<DT, VE>(x, (row < r_end ? row : r_end - 1) * C + colc, t[u]);
      if (BNB) ldv<DT, VE>(bn.xv, (row < r_end ? row : r_end - 1) * C + colc, xv[u]);
    }
    if (BNB) {
#pragma unroll
      for (int u = 0; u < PROJ_UNR; ++u) {
        const long row = rb + (long)u * rpp + sub;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const float v = k1[e] * t[u][e] - k2[e] - xv[u][e] * k3[e];
          t[u][e] = DT == DT_BF16 ? bf2f(f2bf(v)) : v;
        }
        if (valid && row < r_end) stv<DT, VE>(bn.dx, row * C + colc, t[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      const bool ok = valid && rb + (long)u * rpp + sub < r_end;
#pragma unroll
      for (int e = 0; e < VE; ++e) t[u][e] = ok ? t[u][e] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      float p[PROJ_DG];
#pragma unroll
      for (int jl = 0; jl < PROJ_DG; ++jl) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < VE; ++e) a += t[u][e] * w[jl][e];
        p[jl] = a;
      }
      float4* dst = reinterpret_cast<float4*>(slot0 + u * unr_stride + gl * PROJ_DG);
#pragma unroll
      for (int q4 = 0; q4 < PROJ_DG / 4; ++q4) dst[q4] = make_float4(p[4 * q4], p[4 * q4 + 1], p[4 * q4 + 2], p[4 * q4 + 3]);
    }
    // the row group lives inside one wavefront: its DS instructions execute in order, only the compiler must not move
    // the reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      const long row = rb + (long)u * rpp + sub;
      if (gl < ds && row < r_end) {
        const float* src = slot0 + u * unr_stride + oq * lpg * PROJ_DG + ojl;
        float s = 0.f;
        for (int i = 0; i < lpg; ++i) s += src[i * PROJ_DG];
        ste<DT>(y, row * ds + gl, s);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

static void gproj_narrow_launch(const Ctx& ctx, const void* x, long rows, int C, int ds, int g, const float* W, long sg, long sj, long sc,
                                void* y, const BnbArgs* bn) {
  ProjGeom pg;
  if (!proj_geom(ctx.mode, C, ds, g, rows, 2048, pg)) { set_error("gproj_narrow: unsupported shape C=%d ds=%d g=%d", C, ds, g); return; }
  const bool big = ds / g > 8;
  const BnbArgs b = bn ? *bn : BnbArgs{};
#define NARROW_(DT_, VE_, DG_, UNR_, BNB_) \
  hipLaunchKernelGGL((gproj_narrow_k<DT_, VE_, DG_, UNR_, BNB_>), dim3(pg.chunks), dim3(256), 0, STREAM(ctx), x, rows, C, ds, g, W, sg, sj, sc, \
                     pg.gs, pg.rpc, y, b)
  if (bn) {
    if (ctx.mode == DT_BF16) { if (big) NARROW_(DT_BF16, 8, 16, 2, true); else NARROW_(DT_BF16, 8, 8, 3, true); }   // (3 rows in flight: 4 spill at 3 workgroups per CU)
    else { if (big) NARROW_(DT_F32, 4, 16, 2, true); else NARROW_(DT_F32, 4, 8, 4, true); }
  } else {
    if (ctx.mode == DT_BF16) { if (big) NARROW_(DT_BF16, 8, 16, 2, false); else NARROW_(DT_BF16, 8, 8, 4, false); }
    else { if (big) NARROW_(DT_F32, 4, 16, 2, false); else NARROW_(DT_F32, 4, 8, 4, false); }
  }
#undef NARROW_
}
void gproj_narrow(const Ctx& ctx, const void* x, long rows, int C, int ds, int g, const float* W, long sg, long sj, long sc,
                  void* y) {
  gproj_narrow_launch(ctx, x, rows, C, ds, g, W, sg, sj, sc, y, nullptr);
}
// dx = BN backward of dy (bn_bwd_apply, no ReLU), y = dx (x)_g W -- one pass (gproj_narrow_k<BNB>)
void gproj_narrow_bnb(const Ctx& ctx, const void* dy, const void* xv, void* dx, long rows, int C, int ds, int g, const float* W, long sg,
                      long sj, long sc, void* y, const float* mean, const float* rstd, const float* bsc, const float* bsh, const float* sums, int training) {
  if (!rowfuse_mode(-1)) {
    bn_bwd_apply(ctx, dy, xv, dx, rows, C, mean, rstd, bsc, bsh, sums, 0, 1, training);
    gproj_narrow(ctx, dx, rows, C, ds, g, W, sg, sj, sc, y);
    return;
  }
  const BnbArgs b{xv, dx, mean, rstd, bsc, sums, 1.f / (float)rows, training};
  gproj_narrow_launch(ctx, dy, rows, C, ds, g, W, sg, sj, sc, y, &b);
}

// ---- modulation + ln_before + narrow + BatchNorm sums in ONE pass ------------------------------------------------------
// F8 + F9a of the forward schedule (net_trans.py:611-613, 626-631) for the vector-projection shapes (stages 0-1):
//   X2 = X1 * (alpha ch + beta sg + gamma tg + 1 - alpha);  X3 = LN_C(X2) [optional];  Zp = X3 (x)_g Wd;  BN1 sums of Zp
// were three launches (modln_fwd, gproj_narrow, bn_stats: 74 + 55 + 29 us at 655 360 x 96) that read X1, X3 and Zp from HBM;
// here the row stays in the registers of its lane group from the X1 load to the Zp store: X1 is read once, X3 / mu / rstd /
// Zp are written (the backward needs them), nothing is re-read.  The projection consumes X3 AS STORED (rounded to E) and the
// statistics are those of Zp AS STORED, so the numbers are the three-launch path's.  stats layout = bn_stats' (shift = row 0
// of the tensor, which every lane group recomputes for itself: one extra row per group).
template <int DT, int VE, int PROJ_DG, int PROJ_UNR, int GS>
__global__ __launch_bounds__(256, PROJ_DG > 8 ? 2 : 3) void modln_gproj_k(const void* X1, const float* ch, const float* sg, const float* tg,
                                                     float alpha, float beta, float gamma, const float* lnw, const float* lnb, float eps,
                                                     int N, int C, int ds, int g, const float* W, long wsg, long wsj, long wsc,
                                                     int rpc, void* X3, float* mu, float* rstd, void* y, float* stats) {
  constexpr int gs = GS;
  __shared__ __attribute__((aligned(16))) float lds[PROJ_UNR * (256 * PROJ_DG + 128 * 8)];
  const int gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const int cg = C / g, dg = ds / g, lpg = cg / VE;
  const int col = gl * VE;
  const bool valid = col < C;
  const int colc = valid ? col : 0;
  const int gi = colc / cg, cl0 = colc - gi * cg;
  for (int i = threadIdx.x; i < ds * cg; i += 256) {
    const int cl = i % cg, j = i / cg, gq = j / dg, jl = j - gq * dg;
    lds[i] = W[gq * wsg + jl * wsj + (long)cl * wsc];
  }
  __syncthreads();
  float w[PROJ_DG][VE];
#pragma unroll
  for (int jl = 0; jl < PROJ_DG; ++jl)
#pragma unroll
    for (int e = 0; e < VE; ++e) w[jl][e] = (valid && jl < dg) ? lds[(gi * dg + jl) * cg + cl0 + e] : 0.f;
  __syncthreads();
  float lw[VE], lb[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { lw[e] = 1.f; lb[e] = 0.f; }
  if (lnw) { ldf<VE>(lnw, colc, lw); ldf<VE>(lnb, colc, lb); }
  const int slot_stride = gs * PROJ_DG + 8;
  float* const slot0 = lds + sub * slot_stride;
  const int unr_stride = 256 * PROJ_DG + 128 * 8;
  const int oq = gl / dg, ojl = gl - oq * dg;
  const float invC = 1.f / (float)C;
  auto modulation = [&](int b, float (&cm)[VE]) {
    float t[VE];
    ldf<VE>(ch + (long)b * C, colc, t);
    const float tgv = tg ? gamma * tg[b] : 0.f;
#pragma unroll
    for (int e = 0; e < VE; ++e) cm[e] = alpha * t[e] + 1.f - alpha + tgv;
  };
  // PROJ_UNR rows of this lane group: load, modulate, normalise, (store X3, mu, rstd), project; out[u] = Zp[row u][gl] in
  // the lanes gl < ds
  // a row vector is 16 bytes in either mode (8 bf16 / 4 fp32): kept raw while in flight
  auto rowload = [&](const long (&row)[PROJ_UNR], uint4 (&raw)[PROJ_UNR], float (&sgv)[PROJ_UNR]) {
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      raw[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(X1) + (row[u] * C + colc) * (DT == DT_BF16 ? 2 : 4));
      sgv[u] = beta * sg[row[u]];
    }
  };
  auto rowpass = [&](const long (&row)[PROJ_UNR], const bool (&ok)[PROJ_UNR], const float (&cm)[VE], bool store, const uint4 (&raw)[PROJ_UNR],
                     const float (&sgv)[PROJ_UNR], float (&out)[PROJ_UNR]) {
    float t[PROJ_UNR][VE];
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      const unsigned r4[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (DT == DT_BF16) { t[u][(2 * i) % VE] = __uint_as_float(r4[i] << 16); t[u][(2 * i + 1) % VE] = __uint_as_float(r4[i] & 0xffff0000u); }
        else t[u][i % VE] = __uint_as_float(r4[i]);
      }
    }
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < VE; ++e) { t[u][e] = valid ? t[u][e] * (cm[e] + sgv[u]) : 0.f; s += t[u][e]; }
      if (lnw) {
        const float mean = group_sum(s, gs) * invC;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < VE; ++e) { const float d = valid ? t[u][e] - mean : 0.f; q += d * d; }
        const float rs = rsqrtf(group_sum(q, gs) * invC + eps);
#pragma unroll
        for (int e = 0; e < VE; ++e) t[u][e] = (t[u][e] - mean) * rs * lw[e] + lb[e];
        if (store && ok[u] && gl == 0) { mu[row[u]] = mean; rstd[row[u]] = rs; }
      }
      if (DT == DT_BF16) {                              // round once: the stored X3 and the projection operand are the same bits
        unsigned pk[VE / 2];
#pragma unroll
        for (int i = 0; i < VE / 2; ++i) pk[i] = f2bf2(t[u][2 * i], t[u][2 * i + 1]);
        if (store && ok[u] && valid)
          *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(X3) + row[u] * C + col) = make_uint4(pk[0], pk[1], pk[2 % (VE / 2)], pk[3 % (VE / 2)]);
#pragma unroll
        for (int i = 0; i < VE / 2; ++i) { t[u][2 * i] = __uint_as_float(pk[i] << 16); t[u][2 * i + 1] = __uint_as_float(pk[i] & 0xffff0000u); }
      } else if (store && ok[u] && valid) {
        stv<DT, VE>(X3, row[u] * C + col, t[u]);
      }
      if (!(valid && ok[u])) {
#pragma unroll
        for (int e = 0; e < VE; ++e) t[u][e] = 0.f;
      }
      float p[PROJ_DG];
#pragma unroll
      for (int jl = 0; jl < PROJ_DG; ++jl) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < VE; ++e) a += t[u][e] * w[jl][e];
        p[jl] = a;
      }
      float4* dst = reinterpret_cast<float4*>(slot0 + u * unr_stride + gl * PROJ_DG);
#pragma unroll
      for (int q4 = 0; q4 < PROJ_DG / 4; ++q4) dst[q4] = make_float4(p[4 * q4], p[4 * q4 + 1], p[4 * q4 + 2], p[4 * q4 + 3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      float s = 0.f;
      if (gl < ds) {
        const float* src = slot0 + u * unr_stride + oq * lpg * PROJ_DG + ojl;
        for (int i = 0; i < lpg; ++i) s += src[i * PROJ_DG];
      }
      out[u] = s;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto rounded = [](float v) { return DT == DT_BF16 ? bf2f(f2bf(v)) : v; };

  float sft = 0.f, acc0 = 0.f, acc1 = 0.f;
  float cm[VE];
  if (stats) {                                        // shift = Zp[0][gl] (row 0 of frame 0), recomputed by every lane group
    long row[PROJ_UNR]; bool ok[PROJ_UNR]; float out[PROJ_UNR];
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) { row[u] = 0; ok[u] = true; }
    uint4 t[PROJ_UNR]; float sgv[PROJ_UNR];
    modulation(0, cm);
    rowload(row, t, sgv);
    rowpass(row, ok, cm, false, t, sgv, out);
    sft = rounded(out[0]);
    if (blockIdx.x == 0 && blockIdx.y == 0 && sub == 0 && gl < ds) stats[gl] = sft;
  }
  const int b = blockIdx.y;
  modulation(b, cm);
  const int n0 = blockIdx.x * rpc;
  const int n_end = imin_d(N, n0 + rpc);
  const long base = (long)b * N;
  // the next trip's rows are in flight while this one is normalised and projected (unconditional loads from clamped rows)
  auto rows_of = [&](int nb, long (&row)[PROJ_UNR], bool (&ok)[PROJ_UNR]) {
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      const int n = nb + u * rpp + sub;
      ok[u] = n < n_end;
      row[u] = base + (ok[u] ? n : n_end - 1);
    }
  };
  uint4 tc[PROJ_UNR]; float sc_[PROJ_UNR];
  {
    long row[PROJ_UNR]; bool ok[PROJ_UNR];
    rows_of(n0, row, ok);
    rowload(row, tc, sc_);
  }
  for (int nb = n0; nb < n_end; nb += rpp * PROJ_UNR) {
    long row[PROJ_UNR], rown[PROJ_UNR]; bool ok[PROJ_UNR], okn[PROJ_UNR]; float out[PROJ_UNR];
    uint4 tn[PROJ_UNR]; float sn[PROJ_UNR];
    rows_of(nb, row, ok);
    rows_of(nb + rpp * PROJ_UNR, rown, okn);
    rowload(rown, tn, sn);
    rowpass(row, ok, cm, true, tc, sc_, out);
#pragma unroll
    for (int u = 0; u < PROJ_UNR; ++u) {
      if (gl < ds && ok[u]) {
        ste<DT>(y, row[u] * ds + gl, out[u]);
        const float d = rounded(out[u]) - sft;
        acc0 += d; acc1 += d * d;
      }
      tc[u] = tn[u];
      sc_[u] = sn[u];
    }
  }
  if (stats) {                                        // combine the row slots of the workgroup, one atomic per channel and sum
    __syncthreads();
    if (gl < ds) { lds[sub * ds + gl] = acc0; lds[256 + sub * ds + gl] = acc1; }      // rpp * ds <= 256 (ds <= gs)
    __syncthreads();
    if (threadIdx.x < 2 * ds) {
      const int q = threadIdx.x / ds, c = threadIdx.x - q * ds;
      float s = 0.f;
      for (int r = 0; r < rpp; ++r) s += lds[q * 256 + r * ds + c];
      unsafeAtomicAdd(stats + ds + q * ds + c, s);
    }
  }
}

int rowfuse_mode(int set) {                              // -1: query; 0 / 1: off / on (DGSCT_NO_ROWFUSE=1 starts it off); returns the old value
  static std::atomic<int> mode{(getenv("DGSCT_NO_ROWFUSE") && atoi(getenv("DGSCT_NO_ROWFUSE"))) ? 0 : 1};
  const int old = mode.load(std::memory_order_relaxed);
  if (set >= 0) mode.store(set ? 1 : 0, std::memory_order_relaxed);
  return old;
}

bool modln_gproj_supported(int mode, int C, int ds, int g) {
  ProjGeom pg;
  return rowfuse_mode(-1) && gproj_supported(mode, C, ds, g) && proj_geom(mode, C, ds, g, 1 << 20, 1024, pg) && (pg.gs == 16 || pg.gs == 32);   // (DGSCT_NO_GPROJ switches this pass off too)
}

void modln_gproj(const Ctx& ctx, const void* X1, const float* ch, const float* sg, const float* tg, float alpha, float beta, float gamma,
                 const float* lnw, const float* lnb, float eps, int B, int N, int C, int ds, int g, const float* W, long wsg, long wsj,
                 long wsc, void* X3, float* mu, float* rstd, void* y, float* stats) {
  ProjGeom pg;
  if (!proj_geom(ctx.mode, C, ds, g, N, 1, pg) || (pg.gs != 16 && pg.gs != 32)) {
    set_error("modln_gproj: unsupported shape C=%d ds=%d g=%d", C, ds, g);
    return;
  }
  const bool big = ds / g > 8, bf = ctx.mode == DT_BF16, g32 = pg.gs == 32;
  const int unr = big ? 2 : 4;
  const int trip = pg.rpp * unr;
  int rpc = 0;
  long chunks = 0;
#define MG_(DT_, VE_, DG_, UNR_, GS_)                                                                                                   \
  do {                                                                                                                                  \
    if (!rpc) {                                     /* a reduction (BN sums): one round of resident workgroups */                        \
      long cap = wg_capacity(reinterpret_cast<const void*>(&modln_gproj_k<DT_, VE_, DG_, UNR_, GS_>), 0);                                \
      if (cap > 2048) cap = 2048;                                                                                                       \
      chunks = cap / B;                                                                                                                 \
      const long maxc = cdivl(N, trip);                                                                                                 \
      if (chunks > maxc) chunks = maxc;                                                                                                 \
      if (chunks < 1) chunks = 1;                                                                                                       \
      rpc = (int)(cdivl(cdivl(N, chunks), trip) * trip);                                                                                \
      chunks = cdivl(N, rpc);                                                                                                           \
    }                                                                                                                                   \
    hipLaunchKernelGGL((modln_gproj_k<DT_, VE_, DG_, UNR_, GS_>), dim3((unsigned)chunks, B), dim3(256), 0, STREAM(ctx), X1, ch, sg, tg,  \
                       alpha, beta, gamma, lnw, lnb, eps, N, C, ds, g, W, wsg, wsj, wsc, rpc, X3, mu, rstd, y, stats);                   \
  } while (0)
  if (bf) {
    if (big) { if (g32) MG_(DT_BF16, 8, 16, 2, 32); else MG_(DT_BF16, 8, 16, 2, 16); }
    else     { if (g32) MG_(DT_BF16, 8, 8, 4, 32);  else MG_(DT_BF16, 8, 8, 4, 16); }
  } else {
    if (big) { if (g32) MG_(DT_F32, 4, 16, 2, 32); else MG_(DT_F32, 4, 16, 2, 16); }
    else     { if (g32) MG_(DT_F32, 4, 8, 4, 32);  else MG_(DT_F32, 4, 8, 4, 16); }
  }
#undef MG_
}

// ---- wide -----------------------------------------------------------------------------------------------------------
// stats != null: the bn_stats accumulators of y AS STORED (rounded to E): stats[0..C) = y[0][c] (shift),
// stats[C..2C) += sum (y - shift), stats[2C..3C) += sum (y - shift)^2.
template <int DT, int VE, int PROJ_DG, int PROJ_UNR>
__global__ __launch_bounds__(256, PROJ_DG > 8 ? 2 : 3) void gproj_wide_k(const void* x, long rows, int C, int ds, int g, const float* W, long sg,
                                                    long sj, long sc, int gs, int rpc, void* y, float* stats) {
  __shared__ float lds[4096];
  const int gl = threadIdx.x & (gs - 1), sub = threadIdx.x / gs, rpp = 256 / gs;
  const int cg = C / g, dg = ds / g;
  const int col = gl * VE;
  const bool valid = col < C;
  const int colc = valid ? col : 0;
  const int gi = colc / cg, cl0 = colc - gi * cg;
  for (int i = threadIdx.x; i < ds * cg; i += 256) {      // weights through LDS (see gproj_narrow_k)
    const int cl = i % cg, j = i / cg, gq = j / dg, jl = j - gq * dg;
    lds[i] = W[gq * sg + jl * sj + (long)cl * sc];
  }
  __syncthreads();
  float w[VE][PROJ_DG];
#pragma unroll
  for (int e = 0; e < VE; ++e)
#pragma unroll
    for (int jl = 0; jl < PROJ_DG; ++jl) w[e][jl] = (valid && jl < dg) ? lds[(gi * dg + jl) * cg + cl0 + e] : 0.f;
  __syncthreads();                                        // lds is reused by the statistics flush
  auto project = [&](long row, float (&o)[VE]) {
    float xin[PROJ_DG];
#pragma unroll
    for (int jl = 0; jl < PROJ_DG; ++jl) {
      const float v = lde<DT>(x, row * ds + gi * dg + (jl < dg ? jl : 0));
      xin[jl] = jl < dg ? v : 0.f;
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      float s = 0.f;
#pragma unroll
      for (int jl = 0; jl < PROJ_DG; ++jl) s += xin[jl] * w[e][jl];
      o[e] = s;
    }
  };
  auto rounded = [](float v) { return DT == DT_BF16 ? bf2f(f2bf(v)) : v; };
  float sft[VE], acc[2][VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { sft[e] = 0.f; acc[0][e] = acc[1][e] = 0.f; }
  if (stats && valid) {
    float o[VE];
    project(0, o);
#pragma unroll
    for (int e = 0; e < VE; ++e) sft[e] = rounded(o[e]);
    if (blockIdx.x == 0 && sub == 0) {
#pragma unroll
      for (int e = 0; e < VE; ++e) stats[col + e] = sft[e];
    }
  }
  const long r0 = (long)blockIdx.x * rpc;
  const long r_end = lmin_d(rows, r0 + rpc);
  if (valid) {
    for (long rb = r0 + sub; rb < r_end; rb += (long)rpp * PROJ_UNR) {
      // all narrow-row loads of the trip first (dg values = 12..32 bytes per row, shared by the lanes of a group)
      float xin[PROJ_UNR][PROJ_DG];
#pragma unroll
      for (int u = 0; u < PROJ_UNR; ++u) {
        const long row = rb + (long)u * rpp;
        const bool ok = row < r_end;
        if (DT == DT_BF16) {
          unsigned pk[PROJ_DG / 2];
          ld_narrow_bf16<PROJ_DG / 2>(reinterpret_cast<const unsigned short*>(x) + (ok ? row : r0) * ds + gi * dg, dg, pk);
#pragma unroll
          for (int j2 = 0; j2 < PROJ_DG / 2; ++j2) {
            xin[u][2 * j2] = __uint_as_float(pk[j2] << 16);
            xin[u][2 * j2 + 1] = __uint_as_float(pk[j2] & 0xffff0000u);
          }
        } else {
          const float* px = reinterpret_cast<const float*>(x) + (ok ? row : r0) * ds + gi * dg;
#pragma unroll
          for (int jl = 0; jl < PROJ_DG; ++jl) xin[u][jl] = px[jl < dg ? jl : 0];
        }
      }
#pragma unroll
      for (int u = 0; u < PROJ_UNR; ++u) {
        const long row = rb + (long)u * rpp;
        if (row >= r_end) continue;
        float o[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          float a = 0.f;
#pragma unroll
          for (int jl = 0; jl < PROJ_DG; ++jl) a += xin[u][jl] * w[e][jl];
          o[e] = a;
        }
        stv<DT, VE>(y, row * C + col, o);
        if (stats) {
#pragma unroll
          for (int e = 0; e < VE; ++e) { const float d = rounded(o[e]) - sft[e]; acc[0][e] += d; acc[1][e] += d * d; }
        }
      }
    }
  }
  if (stats) {          // combine the row slots of the workgroup in LDS (store / barrier / column sum), one atomic per channel
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      __syncthreads();
      if (valid) {
#pragma unroll
        for (int e = 0; e < VE; e += 4)
          *reinterpret_cast<float4*>(&lds[sub * C + col + e]) = make_float4(acc[q][e], acc[q][e + 1], acc[q][e + 2], acc[q][e + 3]);
      }
      __syncthreads();
      for (int i = threadIdx.x; i < C; i += 256) {
        float s = 0.f;
        for (int r = 0; r < rpp; ++r) s += lds[r * C + i];
        unsafeAtomicAdd(stats + C + q * C + i, s);
      }
    }
  }
}

void gproj_wide(const Ctx& ctx, const void* x, long rows, int C, int ds, int g, const float* W, long sg, long sj, long sc,
                void* y, float* stats) {
  ProjGeom pg;
  long target = 2048;
  if (stats) {          // a reduction: one full round of resident workgroups (see wg_capacity)
    const bool big0 = ds / g > 8;
    const void* fn = ctx.mode == DT_BF16
        ? (big0 ? reinterpret_cast<const void*>(&gproj_wide_k<DT_BF16, 8, 16, 2>) : reinterpret_cast<const void*>(&gproj_wide_k<DT_BF16, 8, 8, 8>))
        : (big0 ? reinterpret_cast<const void*>(&gproj_wide_k<DT_F32, 4, 16, 2>) : reinterpret_cast<const void*>(&gproj_wide_k<DT_F32, 4, 8, 4>));
    target = wg_capacity(fn, 0);
    if (target > 1024) target = 1024;
  }
  if (!proj_geom(ctx.mode, C, ds, g, rows, target, pg) || C > 512) {
    set_error("gproj_wide: unsupported shape C=%d ds=%d g=%d", C, ds, g);
    return;
  }
  const bool big = ds / g > 8;
#define WIDE_(DT_, VE_, DG_, UNR_) \
  hipLaunchKernelGGL((gproj_wide_k<DT_, VE_, DG_, UNR_>), dim3(pg.chunks), dim3(256), 0, STREAM(ctx), x, rows, C, ds, g, W, sg, sj, sc, \
                     pg.gs, pg.rpc, y, stats)
  if (ctx.mode == DT_BF16) { if (big) WIDE_(DT_BF16, 8, 16, 2); else WIDE_(DT_BF16, 8, 8, 8); }
  else { if (big) WIDE_(DT_F32, 4, 16, 2); else WIDE_(DT_F32, 4, 8, 4); }
#undef WIDE_
}

}  // namespace dgsct
